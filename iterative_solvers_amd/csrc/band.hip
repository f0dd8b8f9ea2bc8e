// band.hip -- CSR SpMV with a sliding window of x kept in LDS (band kernel).
//
// Stands in for hypre_ParCSRMatrixMatvec on matrices whose rows reach a bounded distance from the diagonal but share
// neither offsets (no pattern table) nor column lists (no row groups): finite-element matrices in a bandwidth-reducing
// order.  The stream kernel gathers x per entry from global memory, and what bounds it there is the number of L1 -> L2
// requests of those gathers, not HBM (0.34-0.46 of the roofline on tools/fem_probe.py's `rand` cases).  Here
//   chunk  = 512 / G consecutive rows (the stream kernel's chunk for the same lanes per row);
//   block  = a contiguous run of chunks holding about nnz / blocks entries;
//   window = the last M columns up to the largest column any chunk so far has touched (H_k, a running maximum over the
//            whole matrix, so it only moves forward), kept in an LDS ring of M doubles (slot = column mod M).  A block
//            fills the ring once at its start; every chunk then adds only the columns its successor brings in
//            (H_k, H_k+1], loaded one chunk ahead.  M = the largest H_k - (smallest column of chunk k) + 1: the plan
//            exists when M doubles and the product stage fit half a CU's LDS (or all of it, with one block per CU);
//   pass 1 reads the chunk's (value, column) pairs with coalesced 16- and 8-byte loads requested ahead of their use
//            (two register sets; measured: how far ahead does not matter), multiplies every value with its x
//            from the ring and leaves the products in LDS; pass 2 gives every row G lanes that add the row's products (lane g: entries g, g + G, ..., then a
//            shuffle tree: the stream kernel's order for the same G).
// No gather leaves the CU; x is read once per block.  HBM bytes: 12 per entry + 20 per row + 16 per chunk of descriptors.
// Results agree with the serial reference to 1e-12 (tests), and are the same from run to run (no atomics, fixed order).
// Algorithmic bytes: SURVEY 8(d)'s 12 nnz + 20 n + 4.
#include "mk_internal.h"

namespace mk {

constexpr int kBThreads = 512;
typedef double bd2_t __attribute__((ext_vector_type(2)));
typedef int bi2_t __attribute__((ext_vector_type(2)));
typedef int bi4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------- the plan
// out[k] = (smallest column, largest column, rowptr[k R], 0) of chunk k (rows [k R, (k + 1) R)); (INT_MAX, -1, ..) when it has
// no entry.  One wave per chunk.
__global__ __launch_bounds__(256) void band_span_kernel(int64_t nrows, int R, int nchunks, const int *__restrict__ rowptr,
                                                        const int *__restrict__ col, bi4_t *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < nchunks; k += gridDim.x * 4) {
    const int64_t r0 = (int64_t)k * R, r1 = min(r0 + R, nrows);
    const int b = rowptr[r0], e = rowptr[r1];
    int lo = 0x7fffffff, hi = -1;
    for (int j = b + lane; j < e; j += 64) {
      const int cj = col[j];
      lo = min(lo, cj);
      hi = max(hi, cj);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo = min(lo, __shfl_down(lo, off, 64));
      hi = max(hi, __shfl_down(hi, off, 64));
    }
    if (lane == 0) {
      bi4_t s;
      s.x = lo; s.y = hi; s.z = b; s.w = 0;
      out[k] = s;
    }
  }
}

static int band_lds_budget() {   // bytes of LDS per block: half a CU's 160 KB, so that two blocks of 512 threads share a CU
  const int v = getenv("MK_BAND_LDS") ? std::max(16384, std::min(atoi(getenv("MK_BAND_LDS")), 160 * 1024 - 512)) : 160 * 1024 - 512;
  return v;
}

int build_band_plan(mk_matrix *A) {
  // (read per matrix, not once per process: the analysis runs once per matrix, and tests and probes switch per matrix)
  const int off = getenv("MK_NO_BAND") ? atoi(getenv("MK_NO_BAND")) : 0;
  const double min_len = getenv("MK_BAND_MIN_LEN") ? atof(getenv("MK_BAND_MIN_LEN")) : 16.0;
  const int64_t min_rows = getenv("MK_BAND_MIN_ROWS") ? atoll(getenv("MK_BAND_MIN_ROWS")) : 100000;
  if (off || A->pat || A->grow || A->band_desc || A->nrows < min_rows || A->nnz < 1 || A->info.mean_row_len < min_len) return MK_OK;
  if (A->stream_lanes < 1 || A->stream_lanes > 32 || A->nrows >= ((int64_t)1 << 31) - 65536) return MK_OK;
  Context &c = ctx();
  const int R = kBThreads / (2 * A->stream_lanes);
  const int nchunks = (int)((A->nrows + R - 1) / R);
  bi4_t *d_sp = nullptr;
  if (dev_alloc(reinterpret_cast<int **>(&d_sp), (size_t)nchunks * 4) != MK_OK) return MK_OK;   // an optimisation only
  hipLaunchKernelGGL(band_span_kernel, dim3(std::min((nchunks + 3) / 4, 16384)), dim3(256), 0, c.stream, A->nrows, R, nchunks, A->rowptr, A->col, d_sp);
  std::vector<int> h((size_t)nchunks * 4);
  hipError_t e = hipMemcpyAsync(h.data(), d_sp, sizeof(int) * 4 * (size_t)nchunks, hipMemcpyDeviceToHost, c.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  dev_free(reinterpret_cast<int *>(d_sp));
  if (e != hipSuccess) return fail(MK_ERR_HIP, "band plan: %s", hipGetErrorString(e));
  // M = the widest reach back from the running maximum
  int64_t need = 1;
  {
    int H = -1;
    for (int k = 0; k < nchunks; k++) {
      const int lo = h[4 * k], hi = h[4 * k + 1];
      if (hi < lo) continue;
      if (lo < 0 || hi >= A->ncols) return MK_OK;   // (column indices are validated at assembly; never expected)
      H = std::max(H, hi);
      need = std::max<int64_t>(need, (int64_t)H - lo + 1);
    }
    if (H < 0) return MK_OK;
  }
  const int cap = A->stream_cap;   // entries of the fullest chunk of 256 / stream_lanes = 512 / G rows
  const int64_t M = (need + 7) & ~(int64_t)7;
  const int64_t lds = (M + cap + 4) * 8;
  // the window must fit beside the product stage; a window of more than 64 entries per row of the chunk is a matrix whose
  // rows are not near each other's columns at all (every block would load most of x)
  if (lds > band_lds_budget() || M > (int64_t)A->ncols + 8) return MK_OK;
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  int nb = (int)std::min<int64_t>(((int64_t)nchunks + 7) / 8 * 8, (int64_t)c.num_cus * per_cu);
  nb = std::max(8, nb / 8 * 8);
  if (nb > kSlabStride) nb = kSlabStride;
  // every block reads M entries of x at its start: a block's share of the matrix must be at least four times that (fewer
  // blocks on small matrices)
  {
    const int64_t most = (int64_t)(12.0 * (double)A->nnz / (32.0 * (double)M));
    if (most < 8) return MK_OK;
    if (nb > most) nb = (int)(most / 8 * 8);
  }
  // chunk descriptors (first entry, H_k, first new column a_k = max(H_k-1 + 1, H_k + 1 - M), 0), four slack descriptors behind
  std::vector<int> desc(((size_t)nchunks + 5) * 4, 0);
  {
    int H = -1;
    for (int k = 0; k < nchunks; k++) {
      const int lo = h[4 * k], hi = h[4 * k + 1];
      const int Hprev = H;
      if (hi >= lo) H = std::max(H, hi);
      desc[4 * k] = h[4 * k + 2];
      desc[4 * k + 1] = H;
      desc[4 * k + 2] = (int)std::max<int64_t>((int64_t)Hprev + 1, (int64_t)H + 1 - M);
    }
    for (int k = nchunks; k < nchunks + 5; k++) { desc[4 * k] = (int)A->nnz; desc[4 * k + 1] = H; desc[4 * k + 2] = H + 1; }
  }
  // block b starts at the first chunk whose first entry is at or behind b nnz / nb
  std::vector<int> bstart((size_t)nb + 1, nchunks);
  {
    int k = 0;
    for (int b = 0; b < nb; b++) {
      const int64_t target = (int64_t)((double)A->nnz * (double)b / (double)nb);
      while (k < nchunks && (int64_t)h[4 * k + 2] < target) k++;
      bstart[b] = k;
    }
    bstart[0] = 0;
    bstart[nb] = nchunks;
  }
  if (dev_alloc(&A->band_desc, desc.size()) != MK_OK) return MK_OK;
  if (dev_alloc(&A->band_bstart, bstart.size()) != MK_OK) { dev_free(A->band_desc); A->band_desc = nullptr; return MK_OK; }
  MK_HIP(hipMemcpyAsync(A->band_desc, desc.data(), sizeof(int) * desc.size(), hipMemcpyHostToDevice, c.stream));
  MK_HIP(hipMemcpyAsync(A->band_bstart, bstart.data(), sizeof(int) * bstart.size(), hipMemcpyHostToDevice, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  A->band_rows = R;
  A->band_chunks = nchunks;
  A->band_blocks = nb;
  A->band_window = (int)M;
  return MK_OK;
}

// ------------------------------------------------------------------------------------------ the kernel
// MODE 0: y = alpha A x + beta y.  MODE 1: y = A x, per-block partials of <y,x>, <x,x> (the CG step's K1).
// U: capacity class of the product stage (cap <= 1024 U - 8): a thread holds U (value pair, column pair) loads per chunk.
template <int U>
struct BandSet {   // what a thread holds of one chunk in flight
  bd2_t rv[U];
  bi2_t rc[U];
  double xn;       // the column this thread adds to the ring for that chunk
};

template <int G, int MODE, int U, bool NT>
__global__ __launch_bounds__(kBThreads) void spmv_band_kernel(
    int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, double alpha, double beta, double *__restrict__ slab,
    const int *__restrict__ done, int cap, int M, const bi4_t *__restrict__ cdesc, const int *__restrict__ bstart) {
  if (done != nullptr && *done != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *xw = reinterpret_cast<double *>(smem_raw);            // M doubles: the ring, slot = column mod M
  double *sprod = reinterpret_cast<double *>(smem_raw) + M;     // cap + 2 doubles: the chunk's products (M is a multiple of 8)
  constexpr int R = kBThreads / G;
  const int tid = threadIdx.x;
  const int g = tid % G, rl = tid / G;
  // blocks of one XCD (equal blockIdx % 8) take neighbouring runs of chunks: their windows overlap in that XCD's L2
  const int lb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int k0 = bstart[lb], k1 = bstart[lb + 1];
  double acc_yx = 0.0, acc_xx = 0.0;
  const unsigned uM = (unsigned)M;
  auto slot_of = [&](int column, int rb_) -> unsigned {   // rb_ <= column < rb_ + 2 M, rb_ a multiple of M
    const unsigned s = (unsigned)(column - rb_);
    return min(s, s - uM);
  };
  // Addresses: a block-uniform base (scalar registers) plus the thread's own 32-bit pair index.  Lanes behind the end of the
  // chunk load nothing (whole waves skip their loads and their part of pass 1).
  auto pair_loads = [&](BandSet<U> &S, int cb_, int ce_) {
    const int vb_ = cb_ & ~1;
    const unsigned npairs = (unsigned)(ce_ - vb_ + 1) >> 1;
    const bd2_t *vbase = reinterpret_cast<const bd2_t *>(val + vb_);
    const bi2_t *cbase = reinterpret_cast<const bi2_t *>(col + vb_);
#pragma unroll
    for (int i = 0; i < U; i++) {
      const unsigned pi = (unsigned)tid + (unsigned)(i * kBThreads);
      if (pi < npairs) {
        S.rv[i] = NT ? __builtin_nontemporal_load(vbase + pi) : vbase[pi];
        S.rc[i] = NT ? __builtin_nontemporal_load(cbase + pi) : cbase[pi];
      }
    }
  };
  if (k0 < k1) {
    BandSet<U> SA, SB;
    // descriptors of chunks j .. j + 3 (d3.x closes chunk j + 2); the table has slack behind its end
    bi4_t d0 = cdesc[k0], d1 = cdesc[k0 + 1], d2 = cdesc[k0 + 2], d3 = cdesc[k0 + 3];
    // row bounds (and the row's x or y) of the lane group: requested one chunk ahead
    int nb_ = 0, ne_ = 0;
    double nxr = 0.0, nyold = 0.0;
    auto row_loads = [&](int chunk_) {
      const int64_t row_ = (int64_t)chunk_ * R + rl;
      nb_ = 0; ne_ = 0; nxr = 0.0; nyold = 0.0;
      if (chunk_ < k1 && row_ < nrows) {
        nb_ = rowptr[row_];
        ne_ = rowptr[row_ + 1];
        if (MODE == 1 && g == 0) nxr = x[row_];
        if (MODE == 0 && g == 0 && beta != 0.0) nyold = y[row_];
      }
    };
    row_loads(k0);
    SA.xn = 0.0;
    SB.xn = 0.0;
    pair_loads(SA, d0.x, d1.x);
    if (k0 + 1 < k1) {
      pair_loads(SB, d1.x, d2.x);
      if (d1.z + tid <= d1.y) SB.xn = x[d1.z + tid];
    }
    // the ring as chunk k0 needs it: columns (H - M, H]
    int rb;
    {
      const int H = d0.y, lowc = max(H + 1 - M, 0);
      rb = (lowc / M) * M;
      for (int o = lowc + tid; o <= H; o += 8 * kBThreads) {
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; q++) t[q] = (o + q * kBThreads <= H) ? x[o + q * kBThreads] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (o + q * kBThreads <= H) xw[slot_of(o + q * kBThreads, rb)] = t[q];
      }
    }
    __syncthreads();
    int j = k0;
    // one chunk: S holds its pairs (and is refilled with chunk j + 2's), T holds chunk j + 1's new ring column
    auto step = [&](BandSet<U> &S, BandSet<U> &T) {
      const int cb = d0.x, ce = d1.x, vb = cb & ~1;
      while (d0.y + 1 - M >= rb + M) rb += M;   // (block-uniform)
      const int64_t row = (int64_t)j * R + rl;
      const bool have_row = row < nrows;
      const int b = nb_ - vb, e = ne_ - vb;
      const double xr = nxr, yold = nyold;
      row_loads(j + 1);
      const bi4_t d4 = cdesc[j + 4];
      // pass 1: products of the staged pairs; an entry outside the chunk (the odd one a 16-byte load drags in) must not
      // index the ring with a column of another window
      bd2_t pr[U];
#pragma unroll
      for (int i = 0; i < U; i++) {
        const int k = vb + (tid + i * kBThreads) * 2;
        pr[i].x = 0.0;
        pr[i].y = 0.0;
        if (k < ce) {
          const unsigned s0 = (k >= cb) ? slot_of(S.rc[i].x, rb) : 0u;
          const unsigned s1 = (k + 1 < ce) ? slot_of(S.rc[i].y, rb) : 0u;
          const double x0 = xw[s0], x1 = xw[s1];
          pr[i].x = (k >= cb) ? S.rv[i].x * x0 : 0.0;
          pr[i].y = (k + 1 < ce) ? S.rv[i].y * x1 : 0.0;
        }
      }
#pragma unroll
      for (int i = 0; i < U; i++) {
        const int k = vb + (tid + i * kBThreads) * 2;
        if (k < ce) *reinterpret_cast<bd2_t *>(sprod + (k - vb)) = pr[i];
      }
      // chunk j + 2: its pairs and its new ring column
      if (j + 2 < k1) {
        pair_loads(S, d2.x, d3.x);
        S.xn = (d2.z + tid <= d2.y) ? x[d2.z + tid] : 0.0;
      }
      __syncthreads();
      // pass 1 of this chunk is over everywhere: the ring moves on to chunk j + 1's window (columns [a, H] of d1)
      if (j + 1 < k1) {
        int rbn = rb;
        while (d1.y + 1 - M >= rbn + M) rbn += M;
        const int o = d1.z + tid;
        if (o <= d1.y) xw[slot_of(o, rbn)] = T.xn;
        for (int o2 = o + kBThreads; o2 <= d1.y; o2 += kBThreads) xw[slot_of(o2, rbn)] = x[o2];   // (a jump of more than 512 columns)
      }
      // pass 2: G lanes per row
      double sum = 0.0;
      {
        const int lastj = max(e - 1, 0);
        double pj[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const int q = b + g + t * G;
          pj[t] = (have_row && q < e) ? sprod[min(q, lastj)] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 8; t++) sum += pj[t];
        for (int q0 = b + g + 8 * G; have_row && q0 < e; q0 += 4 * G) {
          double p4[4];
#pragma unroll
          for (int t = 0; t < 4; t++) p4[t] = (q0 + t * G < e) ? sprod[min(q0 + t * G, lastj)] : 0.0;
#pragma unroll
          for (int t = 0; t < 4; t++) sum += p4[t];
        }
      }
      if (G > 1) sum = group_sum<G>(sum);
      if (g == 0 && have_row) {
        double yv;
        if (MODE == 0) {
          yv = (beta == 0.0) ? alpha * sum : alpha * sum + beta * yold;
        } else {
          yv = sum;
          acc_yx += sum * xr;
          acc_xx += xr * xr;
        }
        if (NT) __builtin_nontemporal_store(yv, y + row); else y[row] = yv;
      }
      __syncthreads();
      d0 = d1; d1 = d2; d2 = d3; d3 = d4;
      j++;
    };
    while (j < k1) {
      step(SA, SB);
      if (j >= k1) break;
      step(SB, SA);
    }
  }
  if (MODE != 0) {
    double *red = reinterpret_cast<double *>(smem_raw);
    double v[2] = {acc_yx, acc_xx};
    block_sum<2, kBThreads>(v, red);
    if (threadIdx.x == 0) {
      slab[0 * kSlabStride + blockIdx.x] = v[0];
      slab[1 * kSlabStride + blockIdx.x] = v[1];
    }
  }
}

template <int MODE, int U, bool NT>
static int launch_band_u(const mk_matrix *A, double alpha, const double *x, double beta, double *y, double *slab, const int *done, int *nblocks) {
  Context &c = ctx();
  const int G = 2 * A->stream_lanes, cap = A->stream_cap, M = A->band_window;
  const size_t lds = ((size_t)M + (size_t)cap + 4) * 8;
  const int blocks = A->band_blocks;
  if (nblocks) *nblocks = blocks;
  if (dry_launch()) return MK_OK;
#define MK_LAUNCH_B(GG)                                                                                                        \
  case GG: {                                                                                                                   \
    auto kern = spmv_band_kernel<GG, MODE, U, NT>;                                                                             \
    static size_t lds_max = 48 * 1024;   /* dynamic LDS above 48 KB: one opt-in per kernel, only ever raised */                \
    if (lds > lds_max) {                                                                                                       \
      MK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      lds_max = lds;                                                                                                           \
    }                                                                                                                          \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kBThreads), lds, c.stream, A->nrows, A->rowptr, A->col, A->val, x, y, \
                       alpha, beta, slab, done, cap, M, reinterpret_cast<const bi4_t *>(A->band_desc), A->band_bstart);        \
  } break;
  switch (G) {
    MK_LAUNCH_B(2) MK_LAUNCH_B(4) MK_LAUNCH_B(8) MK_LAUNCH_B(16) MK_LAUNCH_B(32) MK_LAUNCH_B(64)
    default: return fail(MK_ERR_INVALID, "spmv: band kernel lanes %d invalid", G);
  }
#undef MK_LAUNCH_B
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// mode 0: y = alpha A x + beta y; mode 1: y = A x + partials of <y,x>, <x,x>
int launch_band(const mk_matrix *A, int mode, double alpha, const double *x, double beta, double *y, double *slab, const int *done, int *nblocks) {
  if (!A->band_desc || !A->band_bstart || A->band_rows != kBThreads / (2 * A->stream_lanes) || A->band_blocks < 8 || (A->band_blocks & 7))
    return fail(MK_ERR_UNSUPPORTED, "spmv: the band kernel needs a matrix whose rows stay within a bounded window of x");
  const bool nt = nt_matrix((double)A->nnz * 8.0);
  const int cap = A->stream_cap;
#define MK_BAND_PICK(UU)                                                                                     \
  if (cap <= 1024 * UU - 8) {                                                                                \
    if (mode == 0) return nt ? launch_band_u<0, UU, true>(A, alpha, x, beta, y, slab, done, nblocks)         \
                             : launch_band_u<0, UU, false>(A, alpha, x, beta, y, slab, done, nblocks);       \
    return nt ? launch_band_u<1, UU, true>(A, alpha, x, beta, y, slab, done, nblocks)                        \
              : launch_band_u<1, UU, false>(A, alpha, x, beta, y, slab, done, nblocks);                      \
  }
  MK_BAND_PICK(1)
  MK_BAND_PICK(2)
  MK_BAND_PICK(3)
  MK_BAND_PICK(4)
#undef MK_BAND_PICK
  return fail(MK_ERR_UNSUPPORTED, "spmv: band chunk of %d entries exceeds the LDS stage", cap);
}

}  // namespace mk
