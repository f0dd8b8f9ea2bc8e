// band.hip -- CSR SpMV with the x window of a row panel staged in LDS (band kernel).
//
// Stands in for hypre_ParCSRMatrixMatvec on matrices whose rows reach a bounded distance from the diagonal but share
// neither offsets (no pattern table) nor column lists (no row groups): finite-element matrices in a bandwidth-reducing
// order.  The stream kernel gathers x per entry from global memory, and what bounds it there is the number of L1 -> L2
// requests of those gathers, not HBM (0.34-0.46 of the roofline on tools/fem_probe.py's `rand` cases).  Here
//   panel  = RP consecutive rows whose columns span at most W entries of x (W + stage <= half a CU's LDS);
//   window = x[wb, wb + wl) of the panel, copied into LDS once per panel by coalesced loads;
//   chunk  = 512 / G consecutive rows of the panel; pass 1 reads the chunk's (value, column) pairs with coalesced 16- and
//            8-byte loads (requested one chunk ahead), multiplies every value with its x from the LDS window and leaves the
//            products in LDS; pass 2 gives every row G lanes that add the row's products (lane g: entries g, g + G, ...,
//            then a shuffle tree: the stream kernel's order for the same G).
// No gather leaves the CU.  HBM bytes: 12 per entry + 20 per row + the windows (W / RP x 8 B per row, served by the L2
// for all but the first panel that touches a line).  Results agree with the serial reference to 1e-12 (tests), and are
// the same from run to run (no atomics, fixed summation order).
// Algorithmic bytes: SURVEY 8(d)'s 12 nnz + 20 n + 4.
#include "mk_internal.h"

namespace mk {

constexpr int kBThreads = 512;
constexpr int kBandBase = 256;   // rows per base panel of the span table (panels are 1, 2, 4, ... of these)
typedef double bd2_t __attribute__((ext_vector_type(2)));
typedef int bi2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------- the plan
// spans[p] = (smallest, largest) column of base panel p (rows [256 p, 256 (p + 1))); (INT_MAX, -1) when it has no entry
__global__ __launch_bounds__(256) void band_span_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                        bi2_t *__restrict__ spans, int npanels) {
  __shared__ int smin[4], smax[4];
  for (int p = blockIdx.x; p < npanels; p += gridDim.x) {
    const int64_t r0 = (int64_t)p * kBandBase, r1 = min(r0 + kBandBase, nrows);
    const int b = rowptr[r0], e = rowptr[r1];
    int lo = 0x7fffffff, hi = -1;
    for (int j = b + threadIdx.x; j < e; j += 256) {
      const int cj = col[j];
      lo = min(lo, cj);
      hi = max(hi, cj);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo = min(lo, __shfl_down(lo, off, 64));
      hi = max(hi, __shfl_down(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      bi2_t s;
      s.x = min(min(smin[0], smin[1]), min(smin[2], smin[3]));
      s.y = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
      spans[p] = s;
    }
    __syncthreads();
  }
}

static int band_lds_budget() {   // bytes of LDS per block: half a CU's 160 KB, so that two blocks of 512 threads share a CU
  const int v = getenv("MK_BAND_LDS") ? std::max(16384, std::min(atoi(getenv("MK_BAND_LDS")), 160 * 1024 - 512)) : 80 * 1024;
  return v;
}

int build_band_plan(mk_matrix *A) {
  // (read per matrix, not once per process: the analysis runs once per matrix, and tests and probes switch per matrix)
  const int off = getenv("MK_NO_BAND") ? atoi(getenv("MK_NO_BAND")) : 0;
  const double min_len = getenv("MK_BAND_MIN_LEN") ? atof(getenv("MK_BAND_MIN_LEN")) : 16.0;
  const int64_t min_rows = getenv("MK_BAND_MIN_ROWS") ? atoll(getenv("MK_BAND_MIN_ROWS")) : 100000;
  if (off || A->pat || A->grow || A->band_span || A->nrows < min_rows || A->nnz < 1 || A->info.mean_row_len < min_len) return MK_OK;
  if (A->stream_lanes < 1 || A->stream_lanes > 32) return MK_OK;
  Context &c = ctx();
  const int nbase = (int)((A->nrows + kBandBase - 1) / kBandBase);
  bi2_t *d_spans = nullptr;
  if (dev_alloc(reinterpret_cast<int **>(&d_spans), (size_t)nbase * 2) != MK_OK) return MK_OK;   // an optimisation only
  hipLaunchKernelGGL(band_span_kernel, dim3(std::min(nbase, 8192)), dim3(256), 0, c.stream, A->nrows, A->rowptr, A->col, d_spans, nbase);
  std::vector<int> h((size_t)nbase * 2);
  hipError_t e = hipMemcpyAsync(h.data(), d_spans, sizeof(int) * 2 * (size_t)nbase, hipMemcpyDeviceToHost, c.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  dev_free(reinterpret_cast<int *>(d_spans));
  if (e != hipSuccess) return fail(MK_ERR_HIP, "band plan: %s", hipGetErrorString(e));
  // the largest panel (256 x 2^k rows, at most 4096) whose widest window fits beside the product stage; a window of more
  // than 8 entries per row of the panel would cost more L2 traffic than it saves requests
  const int cap = A->stream_cap;   // entries of the fullest chunk of 256 / stream_lanes = 512 / G rows
  const int budget = band_lds_budget();
  for (int m = 16; m >= 1; m /= 2) {
    if (m > 1 && nbase < 2 * m) continue;
    const int np = (nbase + m - 1) / m;
    int64_t wmax = 0;
    for (int p = 0; p < np; p++) {
      int lo = 0x7fffffff, hi = -1;
      for (int q = p * m; q < std::min(nbase, (p + 1) * m); q++) { lo = std::min(lo, h[2 * q]); hi = std::max(hi, h[2 * q + 1]); }
      if (hi >= lo) wmax = std::max<int64_t>(wmax, (int64_t)hi - (lo & ~1) + 1);
    }
    const int64_t lds = (wmax + 4 + cap + 2) * 8;
    if (wmax < 1 || lds > budget || wmax > (int64_t)8 * m * kBandBase) continue;
    std::vector<int> tab((size_t)np * 2 + 8, 0);
    for (int p = 0; p < np; p++) {
      int lo = 0x7fffffff, hi = -1;
      for (int q = p * m; q < std::min(nbase, (p + 1) * m); q++) { lo = std::min(lo, h[2 * q]); hi = std::max(hi, h[2 * q + 1]); }
      if (hi >= lo) { tab[2 * p] = lo & ~1; tab[2 * p + 1] = hi - (lo & ~1) + 1; }
      if (hi >= lo && (lo < 0 || hi >= A->ncols)) return MK_OK;   // (column indices are validated at assembly; never expected)
    }
    if (dev_alloc(&A->band_span, tab.size()) != MK_OK) return MK_OK;
    MK_HIP(hipMemcpyAsync(A->band_span, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, c.stream));
    MK_HIP(hipStreamSynchronize(c.stream));
    A->band_rows = m * kBandBase;
    A->band_panels = np;
    A->band_window = (int)wmax;
    return MK_OK;
  }
  return MK_OK;
}

// ------------------------------------------------------------------------------------------ the kernel
// MODE 0: y = alpha A x + beta y.  MODE 1: y = A x, per-block partials of <y,x>, <x,x> (the CG step's K1).
// U: capacity class of the product stage (cap <= 1024 U - 8): a thread holds U (value pair, column pair) loads per chunk.
template <int G, int MODE, int U, bool NT>
__global__ __launch_bounds__(kBThreads) void spmv_band_kernel(
    int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, double alpha, double beta, double *__restrict__ slab,
    const int *__restrict__ done, int cap, int wcap, int rp, int npanels, const bi2_t *__restrict__ spans) {
  if (done != nullptr && *done != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *xw = reinterpret_cast<double *>(smem_raw);                            // wcap + 2 doubles: the panel's window of x
  double *sprod = reinterpret_cast<double *>(smem_raw) + ((wcap + 2 + 1) & ~1);   // cap + 2 doubles: the chunk's products
  constexpr int R = kBThreads / G;
  const int tid = threadIdx.x;
  const int g = tid % G, rl = tid / G;
  // blocks of one XCD (equal blockIdx % 8) take a contiguous range of panels, interleaved among themselves: the windows of
  // neighbouring panels overlap, and that XCD's L2 serves the overlap
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int per_xcd = (npanels + 7) / 8;
  const int first = xcd * per_xcd, last = min(first + per_xcd, npanels);
  const int cpp = rp / R;   // chunks per panel
  double acc_yx = 0.0, acc_xx = 0.0;
  bd2_t rv[U];
  bi2_t rc[U];
  auto stage_loads = [&](int cb_, int ce_) {
    const int vb_ = cb_ & ~1;
#pragma unroll
    for (int i = 0; i < U; i++) {
      const int k = vb_ + (tid + i * kBThreads) * 2;
      if (k < ce_) {
        rv[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const bd2_t *>(val + k)) : *reinterpret_cast<const bd2_t *>(val + k);
        rc[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const bi2_t *>(col + k)) : *reinterpret_cast<const bi2_t *>(col + k);
      }
    }
  };
  int panel = first + slot, ch = 0;
  int64_t r0 = 0, r1 = 0;
  int cb = 0, ce = 0;
  auto chunk_rows = [&](int panel_, int ch_, int64_t &r0_, int64_t &r1_) {
    r0_ = (int64_t)panel_ * rp + (int64_t)ch_ * R;
    r1_ = min(min(r0_ + R, (int64_t)(panel_ + 1) * rp), nrows);
  };
  if (panel < last) {
    chunk_rows(panel, 0, r0, r1);
    cb = rowptr[r0];
    ce = rowptr[r1];
    stage_loads(cb, ce);
  }
  int staged = -1;   // panel whose window is in LDS
  int wb = 0;
  while (panel < last) {
    if (staged != panel) {   // (block-uniform) new panel: its window of x
      const bi2_t sp = spans[panel];
      wb = sp.x;
      const int wl = sp.y;
      const double *xs = x + wb;
      for (int o = tid; o < wl; o += 4 * kBThreads) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        t0 = xs[o];
        if (o + kBThreads < wl) t1 = xs[o + kBThreads];
        if (o + 2 * kBThreads < wl) t2 = xs[o + 2 * kBThreads];
        if (o + 3 * kBThreads < wl) t3 = xs[o + 3 * kBThreads];
        xw[o] = t0;
        if (o + kBThreads < wl) xw[o + kBThreads] = t1;
        if (o + 2 * kBThreads < wl) xw[o + 2 * kBThreads] = t2;
        if (o + 3 * kBThreads < wl) xw[o + 3 * kBThreads] = t3;
      }
      staged = panel;
      __syncthreads();
    }
    const int vb = cb & ~1;
    // the row this lane group sums in pass 2: its bounds are requested now and used behind the barrier
    const int64_t row = r0 + rl;
    const bool have_row = row < r1;
    int b = 0, e = 0;
    if (have_row) { b = rowptr[row] - vb; e = rowptr[row + 1] - vb; }
    const double xr = (MODE == 1 && g == 0 && have_row) ? x[row] : 0.0;
    const double yold = (MODE == 0 && g == 0 && have_row && beta != 0.0) ? y[row] : 0.0;
    // pass 1: products of the staged pairs; an entry in front of the chunk (the odd one a 16-byte load drags in) belongs to
    // another window and must not index this one
    bd2_t pr[U];
#pragma unroll
    for (int i = 0; i < U; i++) {
      const int k = vb + (tid + i * kBThreads) * 2;
      pr[i].x = 0.0;
      pr[i].y = 0.0;
      if (k < ce) {
        const int c0 = (k >= cb) ? rc[i].x - wb : 0;
        const int c1 = (k + 1 < ce) ? rc[i].y - wb : 0;
        const double x0 = xw[c0], x1 = xw[c1];
        pr[i].x = (k >= cb) ? rv[i].x * x0 : 0.0;
        pr[i].y = (k + 1 < ce) ? rv[i].y * x1 : 0.0;
      }
    }
    // the next chunk's pairs: requested now, in flight through pass 2 and both barriers
    int npanel = panel, nch = ch + 1;
    {
      int64_t t0, t1;
      chunk_rows(panel, nch, t0, t1);
      if (nch >= cpp || t0 >= nrows) { npanel = panel + slots; nch = 0; }
    }
    int64_t nr0 = 0, nr1 = 0;
    int ncb = 0, nce = 0;
    if (npanel < last) {
      chunk_rows(npanel, nch, nr0, nr1);
      ncb = rowptr[nr0];
      nce = rowptr[nr1];
      stage_loads(ncb, nce);
    }
#pragma unroll
    for (int i = 0; i < U; i++) {
      const int k = vb + (tid + i * kBThreads) * 2;
      if (k < ce) *reinterpret_cast<bd2_t *>(sprod + (k - vb)) = pr[i];
    }
    __syncthreads();
    // pass 2: G lanes per row
    double sum = 0.0;
    {
      const int lastj = max(e - 1, 0);
      double pj[8];
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const int j = b + g + t * G;
        pj[t] = (have_row && j < e) ? sprod[min(j, lastj)] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < 8; t++) sum += pj[t];
      for (int j0 = b + g + 8 * G; have_row && j0 < e; j0 += 4 * G) {
        double p4[4];
#pragma unroll
        for (int t = 0; t < 4; t++) p4[t] = (j0 + t * G < e) ? sprod[min(j0 + t * G, lastj)] : 0.0;
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
    panel = npanel; ch = nch; r0 = nr0; r1 = nr1; cb = ncb; ce = nce;
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
  const int G = 2 * A->stream_lanes, cap = A->stream_cap, wcap = A->band_window;
  const size_t lds = ((size_t)((wcap + 2 + 1) & ~1) + (size_t)cap + 2) * 8;
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  int64_t blocks = std::min<int64_t>(((int64_t)A->band_panels + 7) / 8 * 8, (int64_t)c.num_cus * per_cu);
  blocks = std::max<int64_t>(8, blocks / 8 * 8);
  if (blocks > kSlabStride) blocks = kSlabStride;
  if (nblocks) *nblocks = (int)blocks;
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
                       alpha, beta, slab, done, cap, wcap, A->band_rows, A->band_panels,                                       \
                       reinterpret_cast<const bi2_t *>(A->band_span));                                                         \
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
  if (!A->band_span || A->band_rows < kBandBase || A->band_rows % (kBThreads / (2 * A->stream_lanes)) != 0)
    return fail(MK_ERR_UNSUPPORTED, "spmv: the band kernel needs a matrix whose row panels reach a bounded window of x");
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
