// spmv.hip -- fp64 CSR SpMV kernels for gfx950 and the per-matrix selector.
//
// Replaces hypre_ParCSRMatrixMatvec (external to the reference; in-repo stand-in
// test/sblas_mv_rd.c:51-69).  Bandwidth-bound: 12 B per stored entry + 20 B per row
// (SURVEY.md 8d), 2 flops per entry -> no MFMA; everything here is about keeping enough
// independent 64-bit loads in flight and about coalescing.
//
//   vector kernel : 2^k lanes of a wave64 share one row (k from the mean row length; k=6 is
//                   wavefront-per-row).  Lanes of a group read consecutive entries, groups
//                   read consecutive rows, so a wave's loads of val/col are one contiguous
//                   span.  Used for short, regular rows (stencils, banded).
//   stream kernel : nnz-balanced chunks staged through LDS, rows split inside the chunk;
//                   rows longer than a chunk are split over blocks (skewed / power-law rows).
#include <cfloat>
#include <map>

#include "mk_internal.h"

namespace mk {

constexpr int kThreads = 256;
// LDS entries per block (12 B each, 48 KB at most -> three blocks per CU); MK_STREAM_MAX_CAP lowers it (3064: the limit until r02).
// A chunk is staged from the last cache line boundary in front of its first value / column (kStreamAlignV / kStreamAlignI
// entries = 128 bytes), so the stage holds up to 15 values and 31 columns more than the chunk has: 4064 instead of 4088.
// (Two streams read side by side run at 7.06 TB/s from aligned starts and at 5.41 TB/s from starts 40 / 20 bytes off a line:
// tools/microbench/read_stream.hip.)
constexpr int kStreamAlignV = 16, kStreamAlignI = 32;
static int stream_max_cap() {
  static const int v = getenv("MK_STREAM_MAX_CAP") ? std::max(64, std::min(atoi(getenv("MK_STREAM_MAX_CAP")), 4064)) : 4064;
  return v;
}
#define kStreamMaxCap stream_max_cap()

// ------------------------------------------------------------------ matrix analysis
struct AnalysisOut {
  unsigned long long hist[16];
  int max_len;
  int min_len;
  int chunk_max[7];   // most entries in any run of 256 >> k consecutive rows (the stream kernel's chunk with 1 << k lanes per row)
};

__global__ void analyse_rows_kernel(int64_t nrows, const int *__restrict__ rowptr, AnalysisOut *out) {
  __shared__ unsigned int sh_hist[16];
  __shared__ int sh_max, sh_min;
  if (threadIdx.x < 16) sh_hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) { sh_max = 0; sh_min = 0x7fffffff; }
  __syncthreads();
  int lmax = 0, lmin = 0x7fffffff;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    const int len = rowptr[r + 1] - rowptr[r];
    int b = (len <= 0) ? 0 : (32 - __clz(len));  // len in [2^(b-1), 2^b)
    if (b > 15) b = 15;
    atomicAdd(&sh_hist[b], 1u);
    lmax = max(lmax, len);
    lmin = min(lmin, len);
  }
  atomicMax(&sh_max, lmax);
  atomicMin(&sh_min, lmin);
  __syncthreads();
  if (threadIdx.x < 16 && sh_hist[threadIdx.x]) atomicAdd(&out->hist[threadIdx.x], (unsigned long long)sh_hist[threadIdx.x]);
  if (threadIdx.x == 0) { atomicMax(&out->max_len, sh_max); atomicMin(&out->min_len, sh_min); }
}

__global__ void analyse_chunks_kernel(int64_t nrows, const int *__restrict__ rowptr, AnalysisOut *out) {
  __shared__ int sh[7];
  if (threadIdx.x < 7) sh[threadIdx.x] = 0;
  __syncthreads();
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, total = (int64_t)gridDim.x * blockDim.x;
  for (int k = 0; k < 7; k++) {
    const int R = 256 >> k;
    const int64_t nch = (nrows + R - 1) / R;
    int lmax = 0;
    for (int64_t c = gid; c < nch; c += total) lmax = max(lmax, rowptr[min((c + 1) * R, nrows)] - rowptr[c * R]);
    if (lmax > 0) atomicMax(&sh[k], lmax);
  }
  __syncthreads();
  if (threadIdx.x < 7 && sh[threadIdx.x] > 0) atomicMax(&out->chunk_max[threadIdx.x], sh[threadIdx.x]);
}

__global__ void diag_inverse_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                    const double *__restrict__ val, double *__restrict__ dinv) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    double d = 0.0;
    for (int j = rowptr[r]; j < rowptr[r + 1]; j++)
      if (col[j] == (int)r) { d = val[j]; break; }
    dinv[r] = 1.0 / d;  // HYPRE_ParCSRDiagScale divides by the diagonal as stored
  }
}

// --------------------------------------------------------- column-index compression
// Lossless: the product reads exactly the same (column, value) pairs in the same order.
constexpr int kPatSlots = 1 << 16;
constexpr int kPatMaxLen = 128;     // longer rows are not looked at (3 dof per node on a 27-point grid: 81 entries, 81 patterns)
static int pat_max_len() {
  static const int v = getenv("MK_PAT_MAX_LEN") ? atoi(getenv("MK_PAT_MAX_LEN")) : kPatMaxLen;
  return v;
}
constexpr int kPatMaxProbes = 512;
constexpr int kPatMaxInts = 8192;   // table must fit 32 KB of LDS beside the value stage

// VAL: the row's values (bit patterns) are part of the key, so rows that share a table entry carry
// the same coefficients as well (constant-coefficient stencils, banded Toeplitz-like operators)
template <bool VAL>
__device__ __forceinline__ unsigned long long pat_hash(int64_t row, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                       const double *__restrict__ val) {
  unsigned long long h = 0xCBF29CE484222325ULL;
  const int b = rowptr[row], e = rowptr[row + 1];
  h = (h ^ (unsigned long long)(unsigned)(e - b)) * 0x100000001B3ULL;
  for (int k = b; k < e; k++) h = (h ^ (unsigned long long)(unsigned)(col[k] - (int)row)) * 0x100000001B3ULL;
  if (VAL)
    for (int k = b; k < e; k++) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(val[k]);
      h = (h ^ (bits & 0xffffffffULL)) * 0x100000001B3ULL;
      h = (h ^ (bits >> 32)) * 0x100000001B3ULL;
    }
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
  return h | 1ULL;   // 0 marks an empty slot
}

template <bool VAL>
__global__ void pat_insert_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                  const double *__restrict__ val, unsigned long long *__restrict__ keys,
                                  int *__restrict__ slot_of_row, int *__restrict__ fail) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long h = pat_hash<VAL>(r, rowptr, col, val);
    unsigned s = (unsigned)(h >> 40) & (kPatSlots - 1);
    // a usable table holds at most kPatMaxInts/2 = 4096 patterns (6 % of the slots), so a long probe
    // sequence means "too many patterns": give up at once, and let every other row give up with it
    // (a matrix of all-distinct rows would otherwise scan the full table once per row)
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    int probes = 0;
    for (; probes < kPatMaxProbes; probes++) {
      // look before the atomic: with a few dozen patterns shared by millions of rows nearly every
      // row finds its key already there (134M contended CAS on 27 words took 0.85 s)
      unsigned long long old = __hip_atomic_load(&keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == 0ULL) old = atomicCAS(&keys[s], 0ULL, h);
      if (old == 0ULL || old == h) break;
      s = (s + 1) & (kPatSlots - 1);
    }
    if (probes == kPatMaxProbes) { *fail = 1; return; }
    slot_of_row[r] = (int)s;
  }
}

// dense ids in slot order (deterministic whatever the insertion order was); one block
__global__ __launch_bounds__(1024) void pat_compact_kernel(const unsigned long long *__restrict__ keys, int *__restrict__ id_of_slot,
                                                           int *__restrict__ npat) {
  __shared__ int sums[1024];
  constexpr int PER = kPatSlots / 1024;
  const int base = threadIdx.x * PER;
  int cnt = 0;
  for (int i = 0; i < PER; i++) cnt += keys[base + i] != 0ULL;
  sums[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int t = (threadIdx.x >= (unsigned)off) ? sums[threadIdx.x - off] : 0;
    __syncthreads();
    sums[threadIdx.x] += t;
    __syncthreads();
  }
  int run = sums[threadIdx.x] - cnt;
  for (int i = 0; i < PER; i++) {
    id_of_slot[base + i] = run;
    run += keys[base + i] != 0ULL;
  }
  if (threadIdx.x == 1023) *npat = sums[1023];
}

template <bool VAL>
__global__ void pat_assign_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                  const double *__restrict__ val, const int *__restrict__ slot_of_row,
                                  const int *__restrict__ id_of_slot, unsigned short *__restrict__ pat, int *__restrict__ ptab,
                                  double *__restrict__ vtab, int pml, int *__restrict__ written) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    const int id = id_of_slot[slot_of_row[r]];
    pat[r] = (unsigned short)id;
    if (__hip_atomic_load(&written[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && atomicExch(&written[id], 1) == 0) {
      const int b = rowptr[r], e = rowptr[r + 1];
      int *t = ptab + (size_t)id * (pml + 1);
      t[0] = e - b;
      for (int k = b; k < e; k++) t[1 + k - b] = col[k] - (int)r;
      for (int k = e - b; k < pml; k++) t[1 + k] = 0;
      if (VAL) {
        double *tv = vtab + (size_t)id * pml;
        for (int k = b; k < e; k++) tv[k - b] = val[k];
        for (int k = e - b; k < pml; k++) tv[k] = 0.0;
      }
    }
  }
}

// hash collisions would silently change the matrix: compare every row with its table entry
template <bool VAL>
__global__ void pat_verify_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                  const double *__restrict__ val, const unsigned short *__restrict__ pat,
                                  const int *__restrict__ ptab, const double *__restrict__ vtab, int pml, int *__restrict__ fail) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    const int *t = ptab + (size_t)pat[r] * (pml + 1);
    const int b = rowptr[r], e = rowptr[r + 1];
    bool ok = (t[0] == e - b);
    for (int k = b; ok && k < e; k++) ok = (t[1 + k - b] == col[k] - (int)r);
    if (VAL) {
      const double *tv = vtab + (size_t)pat[r] * pml;
      for (int k = b; ok && k < e; k++) ok = (__double_as_longlong(tv[k - b]) == __double_as_longlong(val[k]));   // bits, not ==: -0.0 and NaNs too
    }
    if (!ok) *fail = 1;
  }
}

constexpr int kPatMaxValBytes = 32768;
constexpr int kPatternChunkRows = 2 * kThreads;   // pattern kernel: rows per chunk   // value table (doubles) beside the offset table in LDS

// rows per chunk of the sweep kernel = two per lane: 512 with 256 threads, or 1024 with 512 threads when the halo
// (wlo + whi) is at least 512 entries -- the window is then 2x the chunk instead of 3x, and the L2 -> LDS fill,
// which bounds these kernels at ~9 TB/s summed over the CUs, shrinks by a third
constexpr int kSweepNS = 4;             // ring slots: planes k-1, k, k+1 and the one being written
constexpr int kSweepMaxSw = 2048;        // window entries per slab (LDS = 4 slabs x 8 B x this = 64 KB)

// Can the sweep kernel serve this matrix?  tab/vals: the (offsets, values) pattern table.
// with_values = false: only the offsets come from the table, the values are streamed from val[] (spmv_vsweep_kernel,
// vsweep.hip): patterns are matched on offsets alone and the chunk length is chosen so that the LDS stage of one
// plane-step's values fits beside the window ring.
static int plan_sweep(mk_matrix *A, const std::vector<int> &tab, const std::vector<double> &vals, bool with_values = true) {
  static const int off = getenv("MK_NO_SWEEP") ? atoi(getenv("MK_NO_SWEEP")) : 0;
  const int npat = A->npat, pml = A->pml;
  if (off || pml > 32 || (A->nrows + 8) * 8 >= (int64_t)1 << 32 || A->ncols * 8 >= (int64_t)1 << 32) return MK_OK;
  auto len_of = [&](int q) { return tab[(size_t)q * (pml + 1)]; };
  auto off_of = [&](int q, int k) { return tab[(size_t)q * (pml + 1) + 1 + k]; };
  auto bits_of = [&](int q, int k) { unsigned long long u = 0; if (with_values) memcpy(&u, &vals[(size_t)q * pml + k], 8); return u; };
  int base = 0;
  for (int q = 1; q < npat; q++) if (len_of(q) > len_of(base)) base = q;
  const int len = len_of(base);
  if (len < 1) return MK_OK;
  // every pattern must be a sub-list (same order, same offsets, same value bits) of the base
  std::vector<unsigned> mask(npat, 0u);
  for (int q = 0; q < npat; q++) {
    int u = 0;
    for (int k = 0; k < len_of(q); k++) {
      while (u < len && !(off_of(base, u) == off_of(q, k) && bits_of(base, u) == bits_of(q, k))) u++;
      if (u == len) return MK_OK;
      mask[q] |= 1u << u;
      u++;
    }
  }
  SweepArgs S;
  memset(&S, 0, sizeof(S));
  int ea[32] = {0}, eb[32] = {0};
  double ev[32] = {0.0};
  S.len = len;
  S.rpl = 2;
  int maxabs = 0;
  for (int u = 0; u < len; u++) maxabs = std::max(maxabs, std::abs(off_of(base, u)));
  // streamed values leave less LDS for the window: offsets beyond 256 are taken as plane offsets there
  const bool contiguous = (maxabs <= (with_values ? 768 : 256));
  if (contiguous) {   // banded: contiguous sweep, every offset is "near"; P = chunk size, set below
    for (int u = 0; u < len; u++) { ea[u] = 0; eb[u] = off_of(base, u); }
  } else {
    // far offsets cluster around +-P: P = centre of the cluster that holds the largest |offset|
    int lo = maxabs, hi = maxabs;
    for (int u = 0; u < len; u++) {
      const int o = std::abs(off_of(base, u));
      if (2 * o > maxabs) lo = std::min(lo, o);
    }
    // any even stride near the centre will do (b absorbs the difference); a grid gives lo + hi = 2 * plane exactly,
    // and an odd plane cannot be used: a lane's two rows would straddle planes
    if (lo == hi && (lo & 1)) return MK_OK;
    const int P = ((lo + hi) / 2) & ~1;
    if (P < 2) return MK_OK;
    S.P = P;
    for (int u = 0; u < len; u++) {
      const int o = off_of(base, u);
      const int a = (o > P / 2) ? 1 : (o < -(P / 2)) ? -1 : 0;
      ea[u] = a;
      eb[u] = o - a * P;
    }
  }
  int bmin = 0, bmax = 0;
  S.diag = -1;
  for (int u = 0; u < len; u++) {
    bmin = std::min(bmin, eb[u]);
    bmax = std::max(bmax, eb[u]);
    ev[u] = with_values ? vals[(size_t)base * pml + u] : 0.0;
    if (off_of(base, u) == 0) { S.diag = u; S.dv = ev[u]; }
  }
  S.wlo = (-bmin + 1) & ~1;
  const int whi = (bmax + 1) & ~1;
  static const int env_c = getenv("MK_SWEEP_C") ? atoi(getenv("MK_SWEEP_C")) : 0;
  // rows per chunk: the window a block loads per plane is C + halo, so wide halos want long chunks.  Measured CG,
  // ns per row, C = 1024 against 2048: 1024-wide planes 0.0228 / 0.0131, 640-wide 0.0169 / 0.0157, 768-wide equal,
  // 512-wide 0.0124 / 0.0129.
  const int halo = S.wlo + whi;
  S.C = (env_c == 512 || env_c == 1024 || env_c == 2048) ? env_c : (halo >= 1200 ? 2048 : halo >= 512 ? 1024 : 512);
  if (S.C == 2048 && S.C + S.wlo + whi > 4096) S.C = 1024;
  if (S.C == 1024 && S.C + S.wlo + whi > kSweepMaxSw) S.C = 512;
  if (!with_values) {
    // values streamed: LDS = ring (4 slabs of C + halo entries) + one plane-step of values (len * C entries); chunks of
    // 512 rows run two blocks per CU, chunks of 1024 one.  MK_VSWEEP_C picks for A/B runs.
    static const int env_vc = getenv("MK_VSWEEP_C") ? atoi(getenv("MK_VSWEEP_C")) : 0;
    auto fits = [&](int C) {
      const int sw = C + halo;
      const size_t lds = (size_t)kSweepNS * sw * 8 + ((size_t)len * C + 8) * 8 + (size_t)npat * 4;
      // (value positions are 32-bit: vb + 2 (tid + i T) must not wrap)
      return len <= 8 && lds <= 158 * 1024 && sw <= (C == 512 ? 1536 : 3072) && A->nnz < ((int64_t)1 << 31) - 65536;
    };
    // measured (7-pt, product + dots): 512^3 C = 1024 1.955 ms / C = 512 2.05 ms; 256^3 (64 chunks of 1024 per plane)
    // 0.260 / 0.249 ms: the long chunk (window 2x instead of 3x the chunk) wins once a plane fills the chip with them
    // rows of 9 to 28 entries (27-point operators): one row per lane, the stage holds exactly len * C values, the ring three
    // slabs.  Chunks of 512 rows, one block of 8 waves per CU (27 points on 512-wide planes: 110.6 + 37 KB), when that fits;
    // else chunks of 256 rows (wider planes).  Measured, 27-point 256^3, product + dots: C = 512 0.815 ms, C = 256 (two
    // blocks per CU) 0.825 ms, stream kernel 0.896 ms.
    auto lds_long = [&](int C) { return (size_t)3 * (C + halo) * 8 + ((size_t)len * C + 8) * 8 + (size_t)npat * 4; };
    static const int no_long = getenv("MK_NO_VSWEEP_LONG") ? atoi(getenv("MK_NO_VSWEEP_LONG")) : 0;
    static const int env_lc = getenv("MK_VSWEEP_LONG_C") ? atoi(getenv("MK_VSWEEP_LONG_C")) : 0;   // A/B: 256 or 512
    if (len > 8) {
      if (no_long || len > 28 || A->nnz >= ((int64_t)1 << 31) - 65536) return MK_OK;
      const bool ok256 = 256 + halo <= 1536 && lds_long(256) <= 160 * 1024 - 1024;
      const bool two256 = 256 + halo <= 1536 && 2 * (lds_long(256) + 512) <= 160 * 1024;
      const bool ok512 = 512 + halo <= 2048 && lds_long(512) <= 160 * 1024 - 1024;
      if (env_lc == 256 && ok256) S.C = 256;
      else if (env_lc == 512 && ok512) S.C = 512;
      else if (ok512) S.C = 512;
      else if (two256 || ok256) S.C = 256;
      else return MK_OK;
      S.rpl = 1;
    }
    else if ((env_vc == 512 || env_vc == 1024) && fits(env_vc)) S.C = env_vc;
    else if (fits(1024) && S.P / 1024 >= ctx().num_cus && halo >= 512) S.C = 1024;
    else if (fits(512)) S.C = 512;
    else if (fits(1024)) S.C = 1024;
    else return MK_OK;
  }
  S.sw = S.C + S.wlo + whi;
  if (with_values && S.sw > (S.C == 2048 ? 4096 : kSweepMaxSw)) return MK_OK;   // (streamed: fits() has checked)
  if (contiguous) {
    // Banded: a "plane" is G consecutive chunks, so that the blocks of one segment work side by side on one contiguous
    // region (with P = C every block streamed its own range, all ranges the same distance apart).
    static const int env_g = getenv("MK_SWEEP_G") ? atoi(getenv("MK_SWEEP_G")) : 0;
    int G = env_g > 0 ? env_g : 64;   // 2^26 rows, Jacobi sweep: G = 1 352 us, 16 333, 32 323, 64 319, 256 325, 1024 336
    while (G > 1 && (int64_t)S.C * G * 8 > A->nrows) G >>= 1;
    S.P = S.C * std::max(G, 1);
  }
  S.K = (int)((A->nrows + S.P - 1) / S.P);
  S.kb = 0;
  S.ncp = (S.P + S.C - 1) / S.C;
  S.nseg = 1;
  MK_TRY(dev_alloc(&A->pmask, (size_t)npat));
  MK_TRY(dev_alloc(&A->etab, 4 * 32));
  MK_HIP(hipMemcpyAsync(A->pmask, mask.data(), sizeof(unsigned) * npat, hipMemcpyHostToDevice, ctx().stream));
  SweepEntry E[4 * 32];
  memset(E, 0, sizeof(E));
  for (int c4 = 0; c4 < 4; c4++)
    for (int u = 0; u < len; u++) {
      // (long-row plans of the vsweep kernel run a ring of three slabs)
      const int slot0 = (S.rpl == 1 ? (c4 + ea[u] + 3) % 3 : ((c4 + ea[u]) & 3)) * S.sw, q = S.wlo + eb[u];
      if (with_values) {
        // spmv_sweep_kernel keeps a slab de-interleaved (even window entries in the first half, odd in the second, see
        // write_slab): off = where the lane finds the operand of its first row, off1 = of its second row
        const int half = S.sw / 2;
        E[c4 * 32 + u].off = (q & 1) ? (slot0 + half + (q - 1) / 2) * 8 : (slot0 + q / 2) * 8;
        E[c4 * 32 + u].off1 = (q & 1) ? (slot0 + (q + 1) / 2) * 8 : (slot0 + half + q / 2) * 8;
      } else {
        E[c4 * 32 + u].off = (slot0 + q) * 8;   // spmv_vsweep_kernel: window entries in order
      }
      E[c4 * 32 + u].v = ev[u];
    }
  MK_HIP(hipMemcpyAsync(A->etab, E, sizeof(E), hipMemcpyHostToDevice, ctx().stream));
  MK_HIP(hipStreamSynchronize(ctx().stream));
  A->sweep = new SweepArgs(S);
  A->sweep_streamed = !with_values;
  // Long rows (27-point: the kernel is bound by vector issue, and the selects are most of it): deal the row pairs of a chunk
  // so that those with incomplete rows (grid boundaries) come first.  The pairs of one plane in the middle of the grid
  // decide; whether a wave really owns complete rows only is tested per step in the kernel.
  static const int no_perm = getenv("MK_SWEEP_NO_PERM") ? atoi(getenv("MK_SWEEP_NO_PERM")) : 0;
  if (with_values && !contiguous && len > 8 && S.K >= 3 && !no_perm) {
    const int T = S.C / 2;
    const int64_t row0 = (int64_t)(S.K / 2) * S.P;
    const int64_t nrow = std::min<int64_t>(S.P, A->nrows - row0);
    std::vector<unsigned short> ids((size_t)std::max<int64_t>(nrow, 1));
    MK_HIP(hipMemcpyAsync(ids.data(), A->pat + row0, sizeof(unsigned short) * (size_t)nrow, hipMemcpyDeviceToHost, ctx().stream));
    MK_HIP(hipStreamSynchronize(ctx().stream));
    const unsigned fullm = len >= 32 ? 0xffffffffu : (1u << len) - 1u;
    std::vector<unsigned short> perm((size_t)S.ncp * T);
    for (int cp = 0; cp < S.ncp; cp++) {
      int w = 0;
      for (int pass = 0; pass < 2; pass++)
        for (int q = 0; q < T; q++) {
          const int64_t r = (int64_t)cp * S.C + 2 * q;
          bool partial = false;
          if (r < nrow) partial = mask[ids[r]] != fullm || (r + 1 < nrow && mask[ids[r + 1]] != fullm);
          if (partial == (pass == 0)) perm[(size_t)cp * T + w++] = (unsigned short)q;
        }
    }
    MK_TRY(dev_alloc(&A->sweep_perm, perm.size()));
    MK_HIP(hipMemcpyAsync(A->sweep_perm, perm.data(), sizeof(unsigned short) * perm.size(), hipMemcpyHostToDevice, ctx().stream));
    MK_HIP(hipStreamSynchronize(ctx().stream));   // perm is a host temporary
  }
  // one diagonal entry in the base pattern and every row keeps it: 1/diag is one number (fused Jacobi-PCG step)
  int ndiag = 0;
  for (int u = 0; u < len; u++) ndiag += (off_of(base, u) == 0);
  bool all_have = (ndiag == 1);
  for (int q = 0; q < npat && all_have; q++) all_have = (mask[q] >> S.diag) & 1u;
  A->sweep_diag_uniform = all_have;
  // default only when the grid can fill the chip: ncp chunks per plane times segments of >= 8 planes
  static const int env_min = getenv("MK_SWEEP_AUTO_MIN") ? atoi(getenv("MK_SWEEP_AUTO_MIN")) : 0;   // tests: smaller grids take the sweep kernel too
  A->sweep_auto = (int64_t)S.ncp * std::max(1, S.K / 8) >= (env_min > 0 ? (int64_t)env_min : 2 * (int64_t)ctx().num_cus);
  // long rows on the vsweep kernel: one block per CU, so the grid must be large for its prologues to vanish (27-point 128^3:
  // 0.136 ms against 0.115 ms on the stream kernel; 256^3: 0.815 against 0.896 ms)
  // ... and the rows long: 2-D 9-point 3000^2 with its own values runs at 0.61 of the roofline here, 0.87 on the stream kernel with
  // its index patterns (its 72 B of values per row do not pay for a block-wide barrier pair per 512 rows)
  if (S.rpl == 1 && env_min <= 0 && (A->nrows < 6 * 1000 * 1000 || len < 20)) A->sweep_auto = false;
  return MK_OK;
}

// Patterns that differ only by a shift (the unknowns of one mesh node: same neighbours, offsets one less per unknown)
// share one "shape" = offsets minus the first offset.  The stream kernel keeps the whole table in LDS, so for long rows
// the table decides how many blocks fit a CU: shapes + (shape, shift) per pattern instead of npat full rows.
static int build_shape_table(mk_matrix *A, const std::vector<int> &tab) {
  static const int off = getenv("MK_NO_SHAPE_TABLE") ? atoi(getenv("MK_NO_SHAPE_TABLE")) : 0;
  const int npat = A->npat, pml = A->pml, stride = pml + 1;
  if (off || npat < 2) return MK_OK;
  std::map<std::vector<int>, int> start_of;   // shape (length, offsets - first) -> first int of its copy
  std::vector<int> shapes, map2((size_t)npat * 2);
  for (int q = 0; q < npat; q++) {
    const int len = tab[(size_t)q * stride];
    const int first = len > 0 ? tab[(size_t)q * stride + 1] : 0;
    std::vector<int> key(1 + len);
    key[0] = len;
    for (int k = 0; k < len; k++) key[1 + k] = tab[(size_t)q * stride + 1 + k] - first;
    auto it = start_of.find(key);
    if (it == start_of.end()) {
      it = start_of.emplace(key, (int)shapes.size()).first;
      shapes.insert(shapes.end(), key.begin(), key.end());
    }
    map2[(size_t)q * 2] = it->second;
    map2[(size_t)q * 2 + 1] = first;
  }
  if (shapes.size() & 1) shapes.push_back(0);
  if ((shapes.size() + map2.size()) * 4 > (size_t)npat * stride * 3) return MK_OK;   // pays from a quarter saved
  A->stab_shape_ints = (int)shapes.size();
  shapes.insert(shapes.end(), map2.begin(), map2.end());
  A->stab_ints = (int)shapes.size();
  MK_TRY(dev_alloc(&A->stab, shapes.size()));
  MK_HIP(hipMemcpyAsync(A->stab, shapes.data(), sizeof(int) * shapes.size(), hipMemcpyHostToDevice, ctx().stream));
  MK_HIP(hipStreamSynchronize(ctx().stream));
  return MK_OK;
}

// with_values: key = (offsets, values); the pattern kernel then needs neither col[] nor val[] nor rowptr[]
template <bool VAL>
static int try_compress(mk_matrix *A) {
  static const int no_idx = getenv("MK_NO_INDEX_COMPRESSION") ? atoi(getenv("MK_NO_INDEX_COMPRESSION")) : 0;
  static const int no_val = getenv("MK_NO_VALUE_COMPRESSION") ? atoi(getenv("MK_NO_VALUE_COMPRESSION")) : 0;
  if (no_idx || (VAL && no_val) || A->pat || A->nrows == 0 || A->info.max_row_len > pat_max_len() || A->info.max_row_len < 1) return MK_OK;
  Context &c = ctx();
  const int pml = A->info.max_row_len;
  unsigned long long *keys = nullptr;
  int *slot = nullptr, *ids = nullptr, *flags = nullptr, *written = nullptr;   // flags: [0] fail, [1] npat
  MK_TRY(dev_alloc(&keys, (size_t)kPatSlots));
  MK_TRY(dev_alloc(&slot, (size_t)A->nrows));
  MK_TRY(dev_alloc(&ids, (size_t)kPatSlots));
  MK_TRY(dev_alloc(&flags, 2));
  MK_HIP(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * kPatSlots, c.stream));
  MK_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 2, c.stream));
  const int blocks = (int)std::min<int64_t>((A->nrows + 255) / 256, 4096);
  hipLaunchKernelGGL(pat_insert_kernel<VAL>, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->rowptr, A->col, A->val, keys, slot, flags);
  hipLaunchKernelGGL(pat_compact_kernel, dim3(1), dim3(1024), 0, c.stream, keys, ids, flags + 1);
  int h[2] = {0, 0};
  MK_HIP(hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  bool ok = (h[0] == 0) && h[1] >= 1 && h[1] <= 65535 && (int64_t)h[1] * (pml + 1) <= kPatMaxInts;
  // a value table pays only when many rows share an entry (every block loads the whole table)
  if (VAL) ok = ok && (int64_t)h[1] * pml * 8 <= kPatMaxValBytes && (int64_t)h[1] <= std::max<int64_t>(32, A->nrows / 16);
  if (ok) {
    const int npat = h[1];
    MK_TRY(dev_alloc(&A->pat, (size_t)A->nrows));
    MK_TRY(dev_alloc(&A->ptab, (size_t)npat * (pml + 1)));
    if (VAL) MK_TRY(dev_alloc(&A->vtab, (size_t)npat * pml));
    MK_TRY(dev_alloc(&written, (size_t)npat));
    MK_HIP(hipMemsetAsync(written, 0, sizeof(int) * npat, c.stream));
    hipLaunchKernelGGL(pat_assign_kernel<VAL>, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->rowptr, A->col, A->val, slot, ids,
                       A->pat, A->ptab, A->vtab, pml, written);
    hipLaunchKernelGGL(pat_verify_kernel<VAL>, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->rowptr, A->col, A->val, A->pat,
                       A->ptab, A->vtab, pml, flags);
    MK_HIP(hipMemcpyAsync(h, flags, sizeof(int), hipMemcpyDeviceToHost, c.stream));
    MK_HIP(hipStreamSynchronize(c.stream));
    if (h[0] != 0) {   // collision: keep the plain CSR path
      dev_free(A->pat); dev_free(A->ptab); dev_free(A->vtab);
      A->pat = nullptr; A->ptab = nullptr; A->vtab = nullptr;
    } else {
      A->npat = npat; A->pml = pml;
      std::vector<int> tab((size_t)npat * (pml + 1));
      MK_HIP(hipMemcpyAsync(tab.data(), A->ptab, sizeof(int) * tab.size(), hipMemcpyDeviceToHost, c.stream));
      MK_HIP(hipStreamSynchronize(c.stream));
      int mo = 0;
      for (int q = 0; q < npat; q++)
        for (int k = 0; k < tab[(size_t)q * (pml + 1)]; k++) mo = std::max(mo, std::abs(tab[(size_t)q * (pml + 1) + 1 + k]));
      A->max_off = mo;
      MK_TRY(build_shape_table(A, tab));
      if (VAL) {   // padded copies for the pattern kernel (layout: see spmv_pattern_kernel)
        const int P8 = (pml + 7) & ~7, SP = P8 + 8;
        std::vector<double> vals((size_t)npat * pml), v8((size_t)npat * P8, 0.0);
        std::vector<int> o8((size_t)npat * SP, 0);
        MK_HIP(hipMemcpyAsync(vals.data(), A->vtab, sizeof(double) * vals.size(), hipMemcpyDeviceToHost, c.stream));
        MK_HIP(hipStreamSynchronize(c.stream));
        for (int q = 0; q < npat; q++) {
          const int len = tab[(size_t)q * (pml + 1)];
          o8[(size_t)q * SP] = len;
          for (int k = 0; k < len; k++) {
            o8[(size_t)q * SP + 8 + k] = tab[(size_t)q * (pml + 1) + 1 + k];
            v8[(size_t)q * P8 + k] = vals[(size_t)q * pml + k];
          }
        }
        MK_TRY(dev_alloc(&A->otab8, o8.size()));
        MK_TRY(dev_alloc(&A->vtab8, v8.size()));
        MK_HIP(hipMemcpyAsync(A->otab8, o8.data(), sizeof(int) * o8.size(), hipMemcpyHostToDevice, c.stream));
        MK_HIP(hipMemcpyAsync(A->vtab8, v8.data(), sizeof(double) * v8.size(), hipMemcpyHostToDevice, c.stream));
        MK_HIP(hipStreamSynchronize(c.stream));
        A->p8 = P8;
        MK_TRY(plan_sweep(A, tab, vals));
      } else {
        MK_TRY(plan_sweep(A, tab, std::vector<double>(), false));   // offsets from the table, values streamed
        MK_TRY(build_vsweep_rowpair(A));
      }
    }
  }
  dev_free(keys); dev_free(slot); dev_free(ids); dev_free(flags); dev_free(written);
  return MK_OK;
}

// ------------------------------------------------------------- shared column lists
// In a finite-element matrix with several unknowns per mesh node the rows of a node carry the same column list.  Rows whose
// list equals that of the row before them are marked, the lists are stored once (colc) and every row gets the start of the
// list it uses (cptr): the stream kernel then stages 4 B per entry of every third row only.  Lossless: same (column, value)
// pairs in the same order; col[] stays for mk_matrix_get_csr and the other kernels.
__global__ void share_mark_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col, uint32_t *__restrict__ w,
                                  unsigned long long *__restrict__ saved) {
  unsigned long long mine = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    const int b = rowptr[r], len = rowptr[r + 1] - b;
    bool same = false;
    if (r > 0 && len > 0) {
      const int pb = rowptr[r - 1];
      if (b - pb == len) {
        same = true;
        for (int k = 0; k < len; k++)
          if (col[pb + k] != col[b + k]) { same = false; break; }
      }
    }
    w[r] = same ? 0u : (uint32_t)len;
    if (same) mine += (unsigned long long)len;
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(saved, mine);
}

__global__ void share_build_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col, const uint32_t *__restrict__ w,
                                   const uint32_t *__restrict__ scan, int *__restrict__ cptr, int *__restrict__ colc) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    const int b = rowptr[r], len = rowptr[r + 1] - b;
    if (w[r] != 0u) {
      const int at = (int)scan[r];
      cptr[r] = at;
      for (int k = 0; k < len; k++) colc[at + k] = col[b + k];
    } else {
      cptr[r] = (int)scan[r] - len;   // the list of the row before (its own start when the row is empty)
    }
  }
}

static int try_share_columns(mk_matrix *A) {
  static const int off = getenv("MK_NO_SHARED_COLUMNS") ? atoi(getenv("MK_NO_SHARED_COLUMNS")) : 0;
  if (off || A->pat || A->colc || A->nrows < 2 || A->nnz < 1 || A->info.mean_row_len < 12.0) return MK_OK;
  Context &c = ctx();
  uint32_t *w = nullptr, *scan = nullptr;
  unsigned long long *saved = nullptr, h_saved = 0;
  MK_TRY(dev_alloc(&w, (size_t)A->nrows));
  MK_TRY(dev_alloc(&scan, (size_t)A->nrows));
  MK_TRY(dev_alloc(&saved, 1));
  MK_HIP(hipMemsetAsync(saved, 0, sizeof(unsigned long long), c.stream));
  const int blocks = (int)std::min<int64_t>((A->nrows + 255) / 256, 4096);
  hipLaunchKernelGGL(share_mark_kernel, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->rowptr, A->col, w, saved);
  MK_HIP(hipMemcpyAsync(&h_saved, saved, sizeof(h_saved), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  int rc = MK_OK;
  if ((double)h_saved >= 0.4 * (double)A->nnz) {   // pays from two rows in five repeating their neighbour
    const int64_t keep = A->nnz - (int64_t)h_saved;
    rc = exclusive_scan_u32(w, scan, (size_t)A->nrows, nullptr);
    if (rc == MK_OK) rc = dev_alloc(&A->cptr, (size_t)A->nrows);
    if (rc == MK_OK) rc = dev_alloc(&A->colc, (size_t)keep + 4);
    if (rc == MK_OK) {
      MK_HIP(hipMemsetAsync(A->colc + keep, 0, 4 * sizeof(int), c.stream));   // the tail a 16-byte load may touch
      hipLaunchKernelGGL(share_build_kernel, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->rowptr, A->col, w, scan, A->cptr, A->colc);
      MK_HIP(hipStreamSynchronize(c.stream));
      A->colc_len = keep;
    } else {
      dev_free(A->cptr); dev_free(A->colc);
      A->cptr = nullptr; A->colc = nullptr;
      rc = MK_OK;   // an optimisation only
    }
  }
  dev_free(w); dev_free(scan); dev_free(saved);
  return rc;
}

int analyse_matrix(mk_matrix *A) {
  if (A->analysed) return MK_OK;
  Context &c = ctx();
  AnalysisOut *d_out = nullptr, h{};
  MK_TRY(dev_alloc(&d_out, 1));
  memset(&h, 0, sizeof(h));
  h.min_len = 0x7fffffff;
  MK_HIP(hipMemcpyAsync(d_out, &h, sizeof(h), hipMemcpyHostToDevice, c.stream));
  if (A->nrows > 0) {
    int blocks = (int)std::min<int64_t>((A->nrows + 255) / 256, 2048);
    hipLaunchKernelGGL(analyse_rows_kernel, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->rowptr, d_out);
    hipLaunchKernelGGL(analyse_chunks_kernel, dim3(std::min(blocks, 512)), dim3(256), 0, c.stream, A->nrows, A->rowptr, d_out);
  }
  MK_HIP(hipMemcpyAsync(&h, d_out, sizeof(h), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  dev_free(d_out);
  mk_matrix_info &I = A->info;
  I.nrows = A->nrows; I.ncols = A->ncols; I.nnz = A->nnz;
  I.max_row_len = h.max_len;
  I.min_row_len = A->nrows ? h.min_len : 0;
  I.mean_row_len = A->nrows ? (double)A->nnz / (double)A->nrows : 0.0;
  for (int k = 0; k < 16; k++) I.hist[k] = (int64_t)h.hist[k];
  // selector: lanes per row = smallest power of two >= mean row length (2..64)
  int lanes = 2;
  while (lanes < 64 && (double)lanes < I.mean_row_len) lanes *= 2;
  if (A->lanes == 0) A->lanes = lanes;
  // stream kernel: 256/G rows per chunk, G lanes per row in the row pass; usable when the fullest chunk (the most entries
  // in any 256/G consecutive rows, measured above -- not rows x longest row, which one long row among short ones inflates)
  // fits the LDS budget.  G from the mean row length, doubled while the fullest chunk does not fit; a matrix that
  // needs more than two doublings has rows far longer than its mean and goes to the split kernel.
  int sl = 1;
  static const double env_div = getenv("MK_STREAM_DIV") ? atof(getenv("MK_STREAM_DIV")) : 16.0;
  while (sl < 64 && (double)sl * env_div < I.mean_row_len) sl *= 2;
  static const int env_sl = getenv("MK_STREAM_LANES") ? atoi(getenv("MK_STREAM_LANES")) : 0;   // tuning: 1..64, power of two
  if (env_sl >= 1 && env_sl <= 64 && !(env_sl & (env_sl - 1))) sl = env_sl;
  const int sl_mean = sl;
  auto fullest = [&](int lanes_) { int k = 0; while ((1 << k) < lanes_) k++; return h.chunk_max[k]; };
  while (sl < 64 && fullest(sl) > kStreamMaxCap) sl *= 2;
  A->stream_lanes = sl;
  A->stream_cap = (int)((std::max<int64_t>(64, fullest(sl)) + 3) & ~(int64_t)3);
  const bool stream_fits = A->stream_cap <= kStreamMaxCap;
  // ... and so does one whose longest row would keep its G lanes busy for more than 32 entries each while the rest of the
  // block waits at the barrier (200 rows of 1200 entries among 2 M rows of 6: 152 us on this kernel, 107 us on the split kernel)
  const bool stream_ok = stream_fits && sl <= 4 * sl_mean && I.max_row_len <= 32 * sl;
  // regular rows -> stream; rows too long for a row-aligned LDS chunk (skewed / power law) -> split
  const bool was_auto = (A->kernel == MK_SPMV_AUTO);
  if (A->kernel == MK_SPMV_AUTO) A->kernel = stream_ok ? MK_SPMV_STREAM : MK_SPMV_SPLIT;
  // A few fat rows among regular ones (constraint rows of a finite-element matrix) fail the stream kernel's test on the longest
  // row, but not the band kernel's needs: its chunks are packed by entries, a fat row takes a chunk (nearly) of its own and
  // costs that chunk a longer row pass, nothing else.  Where such rows are rare (one row in a thousand beyond the stream
  // kernel's limit) and the band plan exists, the band kernel serves the matrix instead of the split kernel (the stream kernel
  // stays behind it for the Jacobi sweep).
  if (was_auto && A->kernel == MK_SPMV_SPLIT && stream_fits && sl <= 4 * sl_mean) {
    int64_t fat = 0;
    for (int k = 0; k < 16; k++)
      if ((k == 0 ? 0 : (1 << (k - 1))) > 32 * sl) fat += I.hist[k];   // (buckets wholly beyond the limit; the one that straddles it is not counted)
    static const int no_rescue = getenv("MK_NO_BAND_FAT_ROWS") ? atoi(getenv("MK_NO_BAND_FAT_ROWS")) : 0;
    if (!no_rescue && fat * 1000 <= A->nrows && I.max_row_len <= 2048) {
      MK_TRY(build_band_plan(A));
      if (A->band_desc || A->nstrips) A->kernel = MK_SPMV_STREAM;
    }
  }
  if (A->kernel == MK_SPMV_STREAM && !stream_fits) A->kernel = MK_SPMV_SPLIT;
  int64_t longr = 0;
  for (int k = 12; k < 16; k++) longr += I.hist[k];   // rows of 2048 entries or more always cross a chunk border
  I.long_rows = longr;
  I.stream_lanes = A->stream_lanes;
  I.stream_chunk = A->stream_cap;
  I.kernel = A->kernel;
  I.lanes_per_row = A->lanes;
  I.index_patterns = 0;
  if (A->kernel == MK_SPMV_STREAM && !(A->band_desc || A->nstrips)) {
    MK_TRY(try_compress<true>(A));                 // offsets and values shared by rows -> pattern kernel
    if (!A->pat) MK_TRY(try_compress<false>(A));   // offsets only -> stream kernel without col[]
    if (!A->pat) MK_TRY(try_share_columns(A));     // rows repeating their neighbour's column list -> lists stored once
    if (!A->pat && A->cptr) MK_TRY(build_row_groups(A));   // ... and x gathered once per list (row-group kernel)
    if (!A->pat && !A->grow) MK_TRY(build_band_plan(A));   // rows within a bounded distance of the diagonal: x window in LDS (band kernel)
    if (A->vtab && was_auto) A->kernel = (A->sweep && A->sweep_auto) ? MK_SPMV_SWEEP : MK_SPMV_PATTERN;
    static const int no_vsweep = getenv("MK_NO_VSWEEP") ? atoi(getenv("MK_NO_VSWEEP")) : 0;
    if (!A->vtab && A->pat && A->sweep && A->sweep_streamed && A->sweep_auto && was_auto && !no_vsweep) A->kernel = MK_SPMV_VSWEEP;
  }
  I.kernel = A->kernel;
  I.index_patterns = A->pat ? A->npat : 0;
  I.value_patterns = A->vtab ? A->npat : 0;
  I.shared_columns = A->cptr ? A->nnz - A->colc_len : 0;
  I.row_groups = A->grow ? A->ngroups : 0;
  I.group_lanes = A->grow ? A->group_lanes : 0;
  I.band_window = (A->band_desc || A->nstrips) ? A->band_window : 0;
  I.band_strips = A->nstrips;
  A->analysed = true;
  return MK_OK;
}

int ensure_dinv(mk_matrix *A) {
  if (A->dinv) return MK_OK;
  MK_TRY(dev_alloc(&A->dinv, (size_t)A->nrows));
  if (A->nrows) {
    int blocks = (int)std::min<int64_t>((A->nrows + 255) / 256, 4096);
    hipLaunchKernelGGL(diag_inverse_kernel, dim3(blocks), dim3(256), 0, ctx().stream, A->nrows, A->rowptr, A->col, A->val, A->dinv);
    MK_HIP(hipGetLastError());
  }
  return MK_OK;
}

// ------------------------------------------------------------------- vector kernel
// MODE 0: y = alpha*A*x + beta*y.   MODE 1: y = A*x and per-block partial sums of
// <y,x> (slab component 0) and <x,x> (component 1) for the CG step.
template <int G, int MODE>
__global__ __launch_bounds__(kThreads) void spmv_vector_kernel(
    int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, double alpha, double beta, double *__restrict__ slab,
    const int *__restrict__ done) {
  if (done != nullptr && *done != 0) return;
  constexpr int ROWS = kThreads / G;
  const int g = threadIdx.x % G;
  const int rl = threadIdx.x / G;
  double acc_yx = 0.0, acc_xx = 0.0;
  for (int64_t base = (int64_t)blockIdx.x * ROWS; base < nrows; base += (int64_t)gridDim.x * ROWS) {
    const int64_t row = base + rl;
    double sum = 0.0;
    if (row < nrows) {
      const int b = rowptr[row], e = rowptr[row + 1];
      for (int j = b + g; j < e; j += G) sum += val[j] * x[col[j]];
    }
    sum = group_sum<G>(sum);
    if (g == 0 && row < nrows) {
      if (MODE == 0) {
        y[row] = (beta == 0.0) ? alpha * sum : alpha * sum + beta * y[row];
      } else {
        y[row] = sum;
        const double xr = x[row];
        acc_yx += sum * xr;
        acc_xx += xr * xr;
      }
    }
  }
  if (MODE == 1) {
    __shared__ double smem[2 * (kThreads / 64)];
    double v[2] = {acc_yx, acc_xx};
    block_sum<2, kThreads>(v, smem);
    if (threadIdx.x == 0) {
      slab[0 * kSlabStride + blockIdx.x] = v[0];
      slab[1 * kSlabStride + blockIdx.x] = v[1];
    }
  }
}

template <int MODE>
static int launch_vector(const mk_matrix *A, double alpha, const double *x, double beta, double *y, double *slab,
                         const int *done, int *nblocks) {
  Context &c = ctx();
  const int G = A->lanes;
  const int64_t rows_per_block = kThreads / G;
  int64_t want = (A->nrows + rows_per_block - 1) / rows_per_block;
  int blocks = (int)std::min<int64_t>(std::max<int64_t>(want, 1), (int64_t)c.num_cus * 8);
  if (blocks > kSlabStride) blocks = kSlabStride;
  if (nblocks) *nblocks = blocks;
  if (dry_launch()) return MK_OK;   // grid size only (mk_spmv_dot_static checks it against its scratch before anything is enqueued)
#define MK_LAUNCH_G(GG)                                                                                      \
  case GG:                                                                                                   \
    hipLaunchKernelGGL((spmv_vector_kernel<GG, MODE>), dim3(blocks), dim3(kThreads), 0, c.stream, A->nrows, \
                       A->rowptr, A->col, A->val, x, y, alpha, beta, slab, done);                            \
    break;
  switch (G) {
    MK_LAUNCH_G(1)
    MK_LAUNCH_G(2)
    MK_LAUNCH_G(4)
    MK_LAUNCH_G(8)
    MK_LAUNCH_G(16)
    MK_LAUNCH_G(32)
    MK_LAUNCH_G(64)
    default:
      return fail(MK_ERR_INVALID, "spmv: lanes per row %d is not a power of two in 1..64", G);
  }
#undef MK_LAUNCH_G
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// ------------------------------------------------------------------- stream kernel
// Chunk = 256/G consecutive rows.  Pass 1 copies the chunk's val/col spans into LDS with
// 16-byte loads (coalesced whatever the row lengths are); pass 2 gives every row G lanes
// that walk the row out of LDS and gather x.  With G=1 (rows up to ~8 entries) lanes of a
// wave own consecutive rows, so the k-th gather of a stencil row is 64 consecutive x
// entries, and each row is summed left to right exactly as the serial reference does.
// Blocks that share an XCD (blockIdx % 8) take a contiguous range of chunks, so the x
// entries that neighbouring rows share are served by that XCD's L2.
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i4_t __attribute__((ext_vector_type(4)));

// U = LDS capacity class: cap <= 1024*U - 8 entries, so a thread stages at most 2U 16-byte
// val loads and U 16-byte col loads per chunk.  The loads of chunk k+1 are issued between
// the x gathers of chunk k and their first use, so they stay in flight through pass 2 and
// the barrier (loads retire in order: anything issued BEFORE the gathers would be waited
// for with them).
// COMP: column indices come from the pattern table (pat/ptab, staged in LDS) instead of col[].
// NT: the matrix exceeds the 256 MB Infinity Cache, so its values are loaded non-temporally and
// leave the cache to x (measured: +5..8 % on 512^3; on cache-resident matrices plain loads win by 20 %).
template <int G, int MODE, int U, bool COMP, bool NT>
__global__ __launch_bounds__(kThreads) void spmv_stream_kernel(
    int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, double alpha, double beta, double *__restrict__ slab,
    const int *__restrict__ done, int cap, const unsigned short *__restrict__ pat, const int *__restrict__ ptab,
    int tab_ints, int shape_ints, int pml, const double *__restrict__ rhs, const double *__restrict__ extra, int tile_p, int tile_t) {
  // MODE 2 = one Jacobi sweep: y = x + D^-1 (rhs - A x - extra), D picked up from the row while it
  // streams by; slab component 0 receives the partial of ||rhs - A x - extra||^2
  if (done != nullptr && *done != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *sval = reinterpret_cast<double *>(smem_raw);                     // cap + 2 doubles
  int *scol = reinterpret_cast<int *>(smem_raw + (size_t)(cap + kStreamAlignV + 2) * 8);   // cap + 40 ints, or the pattern table
  constexpr int R = kThreads / G;
  constexpr int NV = 2 * U, NI = COMP ? 0 : U;
  if (COMP)
    for (int i = threadIdx.x; i < tab_ints; i += kThreads) scol[i] = ptab[i];   // visible after the first barrier below
  const int tid = threadIdx.x;
  const int g = tid % G, rl = tid / G;
  const int64_t nchunks = (nrows + R - 1) / R;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int64_t per_xcd = (nchunks + 7) / 8;
  const int64_t first = xcd * per_xcd, last = min(first + per_xcd, nchunks);
  double acc_yx = 0.0, acc_xx = 0.0;

  // Traversal order inside the XCD's range.  Linear by default.  When the matrix couples rows a
  // long stride apart (tile_p = that stride in chunks, e.g. one grid plane), the range is walked in
  // tiles of tile_t chunks through all planes before moving to the next tile, so the x entries of
  // planes z-1, z, z+1 are re-used from L2 a few chunks later instead of a whole plane later.
  const int64_t span = last > first ? last - first : 0;
  const int64_t planes = tile_p > 0 ? (span + tile_p - 1) / tile_p : 0;
  const int64_t per_tile = planes * tile_t;
  const int64_t total_q = tile_p > 0 ? ((tile_p + tile_t - 1) / tile_t) * per_tile : span;
  auto decode = [&](int64_t q) -> int64_t {   // chunk index, or -1 for a hole of the tiling
    if (tile_p <= 0) return first + q;
    const int64_t t = q / per_tile, rem = q - t * per_tile;
    const int64_t z = rem / tile_t, i = rem - z * tile_t;
    const int64_t pos = t * tile_t + i;
    if (pos >= tile_p) return -1;
    const int64_t ch = first + z * tile_p + pos;
    return ch < last ? ch : -1;
  };
  auto next_chunk = [&](int64_t &q) -> int64_t {   // first valid chunk at or after position q (step = slots)
    for (; q < total_q; q += slots) {
      const int64_t ch = decode(q);
      if (ch >= 0) return ch;
    }
    return last;
  };
  d2_t rv[NV];
  i4_t ri[NI > 0 ? NI : 1];
  int64_t q = slot;
  int64_t chunk = next_chunk(q);
  int cb = 0, ce = 0;
  int64_t r0 = 0, r1 = 0;
  // shared column lists (try_share_columns): without a pattern table the ptab argument carries cptr, col is the compact
  // array, and the chunk's column span is [cptr[first row], end of the last row's list) instead of the value span
  const int *cptr = COMP ? nullptr : ptab;
  int ccb = 0, cce = 0;
  auto col_span = [&](int64_t r0_, int64_t r1_, int cb_, int ce_, int &ccb_, int &cce_) {
    if (COMP || cptr == nullptr) { ccb_ = cb_; cce_ = ce_; return; }
    ccb_ = cptr[r0_];
    cce_ = cptr[r1_ - 1] + (ce_ - rowptr[r1_ - 1]);
  };
  auto stage_loads = [&](int cb_, int ce_, int ccb_, int cce_) {
    const int vb_ = cb_ & ~(kStreamAlignV - 1), ib_ = ccb_ & ~(kStreamAlignI - 1);
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int k = vb_ + (tid + i * kThreads) * 2;
      if (k < ce_) rv[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(val + k)) : *reinterpret_cast<const d2_t *>(val + k);
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int k = ib_ + (tid + i * kThreads) * 4;
      if (k < cce_) ri[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const i4_t *>(col + k)) : *reinterpret_cast<const i4_t *>(col + k);
    }
    (void)ib_;
  };
  if (chunk < last) {
    r0 = chunk * R;
    r1 = min(r0 + R, nrows);
    cb = rowptr[r0];
    ce = rowptr[r1];
    col_span(r0, r1, cb, ce, ccb, cce);
    stage_loads(cb, ce, ccb, cce);
  }
  while (chunk < last) {
    const int vb = cb & ~(kStreamAlignV - 1), ib = ccb & ~(kStreamAlignI - 1);
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int k = vb + (tid + i * kThreads) * 2;
      if (k < ce) *reinterpret_cast<d2_t *>(sval + (k - vb)) = rv[i];
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int k = ib + (tid + i * kThreads) * 4;
      if (k < cce) *reinterpret_cast<i4_t *>(scol + (k - ib)) = ri[i];
    }
    const int64_t row = r0 + rl;
    const bool have_row = row < r1;
    int b = 0, e = 0, pid = 0, cstart = ccb;
    if (have_row) { b = rowptr[row] - cb; e = rowptr[row + 1] - cb; cstart = (!COMP && cptr) ? cptr[row] : cb + b; }
    if (COMP && have_row) pid = NT ? __builtin_nontemporal_load(pat + row) : pat[row];
    __syncthreads();
    const double *sv = sval + (cb - vb);
    int pbase = pid * (pml + 1), pshift = 0;
    if (COMP && shape_ints > 0) {   // two-level table: shapes, then (shape start, first offset) per pattern
      pbase = scol[shape_ints + 2 * pid];
      pshift = scol[shape_ints + 2 * pid + 1];
    }
    const int *sc = COMP ? scol + pbase + 1 - b : scol + (cstart - ib) - b;   // sc[j]: column (or column - row - shift) of entry j
    const int cadd = COMP ? (int)row + pshift : 0;
    // first 8 entries of the row (per lane): all gathers issued before anything waits
    const int lastj = max(e - 1, 0);
    int cj[8];
    double xj[8], vj[8];
    const bool any = have_row && (b + g < e);
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int j = min(b + g + t * G, lastj);
      cj[t] = any ? sc[j] + cadd : 0;
      vj[t] = any ? sv[j] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 8; t++) xj[t] = any ? x[cj[t]] : 0.0;
    const double xr = (MODE != 0 && g == 0 && have_row) ? x[row] : 0.0;
    const double yold = (MODE == 0 && g == 0 && have_row && beta != 0.0) ? y[row] : 0.0;
    double rhs_r = 0.0;
    if (MODE == 2 && g == 0 && have_row) rhs_r = extra ? rhs[row] - extra[row] : rhs[row];
    // prefetch the next chunk (stays in flight while this one is reduced)
    q += slots;
    const int64_t next = next_chunk(q);
    int ncb = 0, nce = 0, nccb = 0, ncce = 0;
    int64_t nr0 = 0, nr1 = 0;
    if (next < last) {
      nr0 = next * R;
      nr1 = min(nr0 + R, nrows);
      ncb = rowptr[nr0];
      nce = rowptr[nr1];
      col_span(nr0, nr1, ncb, nce, nccb, ncce);
      stage_loads(ncb, nce, nccb, ncce);
    }
    double sum = 0.0, dv = 0.0;
    int dj = -1;
#pragma unroll
    for (int t = 0; t < 8; t++)
      if (have_row && b + g + t * G < e) {
        sum += vj[t] * xj[t];
        if (MODE == 2 && cj[t] == (int)row) { dj = b + g + t * G; dv = vj[t]; }   // the last stored diagonal entry wins
      }
    for (int j0 = b + g + 8 * G; j0 < e; j0 += 4 * G) {   // rows longer than 8 lanes' worth: batches of 4, gathers issued together
      int c4[4];
      double v4[4], x4[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int j = min(j0 + t * G, lastj);
        c4[t] = sc[j] + cadd;
        v4[t] = sv[j];
      }
#pragma unroll
      for (int t = 0; t < 4; t++) x4[t] = x[c4[t]];
#pragma unroll
      for (int t = 0; t < 4; t++)
        if (j0 + t * G < e) {
          sum += v4[t] * x4[t];
          if (MODE == 2 && c4[t] == (int)row) { dj = j0 + t * G; dv = v4[t]; }
        }
    }
    if (G > 1) {
      sum = group_sum<G>(sum);
      if (MODE == 2) {
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) {
          const int oj = __shfl_down(dj, off, G);
          const double ov = __shfl_down(dv, off, G);
          if (oj > dj) { dj = oj; dv = ov; }
        }
      }
    }
    if (g == 0 && have_row) {
      double yv;
      if (MODE == 0) {
        yv = (beta == 0.0) ? alpha * sum : alpha * sum + beta * yold;
      } else if (MODE == 2) {
        const double res = rhs_r - sum;
        yv = xr + res / dv;
        acc_yx += res * res;
      } else {
        yv = sum;
        acc_yx += sum * xr;
        acc_xx += xr * xr;
      }
      if (NT) __builtin_nontemporal_store(yv, y + row); else y[row] = yv;
    }
    __syncthreads();
    chunk = next; cb = ncb; ce = nce; ccb = nccb; cce = ncce; r0 = nr0; r1 = nr1;
  }
  if (MODE != 0) {
    double *red = reinterpret_cast<double *>(smem_raw);
    double v[2] = {acc_yx, acc_xx};
    block_sum<2, kThreads>(v, red);
    if (threadIdx.x == 0) {
      slab[0 * kSlabStride + blockIdx.x] = v[0];
      if (MODE == 1) slab[1 * kSlabStride + blockIdx.x] = v[1];
    }
  }
}

template <int MODE, int U, bool COMP, bool NT>
static int launch_stream_u(const mk_matrix *A, double alpha, const double *x, double beta, double *y, double *slab,
                           const int *done, int *nblocks, const double *rhs = nullptr, const double *extra = nullptr) {
  Context &c = ctx();
  const int G = A->stream_lanes, cap = A->stream_cap;
  const bool shapes = COMP && A->stab != nullptr;
  const int tab_ints = COMP ? (shapes ? A->stab_ints : A->npat * (A->pml + 1)) : 0;
  const bool shared_cols = !COMP && A->cptr != nullptr;
  const int *table = COMP ? (shapes ? A->stab : A->ptab) : (shared_cols ? A->cptr : nullptr);
  const int *colv = shared_cols ? A->colc : A->col;
  const int shape_ints = shapes ? A->stab_shape_ints : 0;
  const size_t lds = (size_t)(cap + kStreamAlignV + 2) * 8 + (COMP ? (size_t)tab_ints * 4 + 16 : (size_t)(cap + kStreamAlignI + 8) * 4);  // cap is a multiple of 4
  const int64_t nchunks = (A->nrows + (kThreads / G) - 1) / (kThreads / G);
  // resident blocks per CU for this instantiation (registers + LDS), so the grid is one full wave of blocks
  static const int env_bpc = getenv("MK_STREAM_BPC") ? atoi(getenv("MK_STREAM_BPC")) : 0;
  int per_cu = env_bpc > 0 ? env_bpc : A->stream_bpc[MODE];
  if (per_cu <= 0) {   // first launch of this mode on this matrix: ask once, the query costs ~10 us of host time
    int occ = 0;
    hipError_t oe = hipErrorInvalidValue;
#define MK_OCC_S(GG) case GG: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spmv_stream_kernel<GG, MODE, U, COMP, NT>, kThreads, lds); break;
    switch (G) { MK_OCC_S(1) MK_OCC_S(2) MK_OCC_S(4) MK_OCC_S(8) MK_OCC_S(16) MK_OCC_S(32) MK_OCC_S(64) default: break; }
#undef MK_OCC_S
    per_cu = (oe == hipSuccess && occ > 0) ? std::min(occ, 8) : 4;
    A->stream_bpc[MODE] = per_cu;
  }
  int64_t blocks = std::min<int64_t>((nchunks + 7) / 8 * 8, (int64_t)c.num_cus * per_cu);
  blocks = std::max<int64_t>(8, blocks / 8 * 8);
  if (blocks > kSlabStride) blocks = kSlabStride;
  if (nblocks) *nblocks = (int)blocks;
  if (dry_launch()) return MK_OK;   // grid size only (mk_spmv_dot_static checks it against its scratch before anything is enqueued)
  // Optional tiled traversal (MK_STREAM_TILE=<chunks>, off by default): measured on 7-pt 512^3 it is
  // 2-4 % SLOWER than the linear sweep (tile 16..128) although it removes the x re-fetches of the z+-1
  // planes: those re-fetches are served by the Infinity Cache and the tiling breaks up the value stream.
  static const int env_tile = getenv("MK_STREAM_TILE") ? atoi(getenv("MK_STREAM_TILE")) : 0;
  int tile_p = 0, tile_t = 0;
  const int rows_per_chunk = kThreads / G;
  if (COMP && env_tile > 0 && A->max_off % rows_per_chunk == 0 && A->max_off / rows_per_chunk >= 4 * env_tile &&
      24.0 * A->max_off > 1.5e6) {
    tile_p = A->max_off / rows_per_chunk;
    tile_t = env_tile;
  }
#define MK_LAUNCH_S(GG)                                                                                            \
  case GG:                                                                                                         \
    hipLaunchKernelGGL((spmv_stream_kernel<GG, MODE, U, COMP, NT>), dim3((unsigned)blocks), dim3(kThreads), lds,      \
                       c.stream, A->nrows, A->rowptr, colv, A->val, x, y, alpha, beta, slab, done, cap, A->pat,      \
                       table, tab_ints, shape_ints, A->pml, rhs, extra, tile_p, tile_t);                                     \
    break;
  switch (G) {
    MK_LAUNCH_S(1)
    MK_LAUNCH_S(2)
    MK_LAUNCH_S(4)
    MK_LAUNCH_S(8)
    MK_LAUNCH_S(16)
    MK_LAUNCH_S(32)
    MK_LAUNCH_S(64)
    default:
      return fail(MK_ERR_INVALID, "spmv: stream lanes per row %d invalid", G);
  }
#undef MK_LAUNCH_S
  MK_HIP(hipGetLastError());
  return MK_OK;
}

template <int MODE>
static int launch_stream(const mk_matrix *A, double alpha, const double *x, double beta, double *y, double *slab,
                         const int *done, int *nblocks, const double *rhs = nullptr, const double *extra = nullptr) {
  const int cap = A->stream_cap;
  // values larger than the Infinity Cache are streamed non-temporally
  const bool nt = nt_matrix((double)A->nnz * 8.0);
#define MK_STREAM_PICK(UU)                                                                                         \
  if (cap <= 1024 * UU - 32) {                                                                                     \
    if (A->pat) return nt ? launch_stream_u<MODE, UU, true, true>(A, alpha, x, beta, y, slab, done, nblocks, rhs, extra)   \
                          : launch_stream_u<MODE, UU, true, false>(A, alpha, x, beta, y, slab, done, nblocks, rhs, extra); \
    return nt ? launch_stream_u<MODE, UU, false, true>(A, alpha, x, beta, y, slab, done, nblocks, rhs, extra)              \
              : launch_stream_u<MODE, UU, false, false>(A, alpha, x, beta, y, slab, done, nblocks, rhs, extra);            \
  }
  MK_STREAM_PICK(1)
  MK_STREAM_PICK(2)
  MK_STREAM_PICK(3)
  MK_STREAM_PICK(4)
#undef MK_STREAM_PICK
  return fail(MK_ERR_UNSUPPORTED, "spmv: stream capacity %d too large", cap);
}

// ------------------------------------------------------------------ pattern kernel
// Rows drawn from a small table of (offsets, values) patterns: constant-coefficient stencils and
// banded operators.  One lane per row; the only matrix data read from HBM is the 2-byte pattern id,
// offsets and values come out of LDS (lanes of a wave mostly share a pattern: broadcast reads), so
// the product is bound by x and y traffic alone.  Each row is summed left to right over the same
// (column, value) pairs as the CSR arrays hold -> bitwise identical to the serial reference.
// Lossless by construction: pat_verify_kernel<true> compared every row with its table entry.
// Chunk = 256 consecutive rows; XCD-contiguous chunk ranges and the optional plane-tiled traversal
// as in the stream kernel (here the x gathers are most of the traffic, so their L2 reuse matters).
// WIDE: n * 8 bytes does not fit 32-bit offsets (n >= 2^29 rows): plain 64-bit addressing.  Otherwise every
// access is base (SGPR pair) + 32-bit byte offset, one v_add_lshl per gather instead of 64-bit pointer math
// (the first version spent more issue slots on addresses than on loads: 1.14 ms at 512^3).
template <bool WIDE, typename T>
__device__ __forceinline__ T ld_at(const T *base, int64_t row, int off) {
  if (WIDE) return base[row + off];
  const unsigned byte = ((unsigned)row + (unsigned)off) * (unsigned)sizeof(T);
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte);
}
template <bool WIDE>
__device__ __forceinline__ void st_at(double *base, int64_t row, double v) {
  if (WIDE) { base[row] = v; return; }
  *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + ((unsigned)row << 3)) = v;
}

// tables: otab = per pattern [len, 7 x pad, offsets padded with 0 up to P8 (a multiple of 8)], vtab = per
// pattern P8 values padded with 0.0: a padded slot addresses x[row], which exists, and is never added.
// A wave whose lanes all carry the same pattern (interior rows) reads offsets and values with scalar
// loads from the global copy of the tables (no vector instruction per entry except the gather and the
// fma); mixed waves (boundary rows) read the LDS copy.
typedef double d2u_t __attribute__((ext_vector_type(2), aligned(8)));   // 16-byte access that is only 8-byte aligned

template <int MODE, bool WIDE>
__global__ __launch_bounds__(kThreads) void spmv_pattern_kernel(
    int64_t nrows, const unsigned short *__restrict__ pat, const int *__restrict__ otab, const double *__restrict__ vtab,
    int npat, int P8, const double *__restrict__ x, double *__restrict__ y, double alpha, double beta,
    double *__restrict__ slab, const int *__restrict__ done, const double *__restrict__ rhs, const double *__restrict__ extra,
    int tile_p, int tile_t) {
  if (done != nullptr && *done != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int SP = P8 + 8;
  double *sv = reinterpret_cast<double *>(smem_raw);                             // npat * P8 values (>= 8 doubles: reduction scratch)
  int *st = reinterpret_cast<int *>(smem_raw + (size_t)max(npat * P8, 8) * 8);   // npat * SP
  for (int i = threadIdx.x; i < npat * P8; i += kThreads) sv[i] = vtab[i];
  for (int i = threadIdx.x; i < npat * SP; i += kThreads) st[i] = otab[i];
  __syncthreads();
  const int tid = threadIdx.x;
  constexpr int CR = kPatternChunkRows;   // rows per chunk: every lane owns two consecutive rows
  const int64_t nchunks = (nrows + CR - 1) / CR;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int64_t per_xcd = (nchunks + 7) / 8;
  const int64_t first = xcd * per_xcd, last = min(first + per_xcd, nchunks);
  const int64_t span = last > first ? last - first : 0;
  const int64_t planes = tile_p > 0 ? (span + tile_p - 1) / tile_p : 0;
  const int64_t per_tile = planes * tile_t;
  const int64_t total_q = tile_p > 0 ? ((tile_p + tile_t - 1) / tile_t) * per_tile : span;
  double acc0 = 0.0, acc1 = 0.0;

  // one row, table entry read per lane from LDS (mixed waves: grid boundaries, the ragged last chunk)
  auto one_row = [&](int64_t row, int pid) {
    const double xr = (MODE != 0) ? ld_at<WIDE>(x, row, 0) : 0.0;
    const double yold = (MODE == 0 && beta != 0.0) ? ld_at<WIDE>(y, row, 0) : 0.0;
    double rhs_r = 0.0;
    if (MODE == 2) rhs_r = extra ? ld_at<WIDE>(rhs, row, 0) - ld_at<WIDE>(extra, row, 0) : ld_at<WIDE>(rhs, row, 0);
    const int *t = st + pid * SP;
    const double *v = sv + pid * P8;
    const int len = t[0];
    double sum = 0.0, dv = 0.0;
    for (int k0 = 0; k0 < len; k0 += 8) {   // 8 gathers in flight per lane before the first is used
      int oj[8];
      double xj[8], vj[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { oj[u] = t[8 + k0 + u]; vj[u] = v[k0 + u]; }
#pragma unroll
      for (int u = 0; u < 8; u++) xj[u] = ld_at<WIDE>(x, row, oj[u]);
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (k0 + u < len) {
          sum += vj[u] * xj[u];
          if (MODE == 2 && oj[u] == 0) dv = vj[u];   // the last stored diagonal entry wins
        }
    }
    if (MODE == 0) {
      st_at<WIDE>(y, row, (beta == 0.0) ? alpha * sum : alpha * sum + beta * yold);
    } else if (MODE == 2) {
      const double res = rhs_r - sum;
      st_at<WIDE>(y, row, xr + res / dv);
      acc0 += res * res;
    } else {
      st_at<WIDE>(y, row, sum);
      acc0 += sum * xr;
      acc1 += xr * xr;
    }
  };

  // traversal: position q -> chunk (or a hole of the tiling); tile t of a plane, then all planes of the
  // range, then the next tile
  auto next_chunk = [&](int64_t &q) -> int64_t {   // first valid chunk at or after position q (step = slots), else -1
    for (; q < total_q; q += slots) {
      if (tile_p <= 0) return first + q;
      const int64_t t = q / per_tile, rem = q - t * per_tile;
      const int64_t z = rem / tile_t, i = rem - z * tile_t;
      const int64_t pos = t * tile_t + i;
      const int64_t ch = first + z * tile_p + pos;
      if (pos < tile_p && ch < last) return ch;
    }
    return -1;
  };
  // the two pattern ids of a lane's rows; issued one chunk ahead so that only the gathers' latency
  // (not pattern id -> table -> gathers) is exposed per chunk
  auto load_ids = [&](int64_t chunk) -> unsigned {
    if (chunk < 0) return 0u;
    const int64_t row = chunk * CR + 2 * tid;
    if (row + 1 < nrows) return ld_at<WIDE>(reinterpret_cast<const unsigned *>(pat), row >> 1, 0);
    if (row < nrows) return (unsigned)ld_at<WIDE>(pat, row, 0);
    return 0u;
  };
  int64_t q = slot;
  int64_t chunk = next_chunk(q);
  unsigned pp = load_ids(chunk);
  while (chunk >= 0) {
    q += slots;
    const int64_t chunk_n = next_chunk(q);
    const unsigned pp_n = load_ids(chunk_n);
    const int64_t row = chunk * CR + 2 * tid;   // even: 4-byte aligned in pat, 16-byte aligned in x and y
    const unsigned pp_c = pp;
    chunk = chunk_n;
    pp = pp_n;
    if (row >= nrows) continue;
    const bool two = row + 1 < nrows;
    const int pidA = (int)(pp_c & 0xffffu);
    const int pidB = two ? (int)(pp_c >> 16) : pidA;
    const int pid0 = __builtin_amdgcn_readfirstlane(pidA);
    if (__all(two && pidA == pid0 && pidB == pid0)) {
      // interior wave: 128 rows of one pattern.  Offsets and values by scalar loads from the global
      // tables; the two rows of a lane gather x[row+off], x[row+1+off] with ONE 16-byte load.
      const int *__restrict__ t = otab + (size_t)pid0 * SP;
      const double *__restrict__ v = vtab + (size_t)pid0 * P8;
      const int len = t[0];
      d2_t xr = {0.0, 0.0}, yold = {0.0, 0.0}, rh = {0.0, 0.0};   // row is even: these are 16-byte aligned
      if (MODE != 0) xr = ld_at<WIDE>(reinterpret_cast<const d2_t *>(x), row >> 1, 0);
      if (MODE == 0 && beta != 0.0) yold = ld_at<WIDE>(reinterpret_cast<const d2_t *>(y), row >> 1, 0);
      if (MODE == 2) {
        rh = ld_at<WIDE>(reinterpret_cast<const d2_t *>(rhs), row >> 1, 0);
        if (extra) rh -= ld_at<WIDE>(reinterpret_cast<const d2_t *>(extra), row >> 1, 0);
      }
      double s0 = 0.0, s1 = 0.0, dv = 0.0;
      for (int k0 = 0; k0 < len; k0 += 8) {
        int oj[8];
        double vj[8];
        d2u_t xj[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { oj[u] = t[8 + k0 + u]; vj[u] = v[k0 + u]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (WIDE) {
            xj[u] = *reinterpret_cast<const d2u_t *>(x + row + oj[u]);
          } else {
            const unsigned byte = ((unsigned)row + (unsigned)oj[u]) << 3;
            xj[u] = *reinterpret_cast<const d2u_t *>(reinterpret_cast<const char *>(x) + byte);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (k0 + u < len) {
            s0 += vj[u] * xj[u].x;
            s1 += vj[u] * xj[u].y;
            if (MODE == 2 && oj[u] == 0) dv = vj[u];
          }
      }
      d2_t out;
      if (MODE == 0) {
        out.x = (beta == 0.0) ? alpha * s0 : alpha * s0 + beta * yold.x;
        out.y = (beta == 0.0) ? alpha * s1 : alpha * s1 + beta * yold.y;
      } else if (MODE == 2) {
        const double r0 = rh.x - s0, r1 = rh.y - s1;
        out.x = xr.x + r0 / dv;
        out.y = xr.y + r1 / dv;
        acc0 += r0 * r0;
        acc0 += r1 * r1;
      } else {
        out.x = s0;
        out.y = s1;
        acc0 += s0 * xr.x;
        acc1 += xr.x * xr.x;
        acc0 += s1 * xr.y;
        acc1 += xr.y * xr.y;
      }
      if (WIDE) *reinterpret_cast<d2_t *>(y + row) = out;
      else *reinterpret_cast<d2_t *>(reinterpret_cast<char *>(y) + ((unsigned)row << 3)) = out;
    } else {
      one_row(row, pidA);
      if (two) one_row(row + 1, pidB);
    }
  }
  if (MODE != 0) {
    __syncthreads();   // the value table doubles as reduction scratch
    double w[2] = {acc0, acc1};
    block_sum<2, kThreads>(w, sv);
    if (threadIdx.x == 0) {
      slab[0 * kSlabStride + blockIdx.x] = w[0];
      if (MODE == 1) slab[1 * kSlabStride + blockIdx.x] = w[1];
    }
  }
}

template <int MODE>
static int launch_pattern(const mk_matrix *A, double alpha, const double *x, double beta, double *y, double *slab,
                          const int *done, int *nblocks, const double *rhs = nullptr, const double *extra = nullptr) {
  Context &c = ctx();
  if (!A->vtab) return fail(MK_ERR_UNSUPPORTED, "spmv: the pattern kernel needs a matrix whose rows share offsets and values");
  const int P8 = A->p8;
  const size_t lds = (size_t)std::max(A->npat * P8, 8) * 8 + (size_t)A->npat * (P8 + 8) * 4;
  const int64_t nchunks = (A->nrows + kPatternChunkRows - 1) / kPatternChunkRows;
  static const int env_wide = getenv("MK_PATTERN_WIDE") ? atoi(getenv("MK_PATTERN_WIDE")) : 0;   // tests: force 64-bit addressing
  const bool wide = env_wide || (A->nrows + 8) * 8 >= (int64_t)1 << 32 || A->ncols * 8 >= (int64_t)1 << 32;
  static const int env_bpc = getenv("MK_PATTERN_BPC") ? atoi(getenv("MK_PATTERN_BPC")) : 0;
  int per_cu = env_bpc > 0 ? env_bpc : A->pattern_bpc[MODE];
  if (per_cu <= 0) {
    int occ = 0;
    hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spmv_pattern_kernel<MODE, false>, kThreads, lds);
    per_cu = (oe == hipSuccess && occ > 0) ? std::min(occ, 8) : 4;
    A->pattern_bpc[MODE] = per_cu;
  }
  int64_t blocks = std::min<int64_t>((nchunks + 7) / 8 * 8, (int64_t)c.num_cus * per_cu);
  blocks = std::max<int64_t>(8, blocks / 8 * 8);
  if (blocks > kSlabStride) blocks = kSlabStride;
  if (nblocks) *nblocks = (int)blocks;
  if (dry_launch()) return MK_OK;   // grid size only (mk_spmv_dot_static checks it against its scratch before anything is enqueued)
  // plane-tiled traversal when the operator couples rows more than ~1.5 MB of x apart (3D grids):
  // tile_p = that stride in chunks, tile_t = chunks of one plane walked before moving to the next plane
  static const int env_tile = getenv("MK_PATTERN_TILE") ? atoi(getenv("MK_PATTERN_TILE")) : 0;
  int tile_p = 0, tile_t = 0;
  if (env_tile > 0 && A->max_off % kPatternChunkRows == 0 && A->max_off / kPatternChunkRows >= 4 * env_tile && 24.0 * A->max_off > 1.5e6) {
    tile_p = A->max_off / kPatternChunkRows;
    tile_t = env_tile;
  }
  if (wide)
    hipLaunchKernelGGL((spmv_pattern_kernel<MODE, true>), dim3((unsigned)blocks), dim3(kThreads), lds, c.stream, A->nrows, A->pat,
                       A->otab8, A->vtab8, A->npat, P8, x, y, alpha, beta, slab, done, rhs, extra, tile_p, tile_t);
  else
    hipLaunchKernelGGL((spmv_pattern_kernel<MODE, false>), dim3((unsigned)blocks), dim3(kThreads), lds, c.stream, A->nrows, A->pat,
                       A->otab8, A->vtab8, A->npat, P8, x, y, alpha, beta, slab, done, rhs, extra, tile_p, tile_t);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// -------------------------------------------------------------------- sweep kernel
// Pattern matrices on grids: every row is a sub-list of ONE base pattern whose offsets split as
// a * P + b (a in {-1,0,1}: previous / same / next plane, |b| small).  Block (cp, seg) owns rows
// [cp*C, cp*C + C) of every plane of its segment and walks plane by plane; the x entries of planes
// k-1, k, k+1 around its rows sit in an LDS ring of four slabs, filled by coalesced 16-byte loads
// that run three planes ahead in registers.  Every gather becomes an LDS read: the pattern kernel
// kept ~7 x 8 B per row in flight in registers and was bound by latency x register file, here one
// 8-byte stream per row feeds LDS and HBM sees x once.  Rows and entries are summed in the order
// of the CSR arrays (a sub-list keeps the base's order) -> bitwise identical to the serial reference.
// Cache policy of the sweep kernel's global accesses (NT).  Streams that are touched once per launch (pattern ids,
// the per-row vectors r / x / rhs / 1/diag, every store) are nontemporal so they do not push the window lines out
// of L2: neighbouring blocks re-read each other's halo rows there.  Measured at 512^3 (CG iteration / SpMV+dot):
// all default 1.661 ms / 0.470 ms; nontemporal stores 1.647 / 0.452; + nontemporal single-use loads 1.533 / 0.447;
// window loads nontemporal as well: 1.856 / 0.539 (the halos miss) -- so the window keeps the default policy.
// Only when the vectors outgrow the 256 MB Infinity Cache: below ~10 M rows the next kernel finds them there and
// the default policy wins by 5-7 % (128^3, 192^3, 2048^2), above it nontemporal wins by 9 % (256^3) -- see nt_rows().
template <bool NT, typename V>
__device__ __forceinline__ void sweep_st(V *ptr, V v) {
  if (NT) __builtin_nontemporal_store(v, ptr); else *ptr = v;
}
template <bool NT, typename V>
__device__ __forceinline__ V sweep_ld(const V *ptr) {
  return NT ? __builtin_nontemporal_load(ptr) : *ptr;
}
// m = 0 or -1: t where m is set, s elsewhere (v_bfi_b32 on both halves)
__device__ __forceinline__ double bit_select(int m, double t, double s) {
  const unsigned long long tb = (unsigned long long)__double_as_longlong(t), sb = (unsigned long long)__double_as_longlong(s);
  unsigned lo, hi;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(m), "v"((unsigned)tb), "v"((unsigned)sb));
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(m), "v"((unsigned)(tb >> 32)), "v"((unsigned)(sb >> 32)));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#define MK_SWEEP_ST(ptr, v) sweep_st<NT>((ptr), (v))
#define MK_SWEEP_LD(ptr) sweep_ld<NT>(ptr)
#define MK_SWEEP_LDW(ptr) (*(ptr))
template <int MODE, int NL, int T, bool NT>   // T threads, chunk = 2 T rows; NL = 16-byte loads per thread per slab = ceil(sw / 2T)
__global__ __launch_bounds__(T, T == 512 ? 2 : 1) void spmv_sweep_kernel(
    const SweepArgs S, const SweepEntry *__restrict__ etab, int64_t nrows, int64_t ncols, const unsigned short *__restrict__ pat,
    const unsigned *__restrict__ pmask, int npat, const double *__restrict__ x, double *__restrict__ y, double alpha, double beta, double *__restrict__ slab,
    const int *__restrict__ done, const double *__restrict__ rhs, const double *__restrict__ extra,
    PcgState *st, const double *__restrict__ slab_in, int nblocks_in, int stride_in, double *__restrict__ xupd,
    int flags, double *__restrict__ norms, const unsigned short *__restrict__ perm, double *__restrict__ y2) {
  // MODE 0: y = alpha A x + beta y.  1: y = A x, partials of <y,x>, <x,x>.  2: Jacobi sweep.
  // MODE 3 (fused CG, first kernel of a step): partials of <A x, x>, <x,x> only, nothing stored.
  // MODE 4 (second kernel; x is CG's p): alpha = gamma / <A x, x> from slab_in (K2's prologue), then
  //         rhs (= y = the residual r) <- r - alpha A x, xupd += alpha x (the solution update: x is in the window
  //         anyway), partials of <r,z>, <r,r>, <xupd,xupd>; `extra` = 1/diag when the Jacobi preconditioner is on.
  // MODE 5 (first kernel of every step but the first): K3 rides in the window fill.  Prologue = K3's (gamma', the
  //         convergence test, beta from slab_in); the window is filled with p' = z + beta x (z = alpha * rhs, alpha = 1
  //         or the uniform 1/diag) instead of x, the block's own rows of p' go to y, then as MODE 3 on p'.
  //         flags: bit 0 two_norm, bit 1 rel_change.
  // MODE 7, 8 (MODE 4 with the solution update halved): x += alpha p is 16 of the step's 68 B per row.  Even steps (MODE 7) leave
  //         x alone (the update stays pending: state record x_pending / pbuf, applied after the loop if the solve ends there);
  //         odd steps (MODE 8) add both, x += alpha_prev p_prev + alpha p, with p_prev = (p - z) / beta rebuilt from what the sweep
  //         holds anyway (p in the window, r before its update, beta of this step's K3): same sum, x rounded once instead of
  //         twice.  <x,x> of a MODE 7 step is the last one known (rel_change looks at it only once the residual test has passed).
  //         flags: bit 3 parity of the step (alpha by parity, as in MODE 6), bit 6 which p buffer x is the window of.
  // MODE 6 (single-reduction CG step, opt-in): MODE 5's window fill and MODE 4's body in ONE kernel behind ONE reduction.
  //         slab_in: <A r, r>, <r,z>, <r,r>, <x,x>, <p,p> (all-reduced).  Prologue: the convergence test of the step before
  //         (K3's), beta = gamma' / gamma, then alpha = gamma' / (zscale^2 <A r, r> - beta gamma' / alpha_old) -- the
  //         Chronopoulos-Gear identity for <A p', p'> with p' = z + beta p, so the product A p' needs no reduction of its
  //         own.  Window = p' (x: old p, rhs: r, alpha: zscale), own rows of p' -> y; then rhs_new = r - alpha (A p' +
  //         extra) -> y2 (r is read through the window of OTHER blocks, so it cannot be updated in place), xupd += alpha
  //         p', partials of <r,z>, <r,r>, <xupd,xupd>, <p',p'>.  flags: bit 2 first step (no test, beta = 0), bit 3 parity
  //         of the step (alpha and gamma of the last step are read from st->sr_*[parity ^ 1], the new ones written to
  //         [parity]: other blocks of this launch still read the old ones).
  if (done != nullptr && *done != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int C = 2 * T, NS = kSweepNS;
  const int sw = S.sw;
  double *ring = reinterpret_cast<double *>(smem_raw);            // NS slabs of sw entries
  unsigned *smask = reinterpret_cast<unsigned *>(ring + NS * sw);   // npat
  const int tid = threadIdx.x;
  double acg = 0.0;   // MODE 6: the step's alpha (`alpha` keeps zscale for the window fill)
  constexpr bool kM4 = (MODE == 4 || MODE == 7 || MODE == 8);
  constexpr bool kM6 = (MODE == 6 || MODE == 9 || MODE == 10);   // 9 / 10: MODE 6 with the halved solution update (as 7 / 8)
  double al_prev = 0.0, inv_beta = 0.0, xx_last = 0.0;   // MODE 7 / 8
  if (MODE >= 3) {
    if (block_flag_raised(&st->done)) return;
    if (kM4) {   // K2's prologue
      const bool leader = (blockIdx.x == 0 && tid == 0);
      double d[2];
      reduce_slab<2, T>(slab_in, nblocks_in, stride_in, d, ring);
      const double sdotp = d[0], gamma = st->gamma;
      const int par4 = (flags >> 3) & 1;
      if (MODE == 8) { al_prev = st->sr_alpha[par4 ^ 1]; inv_beta = 1.0 / st->beta_last; }
      if (MODE == 7) xx_last = st->xx_last;
      __syncthreads();
      if (sdotp == 0.0) {
        if (leader) { st->iter += 1; st->breakdown = 1; st->done = 2; }
        return;
      }
      alpha = gamma / sdotp;
      if (!(alpha > DBL_MIN)) {
        if (leader) { st->iter += 1; st->breakdown = 2; st->done = 2; }
        return;
      }
      if (leader) {
        st->alpha = alpha; st->gamma_old = gamma; st->pp = d[1]; st->sdotp = sdotp; st->iter += 1;
        if (MODE == 7) { st->sr_alpha[par4] = alpha; st->x_pending = 1; st->pbuf = (flags >> 6) & 1; }
        if (MODE == 8) { st->sr_alpha[par4] = alpha; st->x_pending = 0; }
      }
    }
    if (MODE == 5) {   // K3's prologue
      const bool leader = (blockIdx.x == 0 && tid == 0);
      double d[3];
      reduce_slab<3, T>(slab_in, nblocks_in, stride_in, d, ring);
      const double gamma_new = d[0];
      const double i_prod = (flags & 1) ? d[1] : d[0];
      const double xx = d[2];
      const double bi_prod = st->bi_prod, eps = st->eps, al = st->alpha, pp = st->pp, gamma_old = st->gamma_old;
      __syncthreads();
      bool conv = (i_prod / bi_prod < eps);
      if (conv && (flags & 2) && i_prod > 0.0) {
        const double ratio = al * al * pp / xx;
        if (ratio >= eps) conv = false;
      }
      const bool tiny_gamma = (gamma_new < DBL_MIN) && (-gamma_new < DBL_MIN);
      if (leader) {
        st->gamma = gamma_new;
        st->i_prod = i_prod;
        if (norms) norms[st->iter] = sqrt(i_prod / bi_prod);
        if (conv) { st->converged = 1; st->done = 1; }
        else if (tiny_gamma) { st->breakdown = 3; st->done = 2; }
        st->xx_last = xx;                          // (MODE 7 / 8 of this step read them: a later launch)
        st->beta_last = gamma_new / gamma_old;
      }
      if (conv || tiny_gamma) return;
      beta = gamma_new / gamma_old;
    }
    if (kM6) {
      const bool leader = (blockIdx.x == 0 && tid == 0);
      double d[5];
      reduce_slab<5, T>(slab_in, nblocks_in, stride_in, d, ring);
      const int par = (flags >> 3) & 1;
      const bool first = (flags & 4) != 0;
      const double bi_prod = st->bi_prod, eps = st->eps, al_old = st->sr_alpha[par ^ 1], gamma_old = st->sr_gamma[par ^ 1];
      const double gamma_new = first ? st->gamma : d[1];
      __syncthreads();
      // <x,x>: steps that left x alone (MODE 9) hand on zeros; the last value known stands in (every rank holds the same)
      const double xx6 = (d[3] != 0.0) ? d[3] : st->xx_last;
      if (!first) {
        const double i_prod = (flags & 1) ? d[2] : d[1];
        bool conv = (i_prod / bi_prod < eps);
        if (conv && (flags & 2) && i_prod > 0.0) {
          const double ratio = al_old * al_old * d[4] / xx6;
          if (ratio >= eps) conv = false;
        }
        const bool tiny_gamma = (gamma_new < DBL_MIN) && (-gamma_new < DBL_MIN);
        if (leader) {
          st->gamma = gamma_new;
          st->i_prod = i_prod;
          if (norms) norms[st->iter] = sqrt(i_prod / bi_prod);
          if (conv) { st->converged = 1; st->done = 1; }
          else if (tiny_gamma) { st->breakdown = 3; st->done = 2; }
        }
        if (conv || tiny_gamma) return;
        beta = gamma_new / gamma_old;
      } else {
        beta = 0.0;
      }
      if (MODE == 10) { al_prev = al_old; inv_beta = 1.0 / beta; }
      const double azz = (alpha * alpha) * d[0];                                   // <A z, z>, z = zscale r
      const double sdotp = first ? azz : azz - beta * gamma_new / al_old;          // = <A p', p'>
      if (sdotp == 0.0) {
        if (leader) { st->iter += 1; st->breakdown = 1; st->done = 2; }
        return;
      }
      acg = gamma_new / sdotp;
      if (!(acg > DBL_MIN)) {
        if (leader) { st->iter += 1; st->breakdown = 2; st->done = 2; }
        return;
      }
      if (leader) {
        st->alpha = acg; st->gamma_old = gamma_new; st->sdotp = sdotp; st->iter += 1;
        st->sr_alpha[par] = acg; st->sr_gamma[par] = gamma_new;
        st->xx_last = xx6;
        if (MODE == 9) { st->x_pending = 1; st->pbuf = (flags >> 6) & 1; }
        if (MODE == 10) st->x_pending = 0;
      }
    }
    __syncthreads();   // ring was reduction scratch
  }
  for (int i = tid; i < npat; i += T) smask[i] = pmask[i];
  const int nb = gridDim.x;   // multiple of 8; blocks of one XCD take neighbouring chunks of a plane (shared halos in its L2)
  const int lid = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  if (lid < S.ncp * S.nseg) {
    // all row / column indices fit 29 bits here (plan_sweep): 32-bit arithmetic throughout, one v_med3 per clamp
    const int nr = (int)nrows;
    const int cp = lid % S.ncp, seg = lid / S.ncp;
    const int k0 = S.kb + (int)((int64_t)seg * S.K / S.nseg), k1 = S.kb + (int)((int64_t)(seg + 1) * S.K / S.nseg);
    const int r0 = cp * C;
    const int rows_here = min(C, S.P - r0);
    // lw: the lane's place in the window fill (always 2 tid).  lr: the pair of rows it owns.  With a permutation table
    // (rows of more than 8 entries) the pairs that hold boundary rows of the grid are dealt to the first lanes of the
    // chunk, so that all other waves own complete rows only and take the multiply-add loop without the selects.
    const int lw = 2 * tid;
    const int lr = perm ? 2 * (int)perm[cp * T + tid] : lw;
    const double dvb = S.diag >= 0 ? S.dv : 0.0;

    // All loads that run ahead are UNCONDITIONAL (addresses clamped into the arrays instead of predicated):
    // the compiler counts outstanding loads to wait for exactly the one a step needs; a load that may or
    // may not have been issued would force it to wait for everything.  Clamped positions hold entries no
    // row refers to (their mask bit is clear) or are never read.
    const int xlast = ((int)ncols - 1) & ~1;   // x[xlast], x[xlast+1]: inside the allocation (slack behind ncols)
    const int rlast = (nr - 1) & ~1;
    constexpr bool kFill = (MODE == 5 || kM6);   // the window is p' = z + beta p, formed while it is filled
    constexpr int NLR = kFill ? NL : 1;   // second window stream (the residual) only then
    auto load_slab = [&](int j, d2_t(&reg)[NL], d2_t(&regr)[NLR]) {   // window of plane j
      const int g0 = min(j, k1) * S.P + r0 - S.wlo + lw;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        const int g = min(max(g0 + i * 2 * T, 0), xlast);
        reg[i] = MK_SWEEP_LDW(reinterpret_cast<const d2_t *>(reinterpret_cast<const char *>(x) + ((unsigned)g << 3)));
        if (kFill) regr[i] = MK_SWEEP_LDW(reinterpret_cast<const d2_t *>(reinterpret_cast<const char *>(rhs) + ((unsigned)g << 3)));
      }
    };
    // A slab is stored de-interleaved: window entry e at [e / 2] of the first half (e even) or of the second half (e odd).
    // A lane owns rows g, g + 1 and reads x[g + o], x[g + 1 + o] per entry: one from each half, and across the wave both
    // reads are 64 consecutive doubles (entries in order: the two reads were 16 bytes apart from lane to lane, a 2-way bank
    // conflict on every one of the 2 x 27 reads of a 27-point row, which bounds that kernel).
    const int half = sw >> 1;
    auto write_slab = [&](int slot, int j, const d2_t(&reg)[NL], const d2_t(&regr)[NLR]) {
      double *dst = ring + slot * sw + tid;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        const int wi = lw + i * 2 * T;   // position in the window
        if (wi < sw) {
          d2_t v = reg[i];
          if (kFill) {
            d2_t z = regr[i];
            z.x *= alpha;
            z.y *= alpha;
            v.x = beta * v.x + z.x;   // K3: p = beta p + z
            v.y = beta * v.y + z.y;
            const int own = wi - S.wlo;   // the block's own rows of its own planes go to the new p
            if (j >= k0 && j < k1 && own >= 0 && own < rows_here) {
              const int g = j * S.P + r0 + own;
              if (g + 1 < nr) MK_SWEEP_ST(reinterpret_cast<d2_t *>(reinterpret_cast<char *>(y) + ((unsigned)g << 3)), v);
              else if (g < nr) *reinterpret_cast<double *>(reinterpret_cast<char *>(y) + ((unsigned)g << 3)) = v.x;
            }
          }
          dst[i * T] = v.x;
          dst[i * T + half] = v.y;
        }
      }
    };
    auto row_of = [&](int j) -> int { return j * S.P + r0 + lr; };
    auto live = [&](int j) -> bool { return j < k1 && lr < rows_here && row_of(j) < nr; };
    auto load_ids = [&](int j) -> unsigned {   // pattern ids of rows g, g+1 (g even); the id behind the last row is slack
      const int g = min(row_of(j), rlast);
      return MK_SWEEP_LD(reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(pat) + ((unsigned)g << 1)));
    };
    auto load_pair = [&](const double *p, int j) -> d2_t {   // p[g], p[g+1]; p[nrows] is allocation slack
      const int g = min(row_of(j), rlast);
      return MK_SWEEP_LD(reinterpret_cast<const d2_t *>(reinterpret_cast<const char *>(p) + ((unsigned)g << 3)));
    };
    const double *aux = (MODE == 2 || kM4 || kM6) ? rhs : (MODE == 0 && beta != 0.0) ? y : nullptr;
    constexpr bool kExtra = (MODE == 2 || kM4 || kM6);
    constexpr bool kXupd = (MODE == 4 || MODE == 8 || MODE == 6 || MODE == 10);
    // MODE 2, row-block Jacobi: `extra` (the off-rank product) is zero on every row that touches no ghost.  The caller names
    // the largest run of such rows [gap0, gap1) (in nblocks_in / stride_in, which this mode has no other use for): lanes
    // inside it all read the pair at gap0 -- the load stays unconditional, HBM sees one line instead of 8 B per row.
    // (MODE 6: the same for the off-rank part of A p', run named in S.gap0 / S.gap1)
    const int gap0 = (MODE == 2) ? nblocks_in : kM6 ? S.gap0 : 0, gap1 = (MODE == 2) ? stride_in : kM6 ? S.gap1 : 0;
    auto load_extra = [&](int j) -> d2_t {
      int g = min(row_of(j), rlast);
      if (MODE == 2 || kM6) g = (g >= gap0 && g + 1 < gap1) ? gap0 : g;
      return MK_SWEEP_LD(reinterpret_cast<const d2_t *>(reinterpret_cast<const char *>(extra) + ((unsigned)g << 3)));
    };

    // Three register stages, each used in place: stage (k mod 3) holds slab k+2 at step k, is written to the
    // ring and at once re-loaded with slab k+5 -- no register-to-register rotation (a move of a register
    // that a load is still filling would wait for that load and serialise the pipeline).
    d2_t st0[NL], st1[NL], st2[NL], rt0[NLR], rt1[NLR], rt2[NLR];
    load_slab(k0 - 1, st0, rt0);
    load_slab(k0, st1, rt1);
    load_slab(k0 + 1, st2, rt2);
    write_slab(0, k0 - 1, st0, rt0);
    write_slab(1, k0, st1, rt1);
    write_slab(2, k0 + 1, st2, rt2);
    // issue order as in the steady state (ids/aux of a plane, then the slab two planes further): the compiler
    // derives "how many younger loads may stay in flight" from the worst path into the loop, the prologue included
    d2_t ax0 = {0.0, 0.0}, ax1 = ax0, ax2 = ax0, ex0 = ax0, ex1 = ax0, ex2 = ax0;
    d2_t xu0 = ax0, xu1 = ax0, xu2 = ax0;
    unsigned id0 = load_ids(k0);
    if (aux) ax0 = load_pair(aux, k0);
    if (kExtra && extra) ex0 = load_extra(k0);
    if (kXupd) xu0 = load_pair(xupd, k0);
    load_slab(k0 + 2, st0, rt0);
    unsigned id1 = load_ids(k0 + 1);
    if (aux) ax1 = load_pair(aux, k0 + 1);
    if (kExtra && extra) ex1 = load_extra(k0 + 1);
    if (kXupd) xu1 = load_pair(xupd, k0 + 1);
    load_slab(k0 + 3, st1, rt1);
    unsigned id2 = load_ids(k0 + 2);
    if (aux) ax2 = load_pair(aux, k0 + 2);
    if (kExtra && extra) ex2 = load_extra(k0 + 2);
    if (kXupd) xu2 = load_pair(xupd, k0 + 2);
    load_slab(k0 + 4, st2, rt2);
    __syncthreads();

    auto step = [&](int k, d2_t(&st)[NL], d2_t(&rt)[NLR], unsigned &idr, d2_t &axr, d2_t &exr, d2_t &xur) {
      const unsigned id0 = idr;
      const d2_t ax0 = axr, ex0 = exr, xu0 = xur;
      const int jb = (k - k0 + 1) & (NS - 1);   // ring slot of plane k
      if (live(k)) {
        const int g = row_of(k);
        const bool two = g + 1 < nr;
        const unsigned mA = smask[id0 & 0xffffu];
        const unsigned mB = two ? smask[min(id0 >> 16, (unsigned)npat - 1u)] : 0u;
        const char *wl = reinterpret_cast<const char *>(ring) + (lr >> 1) * 8;
        double s0 = 0.0, s1 = 0.0;
        int u0 = 0;
        if (perm != nullptr) {
          // every row of the wave holds the whole base pattern: whole batches without the selects (same operations on
          // the same operands in the same order)
          const unsigned fullm = S.len >= 32 ? 0xffffffffu : (1u << S.len) - 1u;
          if (__all(mA == fullm && mB == fullm)) {
            for (; u0 + 8 <= S.len; u0 += 8) {
              typedef int i16f_t __attribute__((ext_vector_type(16)));
              const i16f_t *eb = reinterpret_cast<const i16f_t *>(etab + jb * 32 + u0);
              const i16f_t b0 = eb[0], b1 = eb[1];
              int off[8], off1[8];
              double vv[8];
#define MK_ENTF(U, B, Q)                                                                                  \
  off[U] = B[4 * Q];                                                                                      \
  off1[U] = B[4 * Q + 1];                                                                                 \
  vv[U] = __longlong_as_double((long long)(((unsigned long long)(unsigned)B[4 * Q + 3] << 32) | (unsigned)B[4 * Q + 2]));
              MK_ENTF(0, b0, 0) MK_ENTF(1, b0, 1) MK_ENTF(2, b0, 2) MK_ENTF(3, b0, 3)
              MK_ENTF(4, b1, 0) MK_ENTF(5, b1, 1) MK_ENTF(6, b1, 2) MK_ENTF(7, b1, 3)
#undef MK_ENTF
              double p0[8], p1[8];
#pragma unroll
              for (int u = 0; u < 8; u++) {
                p0[u] = *reinterpret_cast<const double *>(wl + off[u]);
                p1[u] = *reinterpret_cast<const double *>(wl + off1[u]);
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int u = 0; u < 8; u++) {
                asm("v_fma_f64 %0, %1, %2, %3" : "=v"(s0) : "s"(vv[u]), "v"(p0[u]), "v"(s0));
                asm("v_fma_f64 %0, %1, %2, %3" : "=v"(s1) : "s"(vv[u]), "v"(p1[u]), "v"(s1));
              }
            }
          }
        }
        for (; u0 + 4 < S.len; u0 += 8) {   // (four entries or fewer left: the half batch below)
          // Eight entries per batch by two 64-byte scalar loads (scalar cache).  The table is stored once per
          // ring slot with the LDS offsets of that slot already worked out, which also makes the compiler
          // re-load it per step instead of keeping all 32 entries in SGPRs for the whole kernel (it did, and
          // spilled them: 487 v_readlane per step).  Branch-free: a row takes entry u iff bit u of its mask
          // is set (padding entries have no bit in any mask); the sum is SELECTED, not multiplied by 0/1.
          // (A separate multiply-add chain without the selects for waves whose rows all hold the whole pattern was
          // measured: 27-point CG 0.351 -> 0.345 ms per iteration, 7-point 1.517 -> 1.530: not kept.)
          typedef int i16_t __attribute__((ext_vector_type(16)));
          const i16_t *eb = reinterpret_cast<const i16_t *>(etab + jb * 32 + u0);
          const i16_t b0 = eb[0], b1 = eb[1];
          int off[8], off1[8];
          double vv[8];
#define MK_ENT(U, B, Q)                                                                                   \
  off[U] = B[4 * Q];                                                                                      \
  off1[U] = B[4 * Q + 1];                                                                                 \
  vv[U] = __longlong_as_double((long long)(((unsigned long long)(unsigned)B[4 * Q + 3] << 32) | (unsigned)B[4 * Q + 2]));
          MK_ENT(0, b0, 0) MK_ENT(1, b0, 1) MK_ENT(2, b0, 2) MK_ENT(3, b0, 3)
          MK_ENT(4, b1, 0) MK_ENT(5, b1, 1) MK_ENT(6, b1, 2) MK_ENT(7, b1, 3)
#undef MK_ENT
          double p0[8], p1[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            p0[u] = *reinterpret_cast<const double *>(wl + off[u]);
            p1[u] = *reinterpret_cast<const double *>(wl + off1[u]);
          }
          // 27-point rows are bound by vector issue (91 % of the slots), so the select is spelled out: m = bit u of the
          // row's mask spread over a register (v_bfe_i32), then one v_bfi_b32 per half, s = (m & t) | (~m & s); and the
          // three-operand v_fma_f64 leaves s alone.  10 vector instructions per entry and row pair where the compiler's
          // and / compare / v_cndmask x 2 / v_mov_b64 (its fmac overwrites the sum) took 16.
          const int ka = (int)(mA >> u0), kb = (int)(mB >> u0);
          __builtin_amdgcn_sched_barrier(0);   // all sixteen reads are in flight before the first multiply-add waits
#pragma unroll
          for (int u = 0; u < 8; u++) {
            double t0, t1;
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t0) : "s"(vv[u]), "v"(p0[u]), "v"(s0));
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t1) : "s"(vv[u]), "v"(p1[u]), "v"(s1));
            const int ma = __builtin_amdgcn_sbfe(ka, u, 1), mb = __builtin_amdgcn_sbfe(kb, u, 1);
            s0 = bit_select(ma, t0, s0);
            s1 = bit_select(mb, t1, s1);
          }
        }
        if (u0 < S.len) {   // the last one to four entries (27 = 3 x 8 + 3): half a batch, half the reads and selects
          typedef int i16h_t __attribute__((ext_vector_type(16)));
          const i16h_t b0 = *reinterpret_cast<const i16h_t *>(etab + jb * 32 + u0);
          double p0[4], p1[4], vv[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            p0[u] = *reinterpret_cast<const double *>(wl + b0[4 * u]);
            p1[u] = *reinterpret_cast<const double *>(wl + b0[4 * u + 1]);
            vv[u] = __longlong_as_double((long long)(((unsigned long long)(unsigned)b0[4 * u + 3] << 32) | (unsigned)b0[4 * u + 2]));
          }
          const int ka = (int)(mA >> u0), kb = (int)(mB >> u0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; u++) {
            double t0, t1;
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t0) : "s"(vv[u]), "v"(p0[u]), "v"(s0));
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t1) : "s"(vv[u]), "v"(p1[u]), "v"(s1));
            s0 = bit_select(__builtin_amdgcn_sbfe(ka, u, 1), t0, s0);
            s1 = bit_select(__builtin_amdgcn_sbfe(kb, u, 1), t1, s1);
          }
        }
        d2_t out;
        if (MODE == 0) {
          out.x = (beta == 0.0) ? alpha * s0 : alpha * s0 + beta * ax0.x;
          out.y = (beta == 0.0) ? alpha * s1 : alpha * s1 + beta * ax0.y;
        } else {
          const double *wc = reinterpret_cast<const double *>(wl) + jb * sw + (S.wlo >> 1);   // x[g], x[g+1]
          const double xa = wc[0], xb = wc[half];
          if (MODE == 2) {
            const double ra = (ax0.x - ex0.x) - s0, rb = (ax0.y - ex0.y) - s1;
            const double da = (S.diag >= 0 && ((mA >> S.diag) & 1u)) ? dvb : 0.0;
            const double db = (S.diag >= 0 && ((mB >> S.diag) & 1u)) ? dvb : 0.0;
            out.x = xa + ra / da;
            out.y = xb + rb / db;
            acc0 += ra * ra;
            if (two) acc0 += rb * rb;
          } else if (kM4) {
            if (MODE != 7) {
              d2_t xv = xu0;
              if (MODE == 8) {   // the pending update of the step before: p_prev = (p - z) / beta, z = r (scaled) before this step's update
                const double zpa = (beta != 0.0) ? ax0.x * beta : ax0.x, zpb = (beta != 0.0) ? ax0.y * beta : ax0.y;
                xv.x += al_prev * ((xa - zpa) * inv_beta);
                xv.y += al_prev * ((xb - zpb) * inv_beta);
              }
              xv.x += alpha * xa;
              xv.y += alpha * xb;
              acc2 += xv.x * xv.x;
              if (two) {
                acc2 += xv.y * xv.y;
                MK_SWEEP_ST(reinterpret_cast<d2_t *>(reinterpret_cast<char *>(xupd) + ((unsigned)g << 3)), xv);
              } else {
                *reinterpret_cast<double *>(reinterpret_cast<char *>(xupd) + ((unsigned)g << 3)) = xv.x;
              }
            }
            double ra = ax0.x, rb = ax0.y;
            ra -= alpha * s0;
            rb -= alpha * s1;
            out.x = ra;
            out.y = rb;
            // z = C r: 1/diag per row from `extra`, or one number (beta != 0) when the diagonal is uniform
            const double za = extra ? ra * ex0.x : (beta != 0.0 ? ra * beta : ra), zb = extra ? rb * ex0.y : (beta != 0.0 ? rb * beta : rb);
            acc0 += ra * za;
            acc1 += ra * ra;
            if (two) { acc0 += rb * zb; acc1 += rb * rb; }
          } else if (kM6) {
            acc3 += xa * xa;
            if (two) acc3 += xb * xb;
            if (MODE != 9) {
              d2_t xv = xu0;
              if (MODE == 10) {   // the pending update of the step before: p_prev = (p' - z) / beta, z = zscale r before this step's update
                xv.x += al_prev * ((xa - ax0.x * alpha) * inv_beta);
                xv.y += al_prev * ((xb - ax0.y * alpha) * inv_beta);
              }
              xv.x += acg * xa;
              xv.y += acg * xb;
              acc2 += xv.x * xv.x;
              if (two) {
                acc2 += xv.y * xv.y;
                MK_SWEEP_ST(reinterpret_cast<d2_t *>(reinterpret_cast<char *>(xupd) + ((unsigned)g << 3)), xv);
              } else {
                *reinterpret_cast<double *>(reinterpret_cast<char *>(xupd) + ((unsigned)g << 3)) = xv.x;
              }
            }
            double ra = ax0.x, rb = ax0.y;
            if (extra) { ra -= acg * ex0.x; rb -= acg * ex0.y; }   // off-rank part of A p' first (the default step's order)
            ra -= acg * s0;
            rb -= acg * s1;
            out.x = ra;
            out.y = rb;
            const double za = ra * alpha, zb = rb * alpha;   // z = zscale r (zscale = 1: exact)
            acc0 += ra * za;
            acc1 += ra * ra;
            if (two) { acc0 += rb * zb; acc1 += rb * rb; }
          } else {
            out.x = s0;
            out.y = s1;
            acc0 += s0 * xa;
            acc1 += xa * xa;
            if (two) { acc0 += s1 * xb; acc1 += xb * xb; }
          }
        }
        if (MODE != 3 && MODE != 5) {
          double *dst = kM6 ? y2 : y;   // (MODE 6: y holds the new p, written by the window fill)
          if (two) MK_SWEEP_ST(reinterpret_cast<d2_t *>(reinterpret_cast<char *>(dst) + ((unsigned)g << 3)), out);
          else *reinterpret_cast<double *>(reinterpret_cast<char *>(dst) + ((unsigned)g << 3)) = out.x;
        }
      }
      // slab k+2 (loaded three steps ago) into the slot plane k-2 occupied; nobody reads that slot in this step
      write_slab((k - k0 + 3) & (NS - 1), k + 2, st, rt);
      // look-ahead loads into the registers just consumed: ids / aux of plane k+3, slab k+5
      idr = load_ids(k + 3);
      if (aux) axr = load_pair(aux, k + 3);
      if (kExtra && extra) exr = load_extra(k + 3);
      if (kXupd) xur = load_pair(xupd, k + 3);
      load_slab(k + 5, st, rt);
      __syncthreads();
    };
    for (int k = k0; k < k1; k += 3) {   // straight-line body: steps at or behind k1 compute nothing (live() is false)
      step(k, st0, rt0, id0, ax0, ex0, xu0);
      step(k + 1, st1, rt1, id1, ax1, ex1, xu1);
      step(k + 2, st2, rt2, id2, ax2, ex2, xu2);
    }
  }
  if (MODE != 0) {
    __syncthreads();
    constexpr int NACC = kM6 ? 4 : 3;
    double w[NACC];
    w[0] = acc0; w[1] = acc1; w[2] = acc2;
    if (kM6) w[NACC - 1] = acc3;
    block_sum<NACC, T>(w, ring);
    if (MODE == 7) w[2] = (blockIdx.x == 0) ? xx_last : 0.0;   // <x,x> as last known
    if (threadIdx.x == 0) {
      slab[0 * kSlabStride + blockIdx.x] = w[0];
      if (MODE != 2) slab[1 * kSlabStride + blockIdx.x] = w[1];
      if (kM4 || kM6) slab[2 * kSlabStride + blockIdx.x] = w[2];
      if (kM6) slab[3 * kSlabStride + blockIdx.x] = w[NACC - 1];
    }
  }
}

// Segments a sweep launch over the planes of S cuts them into (grid = ncp * nseg blocks, rounded up to 8); 0 when even one
// segment needs more than max_blocks reduction partials.  Shared by the launcher and by callers that must know the grid
// before they launch (row-block Jacobi: the partials of its interior and boundary launches share one slab).
static int sweep_segments(const mk_matrix *A, int mode, const SweepArgs &S, int max_blocks) {
  Context &c = ctx();
  const size_t lds = (size_t)kSweepNS * S.sw * 8 + (size_t)A->npat * 4;
  static const int env_bpc = getenv("MK_SWEEP_BPC") ? atoi(getenv("MK_SWEEP_BPC")) : 0;
  int per_cu = env_bpc > 0 ? env_bpc : A->sweep_bpc[mode];
  if (per_cu <= 0) {
    const int occ = (S.C == 2048) ? 1 : (S.C == 1024) ? 2 : (int)std::min<size_t>(3, (size_t)(160 * 1024) / std::max<size_t>(lds, 1));   // LDS-bound: 160 KB per CU
    per_cu = occ > 0 ? std::min(occ, 8) : 2;
    A->sweep_bpc[mode] = per_cu;
  }
  // segments: fill the chip, but keep at least 8 planes per segment (each segment re-reads two halo planes)
  static const int env_nseg = getenv("MK_SWEEP_NSEG") ? atoi(getenv("MK_SWEEP_NSEG")) : 0;
  const int64_t target = (int64_t)c.num_cus * per_cu;
  // two rounds of resident blocks (measured at 512^3: 1 segment 647 us, 3 segments 542 us, 6 segments 540 us)
  int nseg = env_nseg > 0 ? env_nseg : (int)std::max<int64_t>(1, (2 * target + S.ncp - 1) / S.ncp);
  // ... unless the grid can be exactly one round with 8 waves on every CU (16 for the 1024-thread variant): then all
  // blocks walk the same plane front and no CU holds a second segment.  Measured CG iteration, ns per row,
  // exact fit against the best of the multi-round counts: 512^3 0.0124 / 0.0133, 512x512x64 0.0130 / 0.0140,
  // 256^3 0.0135 / 0.0148.  A grid that misses the fit (384^3: 144 chunks per plane) keeps the rule above:
  // 288 blocks on 256 CUs was the slowest choice of all.
  static const int no_fit = getenv("MK_SWEEP_NO_FIT") ? atoi(getenv("MK_SWEEP_NO_FIT")) : 0;
  // Only the two CG sweeps (3 and 5 streams each) gain; SpMV+dot (x in, y out) measured 4-10 % SLOWER with the fit
  // (512^3 490 against 469 us, 27-point 256^3 159 against 144 us, banded 2^26 280 against 259 us).
  static const int fit3 = getenv("MK_SWEEP_FIT3") ? atoi(getenv("MK_SWEEP_FIT3")) : 0;   // A/B: the dots-only sweep as one round too
  if (env_nseg <= 0 && !no_fit && (mode == 4 || mode == 5 || mode == 6 || mode == 7 || mode == 8 || mode == 9 || mode == 10 || (mode == 3 && fit3))) {
    const int64_t fit = (int64_t)c.num_cus * std::max(1, 1024 / S.C);
    const int64_t k = fit / S.ncp;
    if (k >= 1 && k * S.ncp * 10 >= fit * 9 && S.K / k >= 8) nseg = (int)k;
  }
  nseg = std::min(nseg, std::max(1, S.K / 8));
  nseg = std::max(1, std::min(nseg, S.K));
  max_blocks = std::min(max_blocks, kSlabStride - 8);
  while ((S.ncp * (int64_t)nseg + 7) / 8 * 8 > max_blocks && nseg > 1) nseg--;
  if ((S.ncp * (int64_t)nseg + 7) / 8 * 8 > max_blocks) return 0;
  return nseg;
}

template <int MODE>
static int launch_sweep(const mk_matrix *A, double alpha, const double *x, double beta, double *y, double *slab,
                        const int *done, int *nblocks, const double *rhs = nullptr, const double *extra = nullptr,
                        PcgState *st = nullptr, const double *slab_in = nullptr, int nblocks_in = 0, int stride_in = kSlabStride,
                        double *xupd = nullptr, int flags = 0, double *norms = nullptr, int plane_begin = 0, int plane_end = 0,
                        int max_blocks = kSlabStride, double *y2 = nullptr, int gap0 = 0, int gap1 = 0) {
  Context &c = ctx();
  if (!A->sweep || A->sweep_streamed) return fail(MK_ERR_UNSUPPORTED, "spmv: the sweep kernel needs rows that are sub-lists of one (offsets, values) pattern with offsets a*P + b");
  SweepArgs S = *A->sweep;
  if (plane_end > plane_begin) {   // a caller's sub-range of the planes (row-block Jacobi: interior / boundary launches)
    if (plane_begin < 0 || plane_end > S.K) return fail(MK_ERR_INVALID, "spmv: planes [%d, %d) of %d", plane_begin, plane_end, S.K);
    S.kb = plane_begin;
    S.K = plane_end - plane_begin;
  }
  const size_t lds = (size_t)kSweepNS * S.sw * 8 + (size_t)A->npat * 4;
  const int nseg = sweep_segments(A, MODE, S, max_blocks);
  if (nseg < 1) return fail(MK_ERR_UNSUPPORTED, "spmv: sweep grid of %d chunks per plane does not fit %d reduction partials", S.ncp, max_blocks);
  S.nseg = nseg;
  S.gap0 = gap0;
  S.gap1 = gap1;
  const int blocks = (S.ncp * nseg + 7) / 8 * 8;
  if (nblocks) *nblocks = blocks;
  if (dry_launch()) return MK_OK;   // grid size only (mk_spmv_dot_static checks it against its scratch before anything is enqueued)
  const int T = S.C / 2;
  const int nl = std::max(2, (S.sw + 2 * T - 1) / (2 * T));
  const bool nt = nt_rows(A->nrows);
#define MK_LAUNCH_SW(NLL, TT)                                                                                              \
  {                                                                                                                        \
    auto kern = nt ? spmv_sweep_kernel<MODE, NLL, TT, true> : spmv_sweep_kernel<MODE, NLL, TT, false>;                     \
    static size_t lds_max[2] = {48 * 1024, 48 * 1024};   /* per kernel instance: only ever raised (the attribute is the kernel's, not the matrix's) */ \
    if (lds > lds_max[nt ? 1 : 0]) {                                                                                       \
      MK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      lds_max[nt ? 1 : 0] = lds;                                                                                           \
    }                                                                                                                      \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(TT), lds, c.stream, S, A->etab, A->nrows, A->ncols, A->pat,     \
                       A->pmask, A->npat, x, y, alpha, beta, slab, done, rhs, extra, st, slab_in, nblocks_in, stride_in,  \
                       xupd, flags, norms, A->sweep_perm, y2);                                                             \
  }
  if (T == 256 && nl == 2) MK_LAUNCH_SW(2, 256)
  else if (T == 256 && nl == 3) MK_LAUNCH_SW(3, 256)
  else if (T == 256 && nl == 4) MK_LAUNCH_SW(4, 256)
  else if (T == 512 && nl == 2) MK_LAUNCH_SW(2, 512)
  else if (T == 1024 && nl == 2) MK_LAUNCH_SW(2, 1024)
  else return fail(MK_ERR_UNSUPPORTED, "spmv: sweep window of %d entries for chunks of %d rows", S.sw, S.C);
#undef MK_LAUNCH_SW
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// -------------------------------------------------------------------- split kernel
// Skewed / power-law rows.  Work is cut by ENTRIES, not rows: chunk c owns entries
// [c*CH, (c+1)*CH), so every block streams the same number of bytes whatever the row lengths
// (merge-path style balance).  Pass 1: coalesced 16-byte loads of val/col, gather x, products
// into LDS.  Pass 2: the rows that START in the chunk (chunk_row[c] .. chunk_row[c+1]-1) plus
// the row that enters from the previous chunk are reduced out of LDS, LP lanes per row with LP
// picked per chunk from the number of rows in it.  A row that ends inside the chunk where it
// started is written at once; pieces of rows that cross chunk borders go to carry[2c] (piece
// of the row entering from the left) and carry[2c+1] (piece of the row leaving to the
// right), and split_fixup_kernel folds them in chunk order: fixed order, no atomics.
__global__ void split_chunk_rows_kernel(int64_t nrows, int64_t nnz, const int *__restrict__ rowptr, int nchunks,
                                        int *__restrict__ chunk_row) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c <= nchunks; c += gridDim.x * blockDim.x) {
    if (c == nchunks) { chunk_row[c] = (int)nrows; continue; }
    const int64_t target = (int64_t)c * kSplitChunk;
    int64_t lo = 0, hi = nrows;   // first row with rowptr[row] >= target
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (rowptr[mid] >= target) hi = mid; else lo = mid + 1;
    }
    chunk_row[c] = (int)lo;
  }
}

#ifndef MK_SPLIT_GROUP
#define MK_SPLIT_GROUP 2
#endif
#ifndef MK_LONG_PIECE
#define MK_LONG_PIECE 64
#endif
template <int MODE, bool NT>   // NT: the matrix does not fit the Infinity Cache, it streams past nontemporally and the caches keep x
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(MODE == 1 ? 6 : 8, 8))) void spmv_split_kernel(
    int64_t nrows, int64_t nnz, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const int *__restrict__ chunk_row, int nchunks, const double *__restrict__ x, double *__restrict__ y, double alpha,
    double beta, double *__restrict__ carry, double *__restrict__ slab, const int *__restrict__ done,
    const double *__restrict__ xg) {   // xg: the vector the column indices refer to (x, or x in the relabelled order)
  if (done != nullptr && *done != 0) return;
  __shared__ __attribute__((aligned(16))) double prod[kSplitChunk];
  __shared__ double red[2 * (kThreads / 64)];
  // pieces of more than kLongPiece entries per lane are left to a whole wave each (second part of pass 2): a 1200-entry row among
  // 140 short ones would otherwise be summed by ONE lane, ~80 clocks per LDS read, while the block waits (measured with fem_probe.py skew)
  constexpr int kLongPiece = MK_LONG_PIECE, kLongMax = kSplitChunk / kLongPiece;
  __shared__ int long_q[kLongMax], long_s[kLongMax];
  __shared__ int long_n;
  const int tid = threadIdx.x;
  double acc_yx = 0.0, acc_xx = 0.0;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t e0 = (int64_t)c * kSplitChunk;
    const int64_t e1 = min(e0 + kSplitChunk, nnz);
    if (tid == 0) long_n = 0;   // read again only behind the barrier that follows pass 1
    // pass 1: all matrix loads of the thread first, then all its gathers, then the products -- written with clamped addresses
    // instead of per-piece branches, which made the compiler wait for each piece's loads and gathers in turn (8 round trips
    // per chunk).  Pairs beyond the end of the last chunk re-read its last pair and store zeros.
    {
      typedef int i2_t __attribute__((ext_vector_type(2)));
      constexpr int NP = kSplitChunk / (2 * kThreads);
      const int64_t klast = max(e0, (e1 - 1) & ~(int64_t)1);   // last pair that starts inside the chunk
      constexpr int GP = MK_SPLIT_GROUP;                          // pieces whose loads and gathers are in flight together
      if (e0 >= e1) {   // a matrix without entries: nothing to read, the rows are written as zeros in pass 2
#pragma unroll
        for (int i = 0; i < NP; i++) *reinterpret_cast<double2 *>(prod + (tid + i * kThreads) * 2) = make_double2(0.0, 0.0);
      } else {
#pragma unroll
        for (int i0 = 0; i0 < NP; i0 += GP) {
          d2_t vv[GP];
          i2_t cc[GP];
          double xa[GP], xb[GP];
#pragma unroll
          for (int i = 0; i < GP; i++) {
            const int64_t k = min(e0 + (int64_t)(tid + (i0 + i) * kThreads) * 2, klast);
            vv[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(val + k)) : *reinterpret_cast<const d2_t *>(val + k);
            cc[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const i2_t *>(col + k)) : *reinterpret_cast<const i2_t *>(col + k);
          }
#pragma unroll
          for (int i = 0; i < GP; i++) {
            const int64_t k = min(e0 + (int64_t)(tid + (i0 + i) * kThreads) * 2, klast);
            const int cy = (k + 1 < e1) ? cc[i].y : cc[i].x;       // the slot behind an odd end holds no column
            xa[i] = xg[cc[i].x];
            xb[i] = xg[cy];
          }
#pragma unroll
          for (int i = 0; i < GP; i++) {
            const int64_t k = e0 + (int64_t)(tid + (i0 + i) * kThreads) * 2;
            const double p0 = (k < e1) ? vv[i].x * xa[i] : 0.0;
            const double p1 = (k + 1 < e1) ? vv[i].y * xb[i] : 0.0;
            *reinterpret_cast<double2 *>(prod + (k - e0)) = make_double2(p0, p1);
          }
        }
      }
    }
    __syncthreads();
    // pass 2
    const int rs = chunk_row[c], re = chunk_row[c + 1];          // rows that start in this chunk
    const bool has_in = (rs > 0) && (rs == (int)nrows || (int64_t)rowptr[rs] > e0) && e0 < e1;   // a row enters from the left
    const int m = (re - rs) + 1;                                  // slot 0 = entering row
    int lp = 1;
    while (lp < 64 && lp * 2 * m <= kThreads) lp *= 2;
    const int groups = kThreads / lp;
    const int g = tid % lp;
    for (int q = tid / lp; q < m; q += groups) {
      const bool entering = (q == 0);
      const int r = entering ? rs - 1 : rs + q - 1;
      double sum = 0.0;
      int64_t sb = 0, se = 0, t_r = 0;
      if (!entering || has_in) {
        t_r = NT ? __builtin_nontemporal_load(rowptr + r + 1) : rowptr[r + 1];
        sb = entering ? e0 : (int64_t)(NT ? __builtin_nontemporal_load(rowptr + r) : rowptr[r]);
        se = min(t_r, e1);
        if (se - sb > (int64_t)kLongPiece * lp) {          // uniform over the lp lanes of the row
          if (g == 0) long_q[atomicAdd(&long_n, 1)] = q;   // at most kSplitChunk / (kLongPiece + 1) such pieces in a chunk
          continue;
        }
        for (int64_t j = sb + g; j < se; j += lp) sum += prod[j - e0];
      }
      for (int off = lp >> 1; off > 0; off >>= 1) sum += __shfl_down(sum, off, lp);
      if (g == 0 && (!entering || has_in)) {
        if (entering) {
          carry[2 * c] = sum;                                     // folded by the fix-up kernel
        } else if (t_r <= e1) {                                   // starts and ends here: final
          if (MODE == 0) {
            const double yv = (beta == 0.0) ? alpha * sum : alpha * sum + beta * y[r];
            if (NT) __builtin_nontemporal_store(yv, y + r); else y[r] = yv;
          } else {
            if (NT) __builtin_nontemporal_store(sum, y + r); else y[r] = sum;
            const double xr = x[r];
            acc_yx += sum * xr;
            acc_xx += xr * xr;
          }
        } else {
          carry[2 * c + 1] = sum;                                 // leaves to the right
          if (MODE == 1) { const double xr = x[r]; acc_xx += xr * xr; }
        }
      }
    }
    __syncthreads();
    const int nl = long_n;                                        // the same in every thread
    if (nl > 0) {
      if (tid < nl) {                                             // list order = order of arrival: sort it (rank of each key), so that
        const int key = long_q[tid];                              // which wave sums which row (and adds to which partial of the dots) is fixed
        int rank = 0;
        for (int k = 0; k < nl; k++) rank += (long_q[k] < key);
        long_s[rank] = key;
      }
      __syncthreads();
      const int lane = tid & 63;
      for (int k = tid >> 6; k < nl; k += kThreads / 64) {
        const int q = long_s[k];
        const bool entering = (q == 0);
        const int r = entering ? rs - 1 : rs + q - 1;
        const int64_t t_r = rowptr[r + 1];
        const int64_t sb = entering ? e0 : (int64_t)rowptr[r];
        const int64_t se = min(t_r, e1);
        double sum = 0.0;
        for (int64_t j = sb + lane; j < se; j += 64) sum += prod[j - e0];
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
        if (lane == 0) {
          if (entering) {
            carry[2 * c] = sum;
          } else if (t_r <= e1) {
            if (MODE == 0) {
              const double yv = (beta == 0.0) ? alpha * sum : alpha * sum + beta * y[r];
              if (NT) __builtin_nontemporal_store(yv, y + r); else y[r] = yv;
            } else {
              if (NT) __builtin_nontemporal_store(sum, y + r); else y[r] = sum;
              const double xr = x[r];
              acc_yx += sum * xr;
              acc_xx += xr * xr;
            }
          } else {
            carry[2 * c + 1] = sum;
            if (MODE == 1) { const double xr = x[r]; acc_xx += xr * xr; }
          }
        }
      }
      __syncthreads();
    }
  }
  if (MODE == 1) {
    double v[2] = {acc_yx, acc_xx};
    block_sum<2, kThreads>(v, red);
    if (threadIdx.x == 0) {
      slab[0 * kSlabStride + blockIdx.x] = v[0];
      slab[1 * kSlabStride + blockIdx.x] = v[1];
    }
  }
}

// one thread per chunk whose last starting row runs past the chunk's end: add the pieces
template <int MODE>
__global__ __launch_bounds__(kThreads) void split_fixup_kernel(
    int64_t nrows, int64_t nnz, const int *__restrict__ rowptr, const int *__restrict__ chunk_row, int nchunks,
    const double *__restrict__ x, double *__restrict__ y, double alpha, double beta, const double *__restrict__ carry,
    double *__restrict__ slab, const int *__restrict__ done) {
  if (done != nullptr && *done != 0) return;
  __shared__ double red[2 * (kThreads / 64)];
  double acc_yx = 0.0;
  for (int c = blockIdx.x * kThreads + threadIdx.x; c < nchunks; c += gridDim.x * kThreads) {
    const int rs = chunk_row[c], re = chunk_row[c + 1];
    if (re <= rs) continue;                                       // no row starts here
    const int r = re - 1;
    const int64_t t_r = rowptr[r + 1];
    const int64_t e1 = min(((int64_t)c + 1) * kSplitChunk, nnz);
    if (t_r <= e1) continue;                                      // ends inside its own chunk
    double total = carry[2 * c + 1];
    for (int d = c + 1; d < nchunks; d++) {
      total += carry[2 * d];
      if (t_r <= ((int64_t)d + 1) * kSplitChunk) break;
    }
    if (MODE == 0) {
      y[r] = (beta == 0.0) ? alpha * total : alpha * total + beta * y[r];
    } else {
      y[r] = total;
      acc_yx += total * x[r];
    }
  }
  if (MODE == 1) {
    double v[2] = {acc_yx, 0.0};
    block_sum<2, kThreads>(v, red);
    if (threadIdx.x == 0) {
      slab[0 * kSlabStride + blockIdx.x] = v[0];
      slab[1 * kSlabStride + blockIdx.x] = v[1];
    }
  }
}

static __global__ void gather_x_kernel(int64_t n, const int *__restrict__ order, const double *__restrict__ x, double *__restrict__ xg,
                                       const int *__restrict__ done) {
  if (done != nullptr && *done != 0) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; j + 3 * stride < n; j += 4 * stride) {   // four index loads, then four gathers in flight per lane
    int o[4];
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = order[j + k * stride];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = x[o[k]];
#pragma unroll
    for (int k = 0; k < 4; k++) xg[j + k * stride] = v[k];
  }
  for (; j < n; j += stride) xg[j] = x[order[j]];
}

static int prepare_split(mk_matrix *A) {
  // x well beyond the 32 MB of L2 and enough entries to pay for one more pass over x per product.  Measured on R-MAT
  // with 16 entries per row: 2^22 columns 586 us plain / 618 us relabelled, 2^23 1392 / 1161, 2^24 3586 / 2387.
  static const int no_relabel = getenv("MK_NO_COLUMN_RELABEL") ? atoi(getenv("MK_NO_COLUMN_RELABEL")) : 0;
  static const int64_t min_cols = getenv("MK_RELABEL_MIN_COLS") ? atoll(getenv("MK_RELABEL_MIN_COLS")) : (int64_t)6 << 20;
  if (!A->relabel_tried && !no_relabel && A->ncols >= min_cols && A->nnz >= 4 * A->ncols) {
    A->relabel_tried = 1;
    (void)build_column_relabel(A);   // an optimisation only: on failure (memory) the product stays on the original numbering
  }
  A->relabel_tried = 1;
  if (A->chunk_row) return MK_OK;
  Context &c = ctx();
  A->nchunks = (int)((A->nnz + kSplitChunk - 1) / kSplitChunk);
  if (A->nchunks < 1) A->nchunks = 1;   // a matrix without entries still has rows to write
  MK_TRY(dev_alloc(&A->chunk_row, (size_t)A->nchunks + 1));
  MK_TRY(dev_alloc(&A->carry, (size_t)2 * A->nchunks));
  const int blocks = std::min((A->nchunks + 1 + 255) / 256, 1024);
  hipLaunchKernelGGL(split_chunk_rows_kernel, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->nnz, A->rowptr, A->nchunks, A->chunk_row);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// carry_override: the carry array to use instead of A->carry; plain_columns: original column numbering even when a
// relabelled copy exists (both for mk_spmv_dot_static, which may touch no buffer the library allocated as scratch)
template <int MODE>
static int launch_split(const mk_matrix *A, double alpha, const double *x, double beta, double *y, double *slab,
                        const int *done, int *nblocks, double *carry_override = nullptr, bool plain_columns = false) {
  Context &c = ctx();
  if (!carry_override) MK_TRY(prepare_split(const_cast<mk_matrix *>(A)));
  double *carry = carry_override ? carry_override : A->carry;
  const int main_blocks = std::max(1, std::min(A->nchunks, c.num_cus * 8));
  const int fix_blocks = std::max(1, std::min((A->nchunks + kThreads - 1) / kThreads, 1024));
  if (nblocks) *nblocks = main_blocks + fix_blocks;
  if (dry_launch()) return MK_OK;   // grid size only (mk_spmv_dot_static checks it against its scratch before anything is enqueued)
  // R-MAT 2^28 entries, x = 134 MB: 3.96 -> 3.54 ms with the matrix stream nontemporal (fewer x gathers miss)
  const int *colv = A->col;
  const double *xg = x;
  if (A->col2 && !plain_columns) {   // columns renumbered hot-first: xg[j] = x[order[j]], then the same product through col2
    hipLaunchKernelGGL(gather_x_kernel, dim3((unsigned)std::min<int64_t>((A->ncols + 255) / 256, 8192)), dim3(256), 0, c.stream,
                       A->ncols, A->xorder, x, A->xg, done);
    colv = A->col2;
    xg = A->xg;
  }
  if (nt_matrix((double)A->nnz * 12.0)) {
    hipLaunchKernelGGL((spmv_split_kernel<MODE, true>), dim3(main_blocks), dim3(kThreads), 0, c.stream, A->nrows, A->nnz, A->rowptr, colv,
                       A->val, A->chunk_row, A->nchunks, x, y, alpha, beta, carry, slab, done, xg);
  } else {
    hipLaunchKernelGGL((spmv_split_kernel<MODE, false>), dim3(main_blocks), dim3(kThreads), 0, c.stream, A->nrows, A->nnz, A->rowptr, colv,
                       A->val, A->chunk_row, A->nchunks, x, y, alpha, beta, carry, slab, done, xg);
  }
  hipLaunchKernelGGL((split_fixup_kernel<MODE>), dim3(fix_blocks), dim3(kThreads), 0, c.stream, A->nrows, A->nnz, A->rowptr,
                     A->chunk_row, A->nchunks, x, y, alpha, beta, carry, slab ? slab + main_blocks : nullptr, done);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

int launch_spmv(const mk_matrix *A, double alpha, const double *x, double beta, double *y, const int *skip) {
  MK_TRY(analyse_matrix(const_cast<mk_matrix *>(A)));
  if (A->nrows == 0) return MK_OK;
  if (A->kernel == MK_SPMV_VSWEEP) return launch_vsweep(A, 0, alpha, x, beta, y, nullptr, skip, nullptr, nullptr, nullptr);
  if (A->kernel == MK_SPMV_SWEEP) return launch_sweep<0>(A, alpha, x, beta, y, nullptr, skip, nullptr);
  if (A->kernel == MK_SPMV_PATTERN) return launch_pattern<0>(A, alpha, x, beta, y, nullptr, skip, nullptr);
  if (A->kernel == MK_SPMV_STREAM && A->grow) return launch_group(A, 0, alpha, x, beta, y, nullptr, skip, nullptr);
  if (A->kernel == MK_SPMV_STREAM && (A->band_desc || A->nstrips)) return launch_band(A, 0, alpha, x, beta, y, nullptr, skip, nullptr);
  if (A->kernel == MK_SPMV_STREAM) return launch_stream<0>(A, alpha, x, beta, y, nullptr, skip, nullptr);
  if (A->kernel == MK_SPMV_SPLIT) return launch_split<0>(A, alpha, x, beta, y, nullptr, skip, nullptr);
  return launch_vector<0>(A, alpha, x, beta, y, nullptr, skip, nullptr);
}

// one Jacobi sweep through the stream kernel; returns MK_ERR_UNSUPPORTED when the matrix does not
// use that kernel (the caller then runs its own lane-per-row sweep)
int launch_jacobi_sweep(const mk_matrix *A, const double *rhs, const double *extra, const double *xold, double *xnew,
                        double *slab, const int *done, int *nblocks, int extra_gap0, int extra_gap1, int plane_begin, int plane_end,
                        int max_blocks) {
  MK_TRY(analyse_matrix(const_cast<mk_matrix *>(A)));
  if (A->kernel == MK_SPMV_VSWEEP) return launch_vsweep(A, 2, 1.0, xold, 0.0, xnew, slab, done, nblocks, rhs, extra);
  if (A->kernel == MK_SPMV_SWEEP)   // rows [extra_gap0, extra_gap1) have extra == 0: the kernel reads one pair for all of them
    return launch_sweep<2>(A, 1.0, xold, 0.0, xnew, slab, done, nblocks, rhs, extra, nullptr, nullptr, (extra_gap0 + 1) & ~1, extra_gap1,
                           nullptr, 0, nullptr, plane_begin, plane_end, max_blocks);
  if (plane_end > plane_begin) return fail(MK_ERR_UNSUPPORTED, "jacobi sweep: plane ranges need the sweep kernel");
  if (A->kernel == MK_SPMV_PATTERN) return launch_pattern<2>(A, 1.0, xold, 0.0, xnew, slab, done, nblocks, rhs, extra);
  if (A->kernel != MK_SPMV_STREAM) return MK_ERR_UNSUPPORTED;
  return launch_stream<2>(A, 1.0, xold, 0.0, xnew, slab, done, nblocks, rhs, extra);
}

// blocks (= reduction partials) a Jacobi sweep over planes [plane_begin, plane_end) launches when at most max_blocks slots
// are left in the caller's slab; 0 when it does not fit.  No launch, no allocation: the row-block solver decides at
// set-up whether its interior + boundary launches fit one slab.
int jacobi_sweep_blocks(const mk_matrix *A, int plane_begin, int plane_end, int max_blocks) {
  if (A->kernel != MK_SPMV_SWEEP || !A->sweep || A->sweep_streamed) return 0;
  SweepArgs S = *A->sweep;
  if (plane_begin < 0 || plane_end > S.K || plane_end <= plane_begin) return 0;
  S.kb = plane_begin;
  S.K = plane_end - plane_begin;
  const int nseg = sweep_segments(A, 2, S, max_blocks);
  return nseg < 1 ? 0 : (S.ncp * nseg + 7) / 8 * 8;
}

int sweep_planes(const mk_matrix *A, int *rows_per_plane, int *nplanes) {
  if (A->kernel != MK_SPMV_SWEEP || !A->sweep || A->sweep_streamed) return 0;
  if (rows_per_plane) *rows_per_plane = A->sweep->P;
  if (nplanes) *nplanes = A->sweep->K;
  return 1;
}

int launch_sweep_cg_dots(const mk_matrix *A, const double *p, double *slab_out, PcgState *st, int *nblocks) {
  return launch_sweep<3>(A, 1.0, p, 0.0, nullptr, slab_out, nullptr, nblocks, nullptr, nullptr, st);
}

// single-reduction CG step (see MODE 6): sr_in = the five all-reduced values; extra (may be null) = off-rank part of A p'
// over the local rows, zero on the ghost-free run [gap0, gap1)
int launch_sweep_cg_single(const mk_matrix *A, const double *r_old, double *r_new, const double *p_old, double *p_new, double *x,
                           double zscale, const double *extra, int gap0, int gap1, const double *sr_in, double *slab_out,
                           PcgState *st, int two_norm, int rel_change, int first, int parity, double *norms, int *nblocks, int xmode, int pbuf) {
  if (r_old == r_new || p_old == p_new) return fail(MK_ERR_INVALID, "single-reduction CG step: r and p are read through other blocks' windows and cannot be updated in place");
  if (!sweep_single_ok(A)) return fail(MK_ERR_UNSUPPORTED, "single-reduction CG step: not with 2048-row chunks (register budget)");
  const int fl = (two_norm ? 1 : 0) | (rel_change ? 2 : 0) | (first ? 4 : 0) | ((parity & 1) << 3) | ((pbuf & 1) << 6);
  // xmode 1 / 2: the halved solution update (launch_sweep_cg_resid_x): the step leaves x alone / adds two steps' updates
  if (xmode == 1) return launch_sweep<9>(A, zscale, p_old, 0.0, p_new, slab_out, nullptr, nblocks, r_old, extra, st, sr_in, 1, 1, x, fl, norms, 0, 0,
                                         kSlabStride, r_new, (gap0 + 1) & ~1, gap1);
  if (xmode == 2) {
    if (first) return fail(MK_ERR_INVALID, "single-reduction CG step: the first step cannot carry a pending solution update");
    return launch_sweep<10>(A, zscale, p_old, 0.0, p_new, slab_out, nullptr, nblocks, r_old, extra, st, sr_in, 1, 1, x, fl, norms, 0, 0,
                            kSlabStride, r_new, (gap0 + 1) & ~1, gap1);
  }
  return launch_sweep<6>(A, zscale, p_old, 0.0, p_new, slab_out, nullptr, nblocks, r_old, extra, st, sr_in, 1, 1, x, fl, norms, 0, 0, kSlabStride,
                         r_new, (gap0 + 1) & ~1, gap1);
}

// the single-reduction step's kernel (MODE 6) needs ~157 VGPRs: chunks of 2048 rows (1024 threads, 128 VGPRs each) would spill
int sweep_single_ok(const mk_matrix *A) { return A->sweep && !A->sweep_streamed && A->sweep->C <= 1024; }

int sweep_uniform_diag(const mk_matrix *A, double *inv_diag) {
  if (!A->sweep || !A->sweep_diag_uniform || A->sweep_streamed) return 0;
  if (inv_diag) *inv_diag = 1.0 / A->sweep->dv;
  return 1;
}

int launch_sweep_cg_pdots(const mk_matrix *A, const double *r, const double *p_old, double *p_new, double zscale,
                          const double *slab_in, int nblocks_in, int stride_in, double *slab_out, PcgState *st, int two_norm,
                          int rel_change, double *norms, int *nblocks) {
  return launch_sweep<5>(A, zscale, p_old, 0.0, p_new, slab_out, nullptr, nblocks, r, nullptr, st, slab_in, nblocks_in, stride_in,
                         nullptr, (two_norm ? 1 : 0) | (rel_change ? 2 : 0), norms);
}

int launch_sweep_cg_resid_x(const mk_matrix *A, const double *p, double *r, double *x, const double *dinv, const double *slab_in,
                            int nblocks_in, int stride_in, double *slab_out, PcgState *st, int *nblocks, int xmode, int parity, int pbuf) {
  double zscale = 0.0;   // Jacobi preconditioner on a uniform diagonal: one number instead of an 8 B/row stream
  if (dinv && sweep_uniform_diag(A, &zscale)) dinv = nullptr; else zscale = 0.0;
  const int fl = ((parity & 1) << 3) | ((pbuf & 1) << 6);
  // xmode 1 / 2: the halved solution update (MODE 7 leaves x alone, MODE 8 adds two steps' worth); needs z = zscale r (no 1/diag stream)
  if (xmode == 1 && !dinv) return launch_sweep<7>(A, 0.0, p, zscale, r, slab_out, nullptr, nblocks, r, dinv, st, slab_in, nblocks_in, stride_in, x, fl);
  if (xmode == 2 && !dinv) return launch_sweep<8>(A, 0.0, p, zscale, r, slab_out, nullptr, nblocks, r, dinv, st, slab_in, nblocks_in, stride_in, x, fl);
  if (xmode != 0) return fail(MK_ERR_INVALID, "launch_sweep_cg_resid_x: the halved solution update needs a uniform diagonal");
  return launch_sweep<4>(A, 0.0, p, zscale, r, slab_out, nullptr, nblocks, r, dinv, st, slab_in, nblocks_in, stride_in, x);
}

int launch_spmv_dot(const mk_matrix *A, const double *x, double *y, double *slab, const int *done, int *nblocks) {
  MK_TRY(analyse_matrix(const_cast<mk_matrix *>(A)));
  if (A->kernel == MK_SPMV_VSWEEP) return launch_vsweep(A, 1, 1.0, x, 0.0, y, slab, done, nblocks, nullptr, nullptr);
  if (A->kernel == MK_SPMV_SWEEP) return launch_sweep<1>(A, 1.0, x, 0.0, y, slab, done, nblocks);
  if (A->kernel == MK_SPMV_PATTERN) return launch_pattern<1>(A, 1.0, x, 0.0, y, slab, done, nblocks);
  if (A->kernel == MK_SPMV_STREAM && A->grow) return launch_group(A, 1, 1.0, x, 0.0, y, slab, done, nblocks);
  if (A->kernel == MK_SPMV_STREAM && (A->band_desc || A->nstrips)) return launch_band(A, 1, 1.0, x, 0.0, y, slab, done, nblocks);
  if (A->kernel == MK_SPMV_STREAM) return launch_stream<1>(A, 1.0, x, 0.0, y, slab, done, nblocks);
  if (A->kernel == MK_SPMV_SPLIT) return launch_split<1>(A, 1.0, x, 0.0, y, slab, done, nblocks);
  return launch_vector<1>(A, 1.0, x, 0.0, y, slab, done, nblocks);
}

// ---------------------------------------------------------- the entry point that allocates nothing
// mk_spmv_dot_static: mk_spmv_dot for callers that cannot call hipMalloc (HIP graph capture, tight serving loops).
// Every byte of scratch its kernels need -- the slab of per-block partials, the two results, and the split kernel's
// carry pairs -- lives in this file-scope device array instead of in buffers the library allocated.  Each GPU holds
// its own copy (until the process ends); the host uses the address hipGetSymbolAddress gave for the current device,
// never the array's name.
constexpr int kStaticSlab = 2 * kSlabStride;        // partials of <y,x> and <x,x>: one slot per block, two components
constexpr int kStaticCarryChunks = 32768;           // split kernel: carry pairs of up to 32768 chunks = 2^26 entries
constexpr size_t kStaticDoubles = (size_t)kStaticSlab + 2 * (size_t)kStaticCarryChunks + 8;   // 592 KB per GPU
static __device__ __align__(256) double scratch[kStaticDoubles];
constexpr int kMaxDevices = 64;
static double *g_scratch_ptr[kMaxDevices] = {nullptr};

// called by mk_init with `device` current
int init_static_scratch(int device) {
  if (device < 0 || device >= kMaxDevices) return fail(MK_ERR_INVALID, "device %d: the static scratch table holds %d devices", device, kMaxDevices);
  if (g_scratch_ptr[device]) return MK_OK;
  double *p = nullptr;
  MK_HIP(hipGetSymbolAddress(reinterpret_cast<void **>(&p), HIP_SYMBOL(scratch)));
  if (!p) return fail(MK_ERR_HIP, "hipGetSymbolAddress returned no address for the static scratch");
  g_scratch_ptr[device] = p;
  return MK_OK;
}

__global__ void finish_dot_kernel(const double *slab, int nblocks, double *out2) {
  __shared__ double smem[2 * (kThreads / 64)];
  double v[2];
  reduce_slab<2, kThreads>(slab, nblocks, kSlabStride, v, smem);
  if (threadIdx.x == 0) { out2[0] = v[0]; out2[1] = v[1]; }
}

}  // namespace mk

using namespace mk;

extern "C" {

int mk_spmv(const mk_matrix *A, double alpha, const mk_vector *x, double beta, mk_vector *y) {
  MK_REQUIRE_INIT();
  if (!A || !x || !y) return fail(MK_ERR_INVALID, "mk_spmv: null argument");
  if (x->n < A->ncols || y->n < A->nrows)
    return fail(MK_ERR_INVALID, "mk_spmv: A is %lld x %lld but x has %lld and y has %lld entries",
                (long long)A->nrows, (long long)A->ncols, (long long)x->n, (long long)y->n);
  if (x->d == y->d) return fail(MK_ERR_INVALID, "mk_spmv: x and y must not alias");
  return launch_spmv(A, alpha, x->d, beta, y->d);
}

int mk_spmv_dot(const mk_matrix *A, const mk_vector *x, mk_vector *y, double *ytx, double *xtx) {
  MK_REQUIRE_INIT();
  if (!A || !x || !y) return fail(MK_ERR_INVALID, "mk_spmv_dot: null argument");
  if (A->nrows != A->ncols) return fail(MK_ERR_INVALID, "mk_spmv_dot: matrix must be square");
  if (x->n < A->ncols || y->n < A->nrows) return fail(MK_ERR_INVALID, "mk_spmv_dot: vector too short");
  if (x->d == y->d) return fail(MK_ERR_INVALID, "mk_spmv_dot: x and y must not alias");
  Context &c = ctx();
  int nb = 0;
  MK_TRY(launch_spmv_dot(A, x->d, y->d, c.slab, nullptr, &nb));
  if (!ytx && !xtx) return MK_OK;  // launch only (timing loops): the partials stay in the slab
  hipLaunchKernelGGL(finish_dot_kernel, dim3(1), dim3(kThreads), 0, c.stream, c.slab, nb, c.scalars);
  double h[2] = {0, 0};
  MK_HIP(hipMemcpyAsync(h, c.scalars, sizeof(h), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  if (ytx) *ytx = h[0];
  if (xtx) *xtx = h[1];
  return MK_OK;
}

int mk_spmv_dot_static(const mk_matrix *A, const mk_vector *x, mk_vector *y, double *ytx, double *xtx) {
  MK_REQUIRE_INIT();
  if (!A || !x || !y) return fail(MK_ERR_INVALID, "mk_spmv_dot_static: null argument");
  if (A->nrows != A->ncols) return fail(MK_ERR_INVALID, "mk_spmv_dot_static: matrix must be square");
  if (x->n < A->ncols || y->n < A->nrows) return fail(MK_ERR_INVALID, "mk_spmv_dot_static: vector too short");
  if (x->d == y->d) return fail(MK_ERR_INVALID, "mk_spmv_dot_static: x and y must not alias");
  Context &c = ctx();
  double *sp = (c.device >= 0 && c.device < kMaxDevices) ? g_scratch_ptr[c.device] : nullptr;
  if (!sp) return fail(MK_ERR_NOT_INIT, "mk_spmv_dot_static: no static scratch address for device %d (mk_init sets it)", c.device);
  // the analysis (row-length histogram, pattern tables, chunk starts) allocates: it must have run before
  if (!A->analysed) return fail(MK_ERR_INVALID, "mk_spmv_dot_static: matrix not analysed yet; call mk_matrix_get_info (it allocates) first");
  double *slab = sp, *out2 = sp + kStaticSlab, *carry = sp + kStaticSlab + 8;
  if (A->kernel == MK_SPMV_SPLIT) {
    if (!A->chunk_row) return fail(MK_ERR_INVALID, "mk_spmv_dot_static: the split kernel's chunk table is built (allocated) by the first mk_spmv / mk_spmv_dot; run one first");
    if (A->nchunks > kStaticCarryChunks)
      return fail(MK_ERR_UNSUPPORTED, "mk_spmv_dot_static: %lld entries in %d chunks exceed the static scratch (split-kernel matrices up to %lld entries)",
                  (long long)A->nnz, A->nchunks, (long long)kStaticCarryChunks * kSplitChunk);
  }
  // two passes over the same launcher: the first only works out the grid (nothing is enqueued), so that the number of
  // reduction partials is checked against the static slab BEFORE any kernel can write there
  int nb = 0;
  for (int pass = 0; pass < 2; pass++) {
    dry_launch() = (pass == 0);
    int rc;
    if (A->kernel == MK_SPMV_SPLIT) rc = launch_split<1>(A, 1.0, x->d, 0.0, y->d, slab, nullptr, &nb, carry, true);
    else rc = launch_spmv_dot(A, x->d, y->d, slab, nullptr, &nb);
    dry_launch() = 0;
    MK_TRY(rc);
    if (nb < 1 || nb > kSlabStride) return fail(MK_ERR_UNSUPPORTED, "mk_spmv_dot_static: %d reduction partials, the static scratch holds %d", nb, kSlabStride);
  }
  if (!ytx && !xtx) return MK_OK;
  hipLaunchKernelGGL(finish_dot_kernel, dim3(1), dim3(kThreads), 0, c.stream, slab, nb, out2);
  double *h = static_cast<double *>(c.pinned);   // pinned at mk_init: the copy needs no staging allocation
  MK_HIP(hipMemcpyAsync(h, out2, 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  if (ytx) *ytx = h[0];
  if (xtx) *xtx = h[1];
  return MK_OK;
}

int mk_matrix_get_info(const mk_matrix *A, mk_matrix_info *info) {
  MK_REQUIRE_INIT();
  if (!A || !info) return fail(MK_ERR_INVALID, "mk_matrix_get_info: null argument");
  MK_TRY(analyse_matrix(const_cast<mk_matrix *>(A)));
  *info = A->info;
  return MK_OK;
}

int mk_matrix_set_spmv_kernel(mk_matrix *A, int kernel, int lanes_per_row) {
  MK_REQUIRE_INIT();
  if (!A) return fail(MK_ERR_INVALID, "mk_matrix_set_spmv_kernel: null matrix");
  MK_TRY(analyse_matrix(A));
  if (kernel != MK_SPMV_AUTO && kernel != MK_SPMV_VECTOR && kernel != MK_SPMV_STREAM && kernel != MK_SPMV_SPLIT &&
      kernel != MK_SPMV_PATTERN && kernel != MK_SPMV_SWEEP && kernel != MK_SPMV_VSWEEP)
    return fail(MK_ERR_INVALID, "mk_matrix_set_spmv_kernel: unknown kernel %d", kernel);
  if (kernel == MK_SPMV_SWEEP && (!A->sweep || A->sweep_streamed))
    return fail(MK_ERR_UNSUPPORTED, "mk_matrix_set_spmv_kernel: the rows of this matrix are not sub-lists of one grid pattern with shared values");
  if (kernel == MK_SPMV_VSWEEP && (!A->sweep || !A->sweep_streamed))
    return fail(MK_ERR_UNSUPPORTED, "mk_matrix_set_spmv_kernel: MK_SPMV_VSWEEP serves matrices whose rows are sub-lists (by column offset) of one grid "
                                    "pattern of at most 28 entries and do not share their values (constant coefficients: MK_SPMV_SWEEP)");
  if (kernel == MK_SPMV_PATTERN && !A->vtab)
    return fail(MK_ERR_UNSUPPORTED, "mk_matrix_set_spmv_kernel: the rows of this matrix do not share (offsets, values) patterns");
  if (lanes_per_row != 0) {
    if (lanes_per_row < 1 || lanes_per_row > 64 || (lanes_per_row & (lanes_per_row - 1)))
      return fail(MK_ERR_INVALID, "mk_matrix_set_spmv_kernel: lanes_per_row must be a power of two in 1..64");
    A->lanes = lanes_per_row;
  }
  if (kernel == MK_SPMV_STREAM && A->stream_cap > kStreamMaxCap)
    return fail(MK_ERR_UNSUPPORTED, "mk_matrix_set_spmv_kernel: rows too long for the stream kernel (needs %d LDS entries, limit %d)",
                A->stream_cap, kStreamMaxCap);
  if (kernel != MK_SPMV_AUTO) A->kernel = kernel;
  A->info.kernel = A->kernel;
  A->info.lanes_per_row = A->lanes;
  return MK_OK;
}

}  // extern "C"
