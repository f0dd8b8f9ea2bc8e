// band.hip -- CSR SpMV with a sliding window of x kept in LDS (band kernel).
//
// Stands in for hypre_ParCSRMatrixMatvec on matrices whose rows reach a bounded distance from the diagonal but share
// neither offsets (no pattern table) nor column lists (no row groups): finite-element matrices in a bandwidth-reducing
// order.  The stream kernel gathers x per entry from global memory, and what bounds it there is the number of L1 -> L2
// requests of those gathers, not HBM (0.34-0.46 of the roofline on tools/fem_probe.py's `rand` cases).  Here
//   chunk  = up to 512 / G consecutive rows (G lanes per row from the mean row length), fewer where that many rows would
//            overflow the product stage;
//   block  = a contiguous run of chunks of about equal cost (entries + an average chunk's entries per chunk);
//   window = the last M columns up to the largest column any chunk so far has touched (H_k, a running maximum over the
//            whole matrix, so it only moves forward), kept in an LDS ring of M doubles (slot = column mod M).  A block
//            fills the ring once at its start; every chunk then adds only the columns its successor brings in
//            (H_k, H_k+1], loaded one chunk ahead.  M = the largest H_k - (smallest column of chunk k) + 1: the plan
//            exists when M doubles and the product stage fit half a CU's LDS; a matrix that needs more is cut into diagonal
//            strips (build_strips below), each with a plan of its own; last resort: one window, one block per CU;
//   pass 1 reads the chunk's (value, column) pairs with coalesced 16- and 8-byte loads requested ahead of their use
//            (two register sets; measured: how far ahead does not matter), multiplies every value with its x
//            from the ring and leaves the products in LDS;
//   pass 2 gives every row G lanes that add the row's products (lane g: entries g, g + G, ..., then a shuffle tree: the
//            stream kernel's order for the same G).
// No gather leaves the CU; x is read once per block.  HBM bytes: 12 per entry + 20 per row + 16 per chunk of descriptors.
// Results agree with the serial reference to 1e-12 (tests), and are the same from run to run (no atomics, fixed order).
// Algorithmic bytes: SURVEY 8(d)'s 12 nnz + 20 n + 4.
#include "mk_internal.h"

namespace mk {

constexpr int kBThreads = 512;
// A chunk's loads start at the last multiple of 32 entries in front of its first entry: 256 bytes of values, 128 of columns, so
// that every wave's load covers whole cache lines.  Measured (tools/microbench/read_stream.hip): two streams read side by side
// at 7.06 TB/s when they start on a line, at 5.41 TB/s when they start 40 / 20 bytes off one -- which is where a chunk of a CSR
// matrix starts.  The up to 31 entries in front are the previous chunk's (an L2 hit), multiplied with clamped ring indices into
// stage slots no row reads.
constexpr int kBandAlign = 32;
typedef double bd2_t __attribute__((ext_vector_type(2)));
typedef int bi2_t __attribute__((ext_vector_type(2)));
typedef int bi4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------- the plan
// out[k] = (smallest column, largest column) of chunk k (rows [crow[k], crow[k + 1])); (INT_MAX, -1) when it has no entry.
// One wave per chunk.
__global__ __launch_bounds__(256) void band_span_kernel(int nchunks, const int *__restrict__ crow, const int *__restrict__ rowptr,
                                                        const int *__restrict__ col, bi2_t *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < nchunks; k += gridDim.x * 4) {
    const int b = rowptr[crow[k]], e = rowptr[crow[k + 1]];
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
      bi2_t s;
      s.x = lo; s.y = hi;
      out[k] = s;
    }
  }
}

constexpr int kBandLdsTwo = 80 * 1024;          // LDS per block that lets two blocks of 512 threads share a CU
constexpr int kBandLdsOne = 160 * 1024 - 512;   // ... one block per CU

// The plan for one CSR matrix (A->info.mean_row_len set).  Lanes per row from the mean row length as for the stream kernel;
// a chunk is up to R = 512 / G consecutive rows, FEWER where R rows would overflow the product stage (greedy, on the host,
// from the row pointers): one stretch of fat rows costs a few more chunks there, not twice the lanes -- half the rows per
// chunk -- everywhere.  *need_out (optional): the window the matrix would need, whether or not the plan exists.
static int plan_band(mk_matrix *A, int64_t *need_out, int lds_budget) {
  if (need_out) *need_out = 0;
  if (A->nrows < 1 || A->nrows >= ((int64_t)1 << 31) - 65536) return MK_OK;
  Context &c = ctx();
  int sl = 1;
  while (sl < 32 && (double)sl * 16.0 < A->info.mean_row_len) sl *= 2;
  const int G = 2 * sl, R = kBThreads / G;
  std::vector<int> rp((size_t)A->nrows + 1);
  MK_HIP(hipMemcpyAsync(rp.data(), A->rowptr, sizeof(int) * rp.size(), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  // the stage: a quarter above the average chunk of R rows, at most the 4 x 1024 - 8 entries a thread's loads can fill
  const int cap_lim = (int)std::min<int64_t>(4080 - kBandAlign, std::max<int64_t>(1016, (int64_t)(1.25 * R * A->info.mean_row_len) + 64));
  std::vector<int> crow;
  crow.reserve((size_t)(A->nrows / R) * 5 / 4 + 16);
  int fullest = 64;
  for (int64_t r = 0; r < A->nrows;) {
    crow.push_back((int)r);
    int64_t z = r + 1;
    if (rp[z] - rp[r] > cap_lim) return MK_OK;   // a single row beyond the stage (the selector sends such matrices elsewhere)
    while (z < A->nrows && z - r < R && rp[z + 1] - rp[r] <= cap_lim) z++;
    fullest = std::max(fullest, rp[z] - rp[r]);
    r = z;
  }
  const int nchunks = (int)crow.size();
  crow.push_back((int)A->nrows);
  for (int q = 0; q < 8; q++) crow.push_back((int)A->nrows);   // (slack: descriptors are read a few chunks ahead)
  int *d_crow = nullptr;
  bi2_t *d_sp = nullptr;
  if (dev_alloc(&d_crow, crow.size()) != MK_OK) return MK_OK;   // an optimisation only
  if (dev_alloc(reinterpret_cast<int **>(&d_sp), (size_t)nchunks * 2) != MK_OK) { dev_free(d_crow); return MK_OK; }
  hipError_t e = hipMemcpyAsync(d_crow, crow.data(), sizeof(int) * crow.size(), hipMemcpyHostToDevice, c.stream);
  hipLaunchKernelGGL(band_span_kernel, dim3(std::min((nchunks + 3) / 4, 16384)), dim3(256), 0, c.stream, nchunks, d_crow, A->rowptr, A->col, d_sp);
  std::vector<int> h((size_t)nchunks * 2);
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_sp, sizeof(int) * 2 * (size_t)nchunks, hipMemcpyDeviceToHost, c.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  dev_free(reinterpret_cast<int *>(d_sp));
  dev_free(d_crow);
  if (e != hipSuccess) return fail(MK_ERR_HIP, "band plan: %s", hipGetErrorString(e));
  auto first_entry = [&](int k) { return rp[crow[k]]; };   // (k <= nchunks)
  // M = the widest reach back from the running maximum
  int64_t need = 1;
  {
    int H = -1;
    for (int k = 0; k < nchunks; k++) {
      const int lo = h[2 * k], hi = h[2 * k + 1];
      if (hi < lo) continue;
      if (lo < 0 || hi >= A->ncols) return MK_OK;   // (column indices are validated at assembly; never expected)
      H = std::max(H, hi);
      need = std::max<int64_t>(need, (int64_t)H - lo + 1);
    }
    if (H < 0) return MK_OK;
  }
  if (need_out) *need_out = need;
  const int cap = ((fullest + 3) & ~3) + kBandAlign;   // (the stage holds the entries in front of the chunk, back to the aligned start, as well)
  const int64_t M = (need + 7) & ~(int64_t)7;
  const int64_t lds = (M + cap + 4) * 8;
  // the window must fit beside the product stage (a window as long as x itself is not a window)
  if (lds > lds_budget || M > (int64_t)A->ncols + 8) return MK_OK;
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  int nb = (int)std::min<int64_t>(((int64_t)nchunks + 7) / 8 * 8, (int64_t)c.num_cus * per_cu);
  if (getenv("MK_BAND_BLOCKS")) nb = std::min(nb, std::max(8, atoi(getenv("MK_BAND_BLOCKS"))));
  nb = std::max(8, nb / 8 * 8);
  if (nb > kSlabStride) nb = kSlabStride;
  // every block reads M entries of x at its start: a block's share of the matrix must be at least four times that (fewer
  // blocks on small matrices)
  {
    const int64_t most = (int64_t)(12.0 * (double)A->nnz / (32.0 * (double)M));
    if (most < 8) return MK_OK;
    if (nb > most) nb = (int)(most / 8 * 8);
  }
  // chunk descriptors (first entry, H_k, first new column a_k = max(H_k-1 + 1, H_k + 1 - M), first row), five slack descriptors behind
  std::vector<int> desc(((size_t)nchunks + 5) * 4, 0);
  {
    int H = -1;
    for (int k = 0; k < nchunks; k++) {
      const int lo = h[2 * k], hi = h[2 * k + 1];
      const int Hprev = H;
      if (hi >= lo) H = std::max(H, hi);
      desc[4 * k] = first_entry(k);
      desc[4 * k + 1] = H;
      desc[4 * k + 2] = (int)std::max<int64_t>((int64_t)Hprev + 1, (int64_t)H + 1 - M);
      desc[4 * k + 3] = crow[k];
    }
    for (int k = nchunks; k < nchunks + 5; k++) { desc[4 * k] = (int)A->nnz; desc[4 * k + 1] = H; desc[4 * k + 2] = H + 1; desc[4 * k + 3] = (int)A->nrows; }
  }
  // Blocks take runs of chunks of equal COST, a chunk costing its entries + the entries of an average chunk: a chunk's step
  // has a fixed part (two barriers, the row pass, the latency of its loads) about as long as the streaming of an average
  // chunk's entries.  Balancing entries alone hands a block that owns a stretch of (nearly) empty rows ten times the chunks
  // of the others -- the first strip of a matrix cut into diagonal strips starts with such a stretch -- and the launch
  // waits for it (measured: 96 us per strip instead of 45).
  std::vector<int> bstart((size_t)nb + 1, nchunks);
  {
    const double avg = (double)A->nnz / (double)nchunks;
    const double total = 2.0 * (double)A->nnz;
    double acc = 0.0;
    int k = 0;
    for (int b = 0; b < nb; b++) {
      const double target = total * (double)b / (double)nb;
      while (k < nchunks && acc < target) {
        acc += (double)(first_entry(k + 1) - first_entry(k)) + avg;
        k++;
      }
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
  A->band_lanes = G;
  A->band_cap = cap;
  A->band_chunks = nchunks;
  A->band_blocks = nb;
  A->band_window = (int)M;
  return MK_OK;
}

// ---------------------------------------------------------------------------------- diagonal strips
// A matrix whose rows reach further than one window (M doubles would not fit the LDS) but whose entries lie within a bounded
// distance d = column - row of the diagonal is cut into S strips by d: A = A_0 + ... + A_S-1, A_s = the entries with
// dmin + s Wd <= d < dmin + (s + 1) Wd, each a CSR matrix of its own (rows in the original order, so every row's sum stays in
// a fixed order).  A strip's rows reach only Wd columns, so its window is Wd + the rows of a chunk, and the product is S
// band-kernel launches, y = A_0 x, y += A_1 x, ...: 12 B per entry as before, plus y and the strip's row pointers once per strip.
constexpr int kMaxStrips = 12;
struct StripPtrs {
  int *rowptr[kMaxStrips];
  int *col[kMaxStrips];
  double *val[kMaxStrips];
};

__global__ void offset_range_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col, int *__restrict__ range) {
  int lo = 0x7fffffff, hi = -0x7fffffff;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x)
    for (int j = rowptr[r]; j < rowptr[r + 1]; j++) {
      const int d = col[j] - (int)r;
      lo = min(lo, d);
      hi = max(hi, d);
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_down(lo, off, 64));
    hi = max(hi, __shfl_down(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    if (lo != 0x7fffffff) atomicMin(&range[0], lo);
    if (hi != -0x7fffffff) atomicMax(&range[1], hi);
  }
}

// counts[s * nrows + r] = entries of row r in strip s
__global__ void strip_count_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col, int dmin, int wd, int S,
                                   uint32_t *__restrict__ counts) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    uint32_t cnt[kMaxStrips];
#pragma unroll
    for (int q = 0; q < kMaxStrips; q++) cnt[q] = 0;
    for (int j = rowptr[r]; j < rowptr[r + 1]; j++) {
      const int q = min((col[j] - (int)r - dmin) / wd, S - 1);
#pragma unroll
      for (int t = 0; t < kMaxStrips; t++) cnt[t] += (t == q) ? 1u : 0u;
    }
    for (int q = 0; q < S; q++) counts[(size_t)q * nrows + r] = cnt[q];
  }
}

__global__ void strip_fill_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                                  int dmin, int wd, int S, StripPtrs P) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    int cur[kMaxStrips];
#pragma unroll
    for (int q = 0; q < kMaxStrips; q++) cur[q] = (q < S) ? P.rowptr[q][r] : 0;
    for (int j = rowptr[r]; j < rowptr[r + 1]; j++) {
      const int cj = col[j];
      const double vj = val[j];
      const int q = min((cj - (int)r - dmin) / wd, S - 1);
      int at = 0;
#pragma unroll
      for (int t = 0; t < kMaxStrips; t++)
        if (t == q) { at = cur[t]; cur[t]++; }
      P.col[q][at] = cj;
      P.val[q][at] = vj;
    }
  }
}

static void free_strips(mk_matrix *A) {
  for (int q = 0; q < A->nstrips; q++) { mk_matrix_destroy(A->strips[q]); A->strips[q] = nullptr; }
  A->nstrips = 0;
}

static int try_strips(mk_matrix *A, int S, int dmin, int64_t span, int *ok);

static int build_strips(mk_matrix *A, int64_t need) {
  const int off = getenv("MK_NO_BAND_STRIPS") ? atoi(getenv("MK_NO_BAND_STRIPS")) : 0;
  if (off || A->nstrips) return MK_OK;
  Context &c = ctx();
  int *range = nullptr, h_range[2] = {0x7fffffff, -0x7fffffff};
  if (dev_alloc(&range, 2) != MK_OK) return MK_OK;
  MK_HIP(hipMemcpyAsync(range, h_range, sizeof(h_range), hipMemcpyHostToDevice, c.stream));
  const int blocks = (int)std::min<int64_t>((A->nrows + 255) / 256, 4096);
  hipLaunchKernelGGL(offset_range_kernel, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->rowptr, A->col, range);
  MK_HIP(hipMemcpyAsync(h_range, range, sizeof(h_range), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  dev_free(range);
  if (h_range[1] < h_range[0]) return MK_OK;
  const int64_t span = (int64_t)h_range[1] - h_range[0] + 1;
  (void)need;
  // worth it while a strip keeps about six entries per row (below that y and the row pointers outweigh the entries) ...
  const double min_per_row = getenv("MK_BAND_STRIP_MIN_LEN") ? atof(getenv("MK_BAND_STRIP_MIN_LEN")) : 6.0;
  // ... and while a strip's launch keeps every block busy for a few chunks: on 400 000 rows of 80 entries the eight launches of
  // three chunks per block each (0.28 of the roofline) lose to the stream kernel's one (0.34)
  const int64_t strip_min_rows = getenv("MK_BAND_STRIP_MIN_ROWS") ? atoll(getenv("MK_BAND_STRIP_MIN_ROWS")) : 750000;
  if (A->nrows < strip_min_rows) return MK_OK;
  // The fewest strips whose windows fit: a strip's window is its width + the rows of a chunk, and two blocks per CU leave it
  // 80 KB less the product stage, which shrinks with the strip's rows (MK_BAND_STRIP_WIDTH fixes the width instead).  The
  // estimate is checked by planning: a strip that does not get its plan sends the loop on to one strip more.
  int S0 = 2;
  if (getenv("MK_BAND_STRIP_WIDTH")) {
    const int wd_env = std::max(64, atoi(getenv("MK_BAND_STRIP_WIDTH")));
    S0 = (int)std::max<int64_t>(2, (span + wd_env - 1) / wd_env);
  } else {
    for (; S0 <= kMaxStrips; S0++) {
      const double mean_s = A->info.mean_row_len / S0;
      int sl = 1;
      while (sl < 32 && (double)sl * 16.0 < mean_s) sl *= 2;
      const int Rs = kBThreads / (2 * sl);
      const int64_t cap_est = std::min<int64_t>(4080, (int64_t)(1.25 * Rs * mean_s) + 64 + kBandAlign + 32);
      const int64_t m_max = kBandLdsTwo / 8 - cap_est - 16;
      if ((span + S0 - 1) / S0 + Rs + 72 <= m_max) break;
    }
  }
  for (int S = S0; S <= kMaxStrips && S <= S0 + 2; S++) {
    if (A->info.mean_row_len < min_per_row * S) return MK_OK;
    int ok = 0;
    MK_TRY(try_strips(A, S, h_range[0], span, &ok));
    if (ok) return MK_OK;
  }
  return MK_OK;
}

// cut into S strips and plan each; *ok = 1 when every strip has its plan (else nothing is left behind)
static int try_strips(mk_matrix *A, int S, int dmin, int64_t span, int *ok) {
  *ok = 0;
  Context &c = ctx();
  const int blocks = (int)std::min<int64_t>((A->nrows + 255) / 256, 4096);
  const int h_range[1] = {dmin};
  const int wd = (int)(((span + S - 1) / S + 7) & ~(int64_t)7);
  uint32_t *counts = nullptr, *d_total = nullptr;
  if (dev_alloc(&counts, (size_t)S * A->nrows) != MK_OK) return MK_OK;
  if (dev_alloc(&d_total, 1) != MK_OK) { dev_free(counts); return MK_OK; }
  hipLaunchKernelGGL(strip_count_kernel, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->rowptr, A->col, h_range[0], wd, S, counts);
  StripPtrs P{};
  int rc = MK_OK;
  for (int q = 0; q < S && rc == MK_OK; q++) {
    mk_matrix *T = new mk_matrix();
    A->strips[A->nstrips++] = T;
    T->nrows = A->nrows; T->ncols = A->ncols;
    T->kernel = MK_SPMV_STREAM;
    T->analysed = true;   // (never analysed: served by the band kernel only)
    rc = dev_alloc(&T->rowptr, (size_t)A->nrows + 1);
    if (rc == MK_OK) rc = exclusive_scan_u32(counts + (size_t)q * A->nrows, reinterpret_cast<uint32_t *>(T->rowptr), (size_t)A->nrows, d_total);
    uint32_t tot = 0;
    if (rc == MK_OK) {
      MK_HIP(hipMemcpyAsync(&tot, d_total, sizeof(tot), hipMemcpyDeviceToHost, c.stream));
      MK_HIP(hipMemcpyAsync(T->rowptr + A->nrows, d_total, sizeof(tot), hipMemcpyDeviceToDevice, c.stream));
      MK_HIP(hipStreamSynchronize(c.stream));
      T->nnz = tot;
      rc = dev_alloc(&T->col, (size_t)tot + 4);
      if (rc == MK_OK) rc = dev_alloc(&T->val, (size_t)tot + 4);
    }
    P.rowptr[q] = T->rowptr; P.col[q] = T->col; P.val[q] = T->val;
  }
  dev_free(counts); dev_free(d_total);
  if (rc != MK_OK) { free_strips(A); return MK_OK; }
  hipLaunchKernelGGL(strip_fill_kernel, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->rowptr, A->col, A->val, h_range[0], wd, S, P);
  MK_HIP(hipStreamSynchronize(c.stream));
  MK_HIP(hipGetLastError());
  int wmax = 0;
  for (int q = 0; q < S; q++) {
    mk_matrix *T = A->strips[q];
    if (T->nnz == 0) continue;   // (an empty strip is skipped by the product)
    T->info.mean_row_len = (double)T->nnz / (double)T->nrows;
    T->info.nnz = A->nnz;   // (the whole matrix streams through the caches, not the strip alone: see launch_band_one)
    rc = plan_band(T, nullptr, kBandLdsTwo);
    if (rc != MK_OK || !T->band_desc) { free_strips(A); return MK_OK; }
    wmax = std::max(wmax, T->band_window);
    if (getenv("MK_BAND_TRACE"))
      fprintf(stderr, "band strip %d of %d: %lld entries, %d lanes per row, stage %d, window %d, %d blocks, %d chunks\n", q, S, (long long)T->nnz,
              T->band_lanes, T->band_cap, T->band_window, T->band_blocks, T->band_chunks);
  }
  A->band_window = wmax;
  *ok = 1;
  return MK_OK;
}

int build_band_plan(mk_matrix *A) {
  // (read per matrix, not once per process: the analysis runs once per matrix, and tests and probes switch per matrix)
  const int off = getenv("MK_NO_BAND") ? atoi(getenv("MK_NO_BAND")) : 0;
  // (short rows gain as well: 7 entries per row within +-1000 columns 0.53 -> 0.65 of the roofline, 12 within +-3000 0.45 -> 0.64)
  const double min_len = getenv("MK_BAND_MIN_LEN") ? atof(getenv("MK_BAND_MIN_LEN")) : 4.0;
  const int64_t min_rows = getenv("MK_BAND_MIN_ROWS") ? atoll(getenv("MK_BAND_MIN_ROWS")) : 100000;
  if (off || A->pat || A->grow || A->band_desc || A->nstrips || A->nrows < min_rows || A->nnz < 1 || A->info.mean_row_len < min_len) return MK_OK;
  // one window with two blocks per CU; else diagonal strips (measured on rand 2000000 45 7000: 251 us against 268 us for one
  // window of 110 KB with a single block per CU); else one window with one block per CU.  MK_BAND_LDS sets the first budget
  // (and then there is no third attempt).
  const int env_lds = getenv("MK_BAND_LDS") ? std::max(16384, std::min(atoi(getenv("MK_BAND_LDS")), kBandLdsOne)) : 0;
  int64_t need = 0;
  MK_TRY(plan_band(A, &need, env_lds ? env_lds : kBandLdsTwo));
  if (!A->band_desc && need > 0) MK_TRY(build_strips(A, need));
  if (!A->band_desc && !A->nstrips && need > 0 && !env_lds && (need + 1024) * 8 <= kBandLdsOne) MK_TRY(plan_band(A, nullptr, kBandLdsOne));   // (not worth the second pass over the row pointers when the window alone is too large)
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
    const int *__restrict__ done, int cap, int M, const bi4_t *__restrict__ cdesc, const int *__restrict__ bstart, int acc) {
  if (done != nullptr && *done != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *xw = reinterpret_cast<double *>(smem_raw);            // M doubles: the ring, slot = column mod M
  double *sprod = reinterpret_cast<double *>(smem_raw) + M;     // cap + 2 doubles: the chunk's products (M is a multiple of 8)
  const int zslot = cap + 2;                                    // ... then a zero no store reaches (read for lanes behind their row's end)
  if (threadIdx.x == 0) sprod[zslot] = 0.0;                     // (visible after the first barrier)
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
    const int vb_ = cb_ & ~(kBandAlign - 1);
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
    auto row_loads = [&](int chunk_, int first_row, int end_row) {
      const int64_t row_ = (int64_t)first_row + rl;
      nb_ = 0; ne_ = 0; nxr = 0.0; nyold = 0.0;
      if (chunk_ < k1 && row_ < end_row) {
        nb_ = rowptr[row_];
        ne_ = rowptr[row_ + 1];
        if (MODE == 1 && g == 0) nxr = x[row_];
        if (g == 0 && ((MODE == 0 && beta != 0.0) || (MODE == 1 && acc))) nyold = y[row_];
      }
    };
    row_loads(k0, d0.w, d1.w);
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
      const int cb = d0.x, ce = d1.x, vb = cb & ~(kBandAlign - 1);
      while (d0.y + 1 - M >= rb + M) rb += M;   // (block-uniform)
      const int64_t row = (int64_t)d0.w + rl;
      const bool have_row = row < d1.w;
      const int b = nb_ - vb, e = ne_ - vb;
      const double xr = nxr, yold = nyold;
      const bi4_t d4 = cdesc[j + 4];
      // pass 1: products of the staged pairs.  No per-entry predicates: the two entries a pair may hold from outside the chunk
      // (the odd one in front that a 16-byte load drags in, the one behind the end) are multiplied like the others, with their
      // ring index clamped into the ring, and land in stage slots no row reads (pass 2 reads [b, e) of its rows only).  What
      // bounds this kernel is its instruction count, so every select per entry counts.
      {
        const unsigned npairs_cur = (unsigned)(ce - vb + 1) >> 1;
        bd2_t *spairs = reinterpret_cast<bd2_t *>(sprod);
#pragma unroll
        for (int i = 0; i < U; i++) {
          const unsigned pi = (unsigned)tid + (unsigned)(i * kBThreads);
          if (pi < npairs_cur) {
            const unsigned s0 = min(slot_of(S.rc[i].x, rb), uM - 1u);
            const unsigned s1 = min(slot_of(S.rc[i].y, rb), uM - 1u);
            bd2_t pr;
            pr.x = S.rv[i].x * xw[s0];
            pr.y = S.rv[i].y * xw[s1];
            spairs[pi] = pr;
          }
        }
      }
      // (requested here, behind pass 1: the wait in front of pass 1 cannot be counted -- predicated loads -- so it waits for
      // every load in flight, and a load requested at the top of the step would put its round trip into every step)
      row_loads(j + 1, d1.w, d2.w);
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
      // (a lane behind its row's end reads the stage's zero slot: one 32-bit select instead of a 64-bit one and a clamp; slots
      // no lane of the wave needs are skipped by a wave-uniform branch)
      double sum = 0.0;
      {
        const int ee = have_row ? e : 0;
        const int q0 = (have_row ? b : 0) + g;   // (a lane without a row must not count as "in front of its end": its b is not a stage index)
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const int q = q0 + t * G;
          if (__any(q < ee)) sum += sprod[q < ee ? q : zslot];
        }
        for (int q = q0 + 8 * G; __any(q < ee); q += G) sum += sprod[q < ee ? q : zslot];
      }
      if (G > 1) sum = group_sum<G>(sum);
      if (g == 0 && have_row) {
        double yv;
        if (MODE == 0) {
          yv = (beta == 0.0) ? alpha * sum : alpha * sum + beta * yold;
        } else {
          yv = acc ? sum + yold : sum;   // (acc: the last strip of a matrix cut into diagonal strips)
          acc_yx += yv * xr;
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
static int launch_band_u(const mk_matrix *A, double alpha, const double *x, double beta, double *y, double *slab, const int *done, int *nblocks, int acc) {
  Context &c = ctx();
  const int G = A->band_lanes, cap = A->band_cap, M = A->band_window;
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
                       alpha, beta, slab, done, cap, M, reinterpret_cast<const bi4_t *>(A->band_desc), A->band_bstart, acc);   \
  } break;
  switch (G) {
    MK_LAUNCH_B(2) MK_LAUNCH_B(4) MK_LAUNCH_B(8) MK_LAUNCH_B(16) MK_LAUNCH_B(32) MK_LAUNCH_B(64)
    default: return fail(MK_ERR_INVALID, "spmv: band kernel lanes %d invalid", G);
  }
#undef MK_LAUNCH_B
  MK_HIP(hipGetLastError());
  return MK_OK;
}

static int launch_band_one(const mk_matrix *A, int mode, double alpha, const double *x, double beta, double *y, double *slab, const int *done, int *nblocks, int acc);

// mode 0: y = alpha A x + beta y; mode 1: y = A x + partials of <y,x>, <x,x>
int launch_band(const mk_matrix *A, int mode, double alpha, const double *x, double beta, double *y, double *slab, const int *done, int *nblocks) {
  if (A->nstrips == 0) return launch_band_one(A, mode, alpha, x, beta, y, slab, done, nblocks, 0);
  // diagonal strips: y = alpha A_0 x + beta y, then y += alpha A_s x; the dots ride in the last strip's launch
  int last = -1;
  for (int q = 0; q < A->nstrips; q++)
    if (A->strips[q]->nnz > 0) last = q;
  if (last < 0) return fail(MK_ERR_UNSUPPORTED, "spmv: band strips without entries");
  bool first = true;
  for (int q = 0; q <= last; q++) {
    const mk_matrix *T = A->strips[q];
    if (T->nnz == 0) continue;
    if (mode == 1 && q == last) MK_TRY(launch_band_one(T, 1, 1.0, x, 0.0, y, slab, done, nblocks, first ? 0 : 1));
    else if (mode == 1) MK_TRY(launch_band_one(T, 0, 1.0, x, first ? 0.0 : 1.0, y, nullptr, done, nullptr, 0));
    else MK_TRY(launch_band_one(T, 0, alpha, x, first ? beta : 1.0, y, nullptr, done, q == last ? nblocks : nullptr, 0));
    first = false;
  }
  return MK_OK;
}

static int launch_band_one(const mk_matrix *A, int mode, double alpha, const double *x, double beta, double *y, double *slab, const int *done, int *nblocks, int acc) {
  if (!A->band_desc || !A->band_bstart || A->band_lanes < 2 || A->band_rows != kBThreads / A->band_lanes || A->band_blocks < 8 || (A->band_blocks & 7))
    return fail(MK_ERR_UNSUPPORTED, "spmv: the band kernel needs a matrix whose rows stay within a bounded window of x");
  static const int env_nt = getenv("MK_BAND_NT") ? atoi(getenv("MK_BAND_NT")) : -1;   // (tuning: 0 / 1 forces plain / nontemporal matrix loads)
  const bool nt = env_nt >= 0 ? env_nt != 0 : nt_matrix((double)std::max<int64_t>(A->nnz, A->info.nnz) * 8.0);
  const int cap = A->band_cap;
#define MK_BAND_PICK(UU)                                                                                     \
  if (cap <= 1024 * UU - 8) {                                                                                \
    if (mode == 0) return nt ? launch_band_u<0, UU, true>(A, alpha, x, beta, y, slab, done, nblocks, acc)    \
                             : launch_band_u<0, UU, false>(A, alpha, x, beta, y, slab, done, nblocks, acc);  \
    return nt ? launch_band_u<1, UU, true>(A, alpha, x, beta, y, slab, done, nblocks, acc)                   \
              : launch_band_u<1, UU, false>(A, alpha, x, beta, y, slab, done, nblocks, acc);                 \
  }
  MK_BAND_PICK(1)
  MK_BAND_PICK(2)
  MK_BAND_PICK(3)
  MK_BAND_PICK(4)
#undef MK_BAND_PICK
  return fail(MK_ERR_UNSUPPORTED, "spmv: band chunk of %d entries exceeds the LDS stage", cap);
}

}  // namespace mk
