// convert.hip -- COO -> row-sorted COO -> CSR on the device, bit-exact against the
// reference's serial counting sort (sparse_matrix.c:471-513 sort_coo_row and :649-742
// convert_coo_to_csr).
//
// What "bit-exact" means here: the reference places entry i of row r at
// rowptr[r] + (number of earlier entries of row r), i.e. a STABLE sort by row; columns
// inside a row keep input order (the per-row column sort is commented out, :714-735).
// A per-row atomic cursor would not be stable, so the device path is a stable LSD radix
// sort on the row key carrying (col, val) along, 8 key bits per pass (1..4 passes for
// the bits nrows needs), followed by a boundary scan of the sorted keys that writes
// row_pointers.  Integer/byte work, HBM-bound: no MFMA.
//
// Per pass: digit histogram per tile (reads 4 B/entry) -> exclusive scan of the
// [digit][tile] table -> stable scatter (reads 16 B, writes 16 B per entry).
// Inputs that are already row-sorted (the usual .mtx case) skip the sort entirely.
#include <algorithm>
#include <chrono>
#include "mk_internal.h"

namespace mk {

// ------------------------------------------------------------------ exclusive scan
constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanChunk = kScanThreads * kScanItems;  // 4096

__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(const uint32_t *__restrict__ in, size_t n,
                                                                   uint32_t *__restrict__ sums) {
  __shared__ uint32_t sh[kScanThreads / 64];
  const size_t base = (size_t)blockIdx.x * kScanChunk;
  uint32_t s = 0;
  for (int k = 0; k < kScanItems; k++) {
    const size_t i = base + (size_t)k * kScanThreads + threadIdx.x;
    if (i < n) s += in[i];
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kScanThreads / 64; w++) t += sh[w];
    sums[blockIdx.x] = t;
  }
}

// one block scans up to kScanChunk*? entries serially in chunks (used for the top level)
__global__ __launch_bounds__(kScanThreads) void scan_single_block_kernel(const uint32_t *in, uint32_t *out, size_t n,
                                                                         uint32_t *total) {
  __shared__ uint32_t sh[kScanThreads];
  uint32_t carry = 0;
  for (size_t base = 0; base < n; base += kScanThreads) {
    const size_t i = base + threadIdx.x;
    const uint32_t v = (i < n) ? in[i] : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kScanThreads; off <<= 1) {
      uint32_t t = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0u;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < n) out[i] = carry + sh[threadIdx.x] - v;
    const uint32_t tot = sh[kScanThreads - 1];
    __syncthreads();
    carry += tot;
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                                  size_t n, const uint32_t *__restrict__ chunk_offsets) {
  // thread t owns kScanItems consecutive entries of its block's chunk
  __shared__ uint32_t sh[kScanThreads];
  const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    v[k] = (base + k < n) ? in[base + k] : 0u;
    s += v[k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < kScanThreads; off <<= 1) {
    uint32_t t = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0u;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = chunk_offsets[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
}

// block sums of every level of the scan of n values
static size_t scan_workspace_count(size_t n) {
  size_t total = 0;
  while (n > (size_t)kScanChunk) {
    n = (n + kScanChunk - 1) / kScanChunk;
    total += n;
  }
  return total;
}

// scan with the caller's workspace (scan_workspace_count(n) words): no allocation, nothing waits for the stream
static int exclusive_scan_u32_ws(const uint32_t *in, uint32_t *out, size_t n, uint32_t *total_out_dev, uint32_t *ws) {
  Context &c = ctx();
  if (n == 0) {
    if (total_out_dev) MK_HIP(hipMemsetAsync(total_out_dev, 0, sizeof(uint32_t), c.stream));
    return MK_OK;
  }
  if (n <= (size_t)kScanChunk) {
    hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(kScanThreads), 0, c.stream, in, out, n, total_out_dev);
    MK_HIP(hipGetLastError());
    return MK_OK;
  }
  const size_t nchunks = (n + kScanChunk - 1) / kScanChunk;
  uint32_t *sums = ws;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nchunks), dim3(kScanThreads), 0, c.stream, in, n, sums);
  MK_TRY(exclusive_scan_u32_ws(sums, sums, nchunks, total_out_dev, ws + nchunks));  // in place is fine for both scan forms
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nchunks), dim3(kScanThreads), 0, c.stream, in, out, n, sums);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

int exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *total_out_dev) {
  uint32_t *ws = nullptr;
  const size_t nws = scan_workspace_count(n);
  if (nws) MK_TRY(dev_alloc(&ws, nws));
  const int rc = exclusive_scan_u32_ws(in, out, n, total_out_dev, ws);
  if (ws) {   // still in use by queued kernels: free after they drain
    (void)hipStreamSynchronize(ctx().stream);
    dev_free(ws);
  }
  return rc;
}

// ------------------------------------------------------------------ input checks
struct CooCheck {
  int unsorted;      // some rows[i] < rows[i-1]
  int bad_row;       // row outside [0,nrows)
  int bad_col;       // col outside [0,ncols)
};

__global__ void coo_check_kernel(int64_t nnz, int64_t nrows, int64_t ncols, const int *__restrict__ rows,
                                 const int *__restrict__ cols, CooCheck *out) {
  int unsorted = 0, bad_row = 0, bad_col = 0;
  // four consecutive entries per thread (16-byte loads) plus the row in front of them
  const int64_t nq = nnz / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int4 r = *reinterpret_cast<const int4 *>(rows + 4 * q);
    const int4 cc = *reinterpret_cast<const int4 *>(cols + 4 * q);
    const int prev = q ? rows[4 * q - 1] : r.x;
    if ((r.x | r.y | r.z | r.w) < 0 || r.x >= nrows || r.y >= nrows || r.z >= nrows || r.w >= nrows) bad_row = 1;
    if ((cc.x | cc.y | cc.z | cc.w) < 0 || cc.x >= ncols || cc.y >= ncols || cc.z >= ncols || cc.w >= ncols) bad_col = 1;
    if (prev > r.x || r.x > r.y || r.y > r.z || r.z > r.w) unsorted = 1;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = 4 * nq; i < nnz; i++) {
      const int r = rows[i], cc = cols[i];
      if (r < 0 || r >= nrows) bad_row = 1;
      if (cc < 0 || cc >= ncols) bad_col = 1;
      if (i > 0 && rows[i - 1] > r) unsorted = 1;
    }
  if (__any(unsorted) && (threadIdx.x & 63) == 0) out->unsorted = 1;
  if (__any(bad_row) && (threadIdx.x & 63) == 0) out->bad_row = 1;
  if (__any(bad_col) && (threadIdx.x & 63) == 0) out->bad_col = 1;
}

// ------------------------------------------------------------------- radix passes
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kSortThreads = 512;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortItems = 8;
constexpr int kSortTile = kSortThreads * kSortItems;  // 8192 entries per block
constexpr int kWaveSpan = 64 * kSortItems;            // consecutive entries owned by one wave

// table[digit * ntiles + tile] = number of keys of that digit in the tile.  One block counts kHistTiles consecutive tiles,
// one after the other, and then writes kHistTiles neighbouring table entries per digit (the table is digit-major for the
// scan: a block per tile wrote 256 isolated 4-byte words).
constexpr int kHistTiles = 8;
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const int *__restrict__ keys, int64_t n, int shift,
                                                                  unsigned mask, uint32_t *__restrict__ table, int ntiles) {
  __shared__ unsigned int h[kHistTiles][kRadix];
  for (int i = threadIdx.x; i < kHistTiles * kRadix; i += kSortThreads) (&h[0][0])[i] = 0;
  __syncthreads();
  const int tile0 = blockIdx.x * kHistTiles;
  for (int t = 0; t < kHistTiles && tile0 + t < ntiles; t++) {
    const int64_t base = (int64_t)(tile0 + t) * kSortTile;
    if (base + kSortTile <= n) {   // whole tile: 16-byte loads (the count does not depend on the order inside the tile)
      typedef int i4v_t __attribute__((ext_vector_type(4)));
      const i4v_t *k4 = reinterpret_cast<const i4v_t *>(keys + base);
      constexpr int NV = kSortItems / 4;
      i4v_t v[NV];
#pragma unroll
      for (int k = 0; k < NV; k++) v[k] = __builtin_nontemporal_load(k4 + k * kSortThreads + threadIdx.x);
#pragma unroll
      for (int k = 0; k < NV; k++) {
        atomicAdd(&h[t][((unsigned)v[k].x >> shift) & mask], 1u);
        atomicAdd(&h[t][((unsigned)v[k].y >> shift) & mask], 1u);
        atomicAdd(&h[t][((unsigned)v[k].z >> shift) & mask], 1u);
        atomicAdd(&h[t][((unsigned)v[k].w >> shift) & mask], 1u);
      }
    } else {
#pragma unroll 4
      for (int k = 0; k < kSortItems; k++) {
        const int64_t i = base + (int64_t)k * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[t][((unsigned)keys[i] >> shift) & mask], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kHistTiles * kRadix; i += kSortThreads) {
    const int d = i / kHistTiles, t = i % kHistTiles;
    if (tile0 + t < ntiles) table[(size_t)d * ntiles + tile0 + t] = h[t][d];
  }
}

// Stable scatter.  Order inside a tile is (wave, item, lane): wave w owns entries
// [w*kWaveSpan, (w+1)*kWaveSpan) of the tile and walks them 64 at a time, so the rank of
// an entry among equal digits = (entries of that digit in earlier waves) + (in earlier
// items of this wave) + (in lower lanes of this item).  Entries are then regrouped by digit
// inside LDS, so that consecutive lanes store consecutive addresses of one digit's run
// (a direct scatter has every lane of a store instruction on a different line: 1 TB/s).
// The LDS staging buffer is used twice: (key, col) pairs first, then the values.
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(
    const int *__restrict__ keys_in, const int *__restrict__ cols_in, const double *__restrict__ vals_in,
    int *__restrict__ keys_out, int *__restrict__ cols_out, double *__restrict__ vals_out, int64_t n, int shift,
    unsigned mask, const uint32_t *__restrict__ table_scanned, int ntiles) {
  __shared__ unsigned int cnt[kSortWaves][kRadix];   // running count per wave and digit -> offset of the wave in the digit
  __shared__ unsigned int gbase[kRadix];             // global offset of (digit, this tile)
  __shared__ unsigned int lstart[kRadix];            // first tile-local sorted slot of the digit
  __shared__ __attribute__((aligned(16))) unsigned long long stage[kSortTile];   // 64 KB
  // Blocks are dealt round-robin over the 8 XCDs: block b takes tile (b % 8) * ceil(ntiles / 8) + b / 8, so that one XCD
  // works through a contiguous range of tiles in order.  The pieces that consecutive tiles write into one digit's output
  // range are adjacent: written through the same L2 they merge into whole lines before they leave it.
  const int per_xcd = (ntiles + 7) / 8;
  const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (tile >= ntiles) return;
  for (int i = threadIdx.x; i < kSortWaves * kRadix; i += kSortThreads) (&cnt[0][0])[i] = 0;
  for (int i = threadIdx.x; i < kRadix; i += kSortThreads) gbase[i] = table_scanned[(size_t)i * ntiles + tile];
  __syncthreads();

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t tbase = (int64_t)tile * kSortTile;
  const int64_t wbase = tbase + (int64_t)w * kWaveSpan;
  const int tile_n = (int)min((int64_t)kSortTile, n - tbase);
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  unsigned int slot[kSortItems];    // rank among equal digits, later the tile-local sorted slot
  unsigned long long kcv[kSortItems];   // col << 32 | key
  double valv[kSortItems];
  // everything the tile needs is requested up front: the column and value loads fly while the keys are ranked
  {
    int key[kSortItems];
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
      const int64_t i = min(wbase + (int64_t)k * 64 + lane, n - 1);
      key[k] = __builtin_nontemporal_load(keys_in + i);
    }
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
      const int64_t i = min(wbase + (int64_t)k * 64 + lane, n - 1);
      kcv[k] = ((unsigned long long)(unsigned)__builtin_nontemporal_load(cols_in + i) << 32) | (unsigned)key[k];
    }
  }
#pragma unroll
  for (int k = 0; k < kSortItems; k++) {
    const int64_t i = min(wbase + (int64_t)k * 64 + lane, n - 1);
    valv[k] = __builtin_nontemporal_load(vals_in + i);
  }
  volatile unsigned int *mycnt = cnt[w];
#pragma unroll
  for (int k = 0; k < kSortItems; k++) {
    const int64_t i = wbase + (int64_t)k * 64 + lane;
    const bool valid = i < n;
    const unsigned d = ((unsigned)kcv[k] >> shift) & mask;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < kRadixBits; b++) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const unsigned before = (unsigned)__popcll(peers & lt);
    const unsigned prev = valid ? mycnt[d] : 0u;
    slot[k] = prev + before;
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) mycnt[d] = prev + (unsigned)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // per digit: exclusive offsets over waves, and the digit's total -> exclusive scan over the 256 digits (threads 0..255 =
  // waves 0..3: shuffle scan inside each wave, the three wave totals through LDS)
  __shared__ unsigned int wtot[4];
  unsigned total = 0, incl = 0;
  if (threadIdx.x < kRadix) {
#pragma unroll
    for (int ww = 0; ww < kSortWaves; ww++) {
      const unsigned t = cnt[ww][threadIdx.x];
      cnt[ww][threadIdx.x] = total;
      total += t;
    }
    incl = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wtot[w] = incl;
  }
  __syncthreads();
  if (threadIdx.x < kRadix) {
    unsigned add = 0;
#pragma unroll
    for (int ww = 0; ww < 3; ww++) add += (ww < w) ? wtot[ww] : 0u;
    lstart[threadIdx.x] = add + incl - total;   // exclusive
  }
  __syncthreads();
  // round A: (key, col) pairs through LDS
#pragma unroll
  for (int k = 0; k < kSortItems; k++) {
    const int64_t i = wbase + (int64_t)k * 64 + lane;
    if (i < n) {
      const unsigned d = ((unsigned)kcv[k] >> shift) & mask;
      slot[k] = lstart[d] + cnt[w][d] + slot[k];
      stage[slot[k]] = kcv[k];
    }
  }
  __syncthreads();
  unsigned int gpos[kSortItems];
#pragma unroll
  for (int k = 0; k < kSortItems; k++) {
    const int j = threadIdx.x + k * kSortThreads;
    gpos[k] = 0;
    if (j < tile_n) {
      const unsigned long long kc = stage[j];
      const unsigned d = ((unsigned)kc >> shift) & mask;
      gpos[k] = gbase[d] + ((unsigned)j - lstart[d]);
      // plain stores: the pieces merge in L2 (nontemporal stores: 11.4 -> 14.1 ms for 2^28 entries).  Carrying the
      // (key, col) pair as one 8-byte word between the passes measured slower as well (scatter 2.46 -> 2.62 ms per pass,
      // and the histogram of the next pass reads twice the bytes)
      keys_out[gpos[k]] = (int)(unsigned)kc;
      cols_out[gpos[k]] = (int)(unsigned)(kc >> 32);
    }
  }
  __syncthreads();
  // round B: values through the same buffer
  double *vstage = reinterpret_cast<double *>(stage);
#pragma unroll
  for (int k = 0; k < kSortItems; k++) {
    const int64_t i = wbase + (int64_t)k * 64 + lane;
    if (i < n) vstage[slot[k]] = valv[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kSortItems; k++) {
    const int j = threadIdx.x + k * kSortThreads;
    if (j < tile_n) vals_out[gpos[k]] = vstage[j];
  }
}

// the histogram kernel above counts entries in (item, thread) order, the scatter kernel
// walks (wave, item, lane); both cover exactly the tile [blockIdx*kSortTile, +kSortTile).

// row_pointers from the sorted keys: entry i opens every row in (rows[i-1], rows[i]]
__global__ void rowptr_from_sorted_kernel(int64_t nnz, int64_t nrows, const int *__restrict__ rows, int *__restrict__ rowptr) {
  // four consecutive entries per thread (one 16-byte load) plus the entry in front of them
  const int64_t nq = (nnz + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = q * 4;
    int r[4];
    if (i0 + 4 <= nnz) {
      const int4 v = *reinterpret_cast<const int4 *>(rows + i0);
      r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
    } else {
      for (int j = 0; j < 4; j++) r[j] = (i0 + j < nnz) ? rows[i0 + j] : 0;
    }
    int prev = (i0 == 0) ? -1 : rows[i0 - 1];
    for (int j = 0; j < 4 && i0 + j < nnz; j++) {
      const int64_t i = i0 + j;
      for (int t = prev + 1; t <= r[j]; t++) rowptr[t] = (int)i;
      if (i == nnz - 1)
        for (int64_t t = (int64_t)r[j] + 1; t <= nrows; t++) rowptr[t] = (int)nnz;
      prev = r[j];
    }
  }
}

__global__ void fill_int_kernel(int *p, int64_t n, int v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

static int key_bits(int64_t nrows) {
  int bits = 1;
  while (bits < 31 && ((int64_t)1 << bits) < nrows) bits++;
  return bits;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define MK_TR(label) do { if (trace) { (void)hipStreamSynchronize(c.stream); double t_ = now_ms(); fprintf(stderr, "  [convert] %-28s %8.3f ms\n", label, t_ - t_prev); t_prev = t_; } } while (0)
int device_coo_to_csr(mk_coo *coo, mk_matrix *A, int **sorted_rows_out) {
  Context &c = ctx();
  static const int trace = getenv("MK_CONVERT_TRACE") ? atoi(getenv("MK_CONVERT_TRACE")) : 0;
  double t_prev = now_ms();
  const int64_t nnz = coo->nnz, nrows = coo->nrows;
  if (nnz > (int64_t)0x7fffffff) return fail(MK_ERR_UNSUPPORTED, "local nnz %lld exceeds int32 row pointers", (long long)nnz);
  A->nrows = nrows; A->ncols = coo->ncols; A->nnz = nnz;
  MK_TRY(dev_alloc(&A->rowptr, (size_t)nrows + 1));
  MK_TR("alloc rowptr");

  // 1. validate + detect the already-sorted case
  // flags in the context's reduction scratch (nothing else runs on the stream during a conversion), read back through
  // the pinned staging buffer: no allocation, no pageable copy
  CooCheck *d_chk = reinterpret_cast<CooCheck *>(c.slab);
  CooCheck &h = *static_cast<CooCheck *>(c.pinned);
  MK_HIP(hipMemsetAsync(d_chk, 0, sizeof(CooCheck), c.stream));
  if (nnz > 0) {
    int blocks = (int)std::min<int64_t>((nnz + 255) / 256, 4096);
    hipLaunchKernelGGL(coo_check_kernel, dim3(blocks), dim3(256), 0, c.stream, nnz, nrows, coo->ncols, coo->rows, coo->cols, d_chk);
  }
  MK_HIP(hipMemcpyAsync(&h, d_chk, sizeof(h), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  MK_TR("check");
  if (h.bad_row) return fail(MK_ERR_INVALID, "COO row index outside [0,%lld)", (long long)nrows);
  if (h.bad_col) return fail(MK_ERR_INVALID, "COO column index outside [0,%lld)", (long long)coo->ncols);

  int *keys = coo->rows, *cols = coo->cols;
  double *vals = coo->vals;
  char *scratch = nullptr;   // one allocation for everything that does not outlive the call (hipFree costs 0.2 ms a call)
  std::vector<void *> spent;   // the caller's arrays once the sorted entries live elsewhere
  if (h.unsorted) {
    // 2. stable LSD radix sort on the row key.  Passes alternate between a scratch set S and the final set F, arranged so
    // that the LAST pass writes F: the column and value arrays of F are allocated on their own (they become the matrix),
    // the keys of F, all of S, the digit table and the scan workspace share the scratch block.
    const int bits = key_bits(nrows);
    const int passes = (bits + kRadixBits - 1) / kRadixBits;
    const int ntiles = (int)((nnz + kSortTile - 1) / kSortTile);
    const size_t ntab = (size_t)kRadix * ntiles;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 511) & ~(size_t)255; return o; };   // >= 256 B slack each
    const size_t o_keys_f = carve((size_t)nnz * 4);
    // S is first written by pass 1 when the number of passes is odd: the caller's own arrays (pass 0 has read them by then)
    // serve, and the scratch block shrinks from 20 to 4 bytes per entry
    const bool s_is_input = (passes % 2 == 1);
    const bool s_in_scratch = passes > 1 && !s_is_input;
    const size_t o_keys_s = s_in_scratch ? carve((size_t)nnz * 4) : 0, o_cols_s = s_in_scratch ? carve((size_t)nnz * 4) : 0;
    const size_t o_vals_s = s_in_scratch ? carve((size_t)nnz * 8) : 0;
    const size_t o_table = carve(ntab * 4), o_ws = carve(scan_workspace_count(ntab) * 4);
    MK_TRY(dev_alloc(&scratch, off));
    int *cols_f = nullptr;
    double *vals_f = nullptr;
    int rc = dev_alloc(&cols_f, (size_t)nnz);
    if (rc == MK_OK) rc = dev_alloc(&vals_f, (size_t)nnz);
    if (rc != MK_OK) { dev_free(scratch); dev_free(cols_f); return rc; }
    int *keys_f = reinterpret_cast<int *>(scratch + o_keys_f);
    int *keys_s = s_is_input ? keys : reinterpret_cast<int *>(scratch + o_keys_s);
    int *cols_s = s_is_input ? cols : reinterpret_cast<int *>(scratch + o_cols_s);
    double *vals_s = s_is_input ? vals : reinterpret_cast<double *>(scratch + o_vals_s);
    uint32_t *table = reinterpret_cast<uint32_t *>(scratch + o_table), *ws = reinterpret_cast<uint32_t *>(scratch + o_ws);
    MK_TR("alloc sort buffers");
    spent.push_back(keys);
    spent.push_back(cols);
    spent.push_back(vals);
    int shift = 0;
    for (int p = 0; p < passes; p++) {
      // spread the key bits evenly over the passes
      const int pb = (bits - shift + (passes - p) - 1) / (passes - p);
      const unsigned mask = (1u << pb) - 1u;
      const bool to_f = ((passes - 1 - p) % 2) == 0;
      int *keys_o = to_f ? keys_f : keys_s, *cols_o = to_f ? cols_f : cols_s;
      double *vals_o = to_f ? vals_f : vals_s;
      hipLaunchKernelGGL(radix_hist_kernel, dim3((ntiles + kHistTiles - 1) / kHistTiles), dim3(kSortThreads), 0, c.stream, keys, nnz, shift, mask, table, ntiles);
      MK_TR("hist");
      rc = exclusive_scan_u32_ws(table, table, ntab, nullptr, ws);
      if (rc != MK_OK) break;
      MK_TR("scan");
      hipLaunchKernelGGL(radix_scatter_kernel, dim3((ntiles + 7) / 8 * 8), dim3(kSortThreads), 0, c.stream, keys, cols, vals, keys_o, cols_o,
                         vals_o, nnz, shift, mask, table, ntiles);
      keys = keys_o;
      cols = cols_o;
      vals = vals_o;
      shift += pb;
      MK_TR("scatter");
    }
    if (rc == MK_OK && hipGetLastError() != hipSuccess) rc = fail(MK_ERR_HIP, "COO -> CSR: a sort kernel failed to launch");
    if (rc != MK_OK) {
      (void)hipStreamSynchronize(c.stream);
      dev_free(scratch); dev_free(cols_f); dev_free(vals_f);
      return rc;
    }
  }
  // 3. row pointers from the sorted keys
  if (nnz == 0) {
    int blocks = (int)std::min<int64_t>((nrows + 1 + 255) / 256, 4096);
    hipLaunchKernelGGL(fill_int_kernel, dim3(blocks), dim3(256), 0, c.stream, A->rowptr, nrows + 1, 0);
  } else {
    int blocks = (int)std::min<int64_t>(((nnz + 3) / 4 + 255) / 256, 8192);
    hipLaunchKernelGGL(rowptr_from_sorted_kernel, dim3(blocks), dim3(256), 0, c.stream, nnz, nrows, keys, A->rowptr);
  }
  MK_HIP(hipGetLastError());
  A->col = cols;
  A->val = vals;
  if (scratch) {   // sorted: keys live in the scratch block
    if (sorted_rows_out) {
      int *copy = nullptr;
      MK_TRY(dev_alloc(&copy, (size_t)nnz));
      MK_HIP(hipMemcpyAsync(copy, keys, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToDevice, c.stream));
      *sorted_rows_out = copy;
    }
    MK_HIP(hipStreamSynchronize(c.stream));
    dev_free(scratch);
    for (void *q : spent) dev_free(q);
    MK_TR("rowptr + free");
  } else if (sorted_rows_out) {
    *sorted_rows_out = keys;
  } else {
    MK_HIP(hipStreamSynchronize(c.stream));
    dev_free(keys);
    MK_TR("free keys");
  }
  coo->rows = nullptr; coo->cols = nullptr; coo->vals = nullptr; coo->nnz = 0;
  return MK_OK;
}

static int upload_coo(const SparseMatrixCOO *h, mk_coo *d) {
  Context &c = ctx();
  d->nrows = h->nrows; d->ncols = h->ncolumns; d->nnz = h->nnz;
  MK_TRY(dev_alloc(&d->rows, (size_t)d->nnz));
  MK_TRY(dev_alloc(&d->cols, (size_t)d->nnz));
  MK_TRY(dev_alloc(&d->vals, (size_t)d->nnz));
  if (d->nnz) {
    if (!h->rows || !h->columns || !h->values) return fail(MK_ERR_INVALID, "COO arrays are null");
    MK_HIP(hipMemcpyAsync(d->rows, h->rows, sizeof(int) * d->nnz, hipMemcpyHostToDevice, c.stream));
    MK_HIP(hipMemcpyAsync(d->cols, h->columns, sizeof(int) * d->nnz, hipMemcpyHostToDevice, c.stream));
    MK_HIP(hipMemcpyAsync(d->vals, h->values, sizeof(double) * d->nnz, hipMemcpyHostToDevice, c.stream));
    MK_HIP(hipStreamSynchronize(c.stream));
  }
  return MK_OK;
}

static void free_coo_buffers(mk_coo *d) {
  dev_free(d->rows); dev_free(d->cols); dev_free(d->vals);
  d->rows = nullptr; d->cols = nullptr; d->vals = nullptr;
}

static void free_matrix_buffers(mk_matrix *A) {
  dev_free(A->rowptr); dev_free(A->col); dev_free(A->val); dev_free(A->dinv);
  dev_free(A->chunk_row); dev_free(A->long_rows); dev_free(A->carry); dev_free(A->pat); dev_free(A->ptab); dev_free(A->stab); dev_free(A->colc); dev_free(A->cptr); dev_free(A->rowpair); dev_free(A->grow); dev_free(A->gcol); dev_free(A->gdesc); dev_free(A->cdesc); dev_free(A->band_desc); dev_free(A->band_bstart);
  for (int q = 0; q < A->nstrips; q++) { if (A->strips[q]) { free_matrix_buffers(A->strips[q]); delete A->strips[q]; A->strips[q] = nullptr; } }
  A->nstrips = 0;
  dev_free(A->vtab); dev_free(A->otab8); dev_free(A->vtab8); dev_free(A->pmask); dev_free(A->etab); dev_free(A->sweep_perm); delete A->sweep;
  dev_free(A->col2); dev_free(A->xg); dev_free(A->xorder);
}

template <typename T>
static T *host_alloc(size_t count) {
  size_t bytes = count * sizeof(T);
  if (bytes == 0) bytes = 64;
  bytes = (bytes + 63) / 64 * 64;  // aligned_alloc wants a multiple of the alignment
  return static_cast<T *>(aligned_alloc(64, bytes));
}

}  // namespace mk

using namespace mk;

extern "C" {

int mk_coo_from_host(const SparseMatrixCOO *coo, mk_coo **out) {
  MK_REQUIRE_INIT();
  if (!coo || !out) return fail(MK_ERR_INVALID, "mk_coo_from_host: null argument");
  mk_coo *d = new mk_coo();
  int rc = upload_coo(coo, d);
  if (rc != MK_OK) { free_coo_buffers(d); delete d; return rc; }
  *out = d;
  return MK_OK;
}

int mk_coo_download(const mk_coo *coo, SparseMatrixCOO *out) {
  MK_REQUIRE_INIT();
  if (!coo || !out) return fail(MK_ERR_INVALID, "mk_coo_download: null argument");
  Context &c = ctx();
  out->nrows = (unsigned)coo->nrows; out->ncolumns = (unsigned)coo->ncols; out->nnz = (unsigned)coo->nnz;
  out->rows = host_alloc<int>((size_t)coo->nnz);
  out->columns = host_alloc<int>((size_t)coo->nnz);
  out->values = host_alloc<double>((size_t)coo->nnz);
  if (!out->rows || !out->columns || !out->values) return fail(MK_ERR_ALLOC, "mk_coo_download: host allocation failed");
  if (coo->nnz) {
    MK_HIP(hipMemcpyAsync(out->rows, coo->rows, sizeof(int) * coo->nnz, hipMemcpyDeviceToHost, c.stream));
    MK_HIP(hipMemcpyAsync(out->columns, coo->cols, sizeof(int) * coo->nnz, hipMemcpyDeviceToHost, c.stream));
    MK_HIP(hipMemcpyAsync(out->values, coo->vals, sizeof(double) * coo->nnz, hipMemcpyDeviceToHost, c.stream));
    MK_HIP(hipStreamSynchronize(c.stream));
  }
  return MK_OK;
}

int mk_coo_destroy(mk_coo *coo) {
  if (!coo) return MK_OK;
  if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
  free_coo_buffers(coo);
  delete coo;
  return MK_OK;
}

int mk_matrix_from_device_coo(mk_coo *coo, mk_matrix **A) {
  MK_REQUIRE_INIT();
  if (!coo || !A) return fail(MK_ERR_INVALID, "mk_matrix_from_device_coo: null argument");
  mk_matrix *M = new mk_matrix();
  int rc = device_coo_to_csr(coo, M, nullptr);
  if (rc != MK_OK) { free_matrix_buffers(M); delete M; return rc; }
  *A = M;
  return MK_OK;
}

int mk_matrix_from_coo(const SparseMatrixCOO *coo, mk_matrix **A) {
  MK_REQUIRE_INIT();
  if (!coo || !A) return fail(MK_ERR_INVALID, "mk_matrix_from_coo: null argument");
  mk_coo d;
  int rc = upload_coo(coo, &d);
  if (rc != MK_OK) { free_coo_buffers(&d); return rc; }
  mk_matrix *M = new mk_matrix();
  rc = device_coo_to_csr(&d, M, nullptr);
  free_coo_buffers(&d);
  if (rc != MK_OK) { free_matrix_buffers(M); delete M; return rc; }
  *A = M;
  return MK_OK;
}

int mk_matrix_from_csr(const SparseMatrixCSR *csr, mk_matrix **A) {
  MK_REQUIRE_INIT();
  if (!csr || !A) return fail(MK_ERR_INVALID, "mk_matrix_from_csr: null argument");
  if (!csr->row_pointers || (csr->nnz && (!csr->column_indices || !csr->values)))
    return fail(MK_ERR_INVALID, "mk_matrix_from_csr: null array");
  if (csr->nnz > 0x7fffffffu) return fail(MK_ERR_UNSUPPORTED, "mk_matrix_from_csr: nnz exceeds int32");
  // host-side shape check before anything reaches a kernel
  if (csr->row_pointers[0] != 0 || (unsigned)csr->row_pointers[csr->nrows] != csr->nnz)
    return fail(MK_ERR_INVALID, "mk_matrix_from_csr: row_pointers[0]=%d, row_pointers[nrows]=%d, nnz=%u",
                csr->row_pointers[0], csr->row_pointers[csr->nrows], csr->nnz);
  for (unsigned r = 0; r < csr->nrows; r++)
    if (csr->row_pointers[r + 1] < csr->row_pointers[r])
      return fail(MK_ERR_INVALID, "mk_matrix_from_csr: row_pointers decrease at row %u", r);
  for (unsigned k = 0; k < csr->nnz; k++)
    if (csr->column_indices[k] < 0 || (unsigned)csr->column_indices[k] >= csr->ncolumns)
      return fail(MK_ERR_INVALID, "mk_matrix_from_csr: column index %d at entry %u outside [0,%u)",
                  csr->column_indices[k], k, csr->ncolumns);
  Context &c = ctx();
  mk_matrix *M = new mk_matrix();
  M->nrows = csr->nrows; M->ncols = csr->ncolumns; M->nnz = csr->nnz;
  int rc = dev_alloc(&M->rowptr, (size_t)M->nrows + 1);
  // (a few entries of slack: the staging kernels read whole 8- / 16-byte pairs, so the load that holds the last entry may reach
  // one entry behind it)
  if (rc == MK_OK) rc = dev_alloc(&M->col, (size_t)M->nnz + 16);
  if (rc == MK_OK) rc = dev_alloc(&M->val, (size_t)M->nnz + 16);
  if (rc != MK_OK) { free_matrix_buffers(M); delete M; return rc; }
  hipError_t e = hipMemcpyAsync(M->rowptr, csr->row_pointers, sizeof(int) * (M->nrows + 1), hipMemcpyHostToDevice, c.stream);
  if (e == hipSuccess && M->nnz) e = hipMemcpyAsync(M->col, csr->column_indices, sizeof(int) * M->nnz, hipMemcpyHostToDevice, c.stream);
  if (e == hipSuccess && M->nnz) e = hipMemcpyAsync(M->val, csr->values, sizeof(double) * M->nnz, hipMemcpyHostToDevice, c.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  if (e != hipSuccess) { free_matrix_buffers(M); delete M; return fail(MK_ERR_HIP, "mk_matrix_from_csr: %s", hipGetErrorString(e)); }
  *A = M;
  return MK_OK;
}

int mk_matrix_destroy(mk_matrix *A) {
  if (!A) return MK_OK;
  if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
  free_matrix_buffers(A);
  delete A;
  return MK_OK;
}

int mk_matrix_get_csr(const mk_matrix *A, SparseMatrixCSR *csr) {
  MK_REQUIRE_INIT();
  if (!A || !csr) return fail(MK_ERR_INVALID, "mk_matrix_get_csr: null argument");
  Context &c = ctx();
  csr->nrows = (unsigned)A->nrows; csr->ncolumns = (unsigned)A->ncols; csr->nnz = (unsigned)A->nnz;
  csr->row_pointers = host_alloc<int>((size_t)A->nrows + 1);
  csr->column_indices = host_alloc<int>((size_t)A->nnz);
  csr->values = host_alloc<double>((size_t)A->nnz);
  if (!csr->row_pointers || !csr->column_indices || !csr->values) return fail(MK_ERR_ALLOC, "mk_matrix_get_csr: host allocation failed");
  MK_HIP(hipMemcpyAsync(csr->row_pointers, A->rowptr, sizeof(int) * (A->nrows + 1), hipMemcpyDeviceToHost, c.stream));
  if (A->nnz) {
    MK_HIP(hipMemcpyAsync(csr->column_indices, A->col, sizeof(int) * A->nnz, hipMemcpyDeviceToHost, c.stream));
    MK_HIP(hipMemcpyAsync(csr->values, A->val, sizeof(double) * A->nnz, hipMemcpyDeviceToHost, c.stream));
  }
  MK_HIP(hipStreamSynchronize(c.stream));
  return MK_OK;
}

int mk_coo_to_csr(const SparseMatrixCOO *coo, SparseMatrixCSR *csr) {
  MK_REQUIRE_INIT();
  if (!coo || !csr) return fail(MK_ERR_INVALID, "mk_coo_to_csr: null argument");
  mk_matrix *M = nullptr;
  MK_TRY(mk_matrix_from_coo(coo, &M));
  int rc = mk_matrix_get_csr(M, csr);
  mk_matrix_destroy(M);
  return rc;
}

int mk_sort_coo_row(const SparseMatrixCOO *coo, SparseMatrixCOO *sorted) {
  MK_REQUIRE_INIT();
  if (!coo || !sorted) return fail(MK_ERR_INVALID, "mk_sort_coo_row: null argument");
  Context &c = ctx();
  mk_coo d;
  int rc = upload_coo(coo, &d);
  if (rc != MK_OK) { free_coo_buffers(&d); return rc; }
  mk_matrix M;
  int *keys = nullptr;
  rc = device_coo_to_csr(&d, &M, &keys);
  free_coo_buffers(&d);
  if (rc != MK_OK) { free_matrix_buffers(&M); return rc; }
  sorted->nrows = coo->nrows; sorted->ncolumns = coo->ncolumns; sorted->nnz = coo->nnz;
  sorted->rows = host_alloc<int>(coo->nnz);
  sorted->columns = host_alloc<int>(coo->nnz);
  sorted->values = host_alloc<double>(coo->nnz);
  hipError_t e = hipSuccess;
  if (!sorted->rows || !sorted->columns || !sorted->values) rc = fail(MK_ERR_ALLOC, "mk_sort_coo_row: host allocation failed");
  if (rc == MK_OK && coo->nnz) {
    e = hipMemcpyAsync(sorted->rows, keys, sizeof(int) * coo->nnz, hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sorted->columns, M.col, sizeof(int) * coo->nnz, hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sorted->values, M.val, sizeof(double) * coo->nnz, hipMemcpyDeviceToHost, c.stream);
  }
  hipError_t e2 = hipStreamSynchronize(c.stream);
  if (e == hipSuccess) e = e2;
  if (rc == MK_OK && e != hipSuccess) rc = fail(MK_ERR_HIP, "mk_sort_coo_row: %s", hipGetErrorString(e));
  dev_free(keys);
  free_matrix_buffers(&M);
  return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------ transpose
namespace mk {

static __global__ void expand_rows_kernel(int64_t nrows, const int *__restrict__ rowptr, int *__restrict__ rows) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x)
    for (int k = rowptr[r]; k < rowptr[r + 1]; k++) rows[k] = (int)r;
}

// A^T as a matrix of its own (CGNR needs A^T r every iteration): entries as (col, row, val) through the same
// stable radix sort as any other COO, so rows of A^T hold their entries by ascending row of A
int build_transpose(const mk_matrix *A, mk_matrix **out) {
  Context &c = ctx();
  mk_coo d;
  d.nrows = A->ncols; d.ncols = A->nrows; d.nnz = A->nnz;
  int rc = dev_alloc(&d.rows, (size_t)A->nnz);
  if (rc == MK_OK) rc = dev_alloc(&d.cols, (size_t)A->nnz);
  if (rc == MK_OK) rc = dev_alloc(&d.vals, (size_t)A->nnz);
  if (rc != MK_OK) { dev_free(d.rows); dev_free(d.cols); dev_free(d.vals); return rc; }
  if (A->nnz > 0) {
    (void)hipMemcpyAsync(d.rows, A->col, sizeof(int) * (size_t)A->nnz, hipMemcpyDeviceToDevice, c.stream);
    (void)hipMemcpyAsync(d.vals, A->val, sizeof(double) * (size_t)A->nnz, hipMemcpyDeviceToDevice, c.stream);
    const int blocks = (int)std::min<int64_t>((A->nrows + 255) / 256, 4096);
    hipLaunchKernelGGL(expand_rows_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, c.stream, A->nrows, A->rowptr, d.cols);
  }
  mk_matrix *M = new mk_matrix();
  rc = device_coo_to_csr(&d, M, nullptr);
  dev_free(d.rows); dev_free(d.cols); dev_free(d.vals);
  if (rc != MK_OK) { mk_matrix_destroy(M); return rc; }
  *out = M;
  return MK_OK;
}


// ---- column relabelling for the split kernel (skewed matrices whose x does not fit the caches) ----------------
// The x gathers of a power-law matrix miss because hot columns are scattered: every cache line holds one hot entry
// among cold ones.  Columns are renumbered by descending in-degree class, so the hot entries of x sit in a few
// densely used lines (R-MAT scale 24: the 2^18 hottest columns carry 69 % of the entries).  The product reads
// xg[j] = x[order[j]] (one gather pass per product) through col2 = perm[col]: the same values multiplied and added
// in the same order -> bit-identical to the product on the original numbering; col[] stays for mk_matrix_get_csr.
// Column degrees.  One global atomic per entry took 31 ms for R-MAT scale 24 (2^28 entries): a device-scope atomic is carried
// out behind the XCDs' L2s (they are not coherent with each other), 32 bytes written through per atomic, 11.6 G atomics/s
// whatever the locality.  So every XCD counts into a copy of its own with atomics of WORKGROUP scope, which its L2 carries out
// in place (nobody else touches that copy before the kernel ends and the L2 is written back); the copy is picked by the
// hardware's XCC id, not by a guess about the dispatch order.  An LDS table in front takes the hottest columns (a column
// claims the slot it hashes to, later entries of the same column count there), flushed once per block.  A second kernel
// adds the eight copies.  Measured: 31 ms one atomic per entry, 23 ms with the LDS table, 12.9 ms with the per-XCD copies;
// giving every XCD an eighth of the COLUMNS instead (one set of counters, each XCD walks the whole entry stream) took 17 ms.
constexpr int kDegSlots = 8192;
constexpr int kDegCopies = 8;
static __global__ __launch_bounds__(256) void col_degree_kernel(int64_t nnz, int64_t ncols, const int *__restrict__ col,
                                                              unsigned *__restrict__ deg8, int sample) {
  __shared__ int skey[kDegSlots];
  __shared__ unsigned scnt[kDegSlots];
  for (int i = threadIdx.x; i < kDegSlots; i += blockDim.x) {
    skey[i] = -1;
    scnt[i] = 0;
  }
  // HW_REG_XCC_ID (id 20), bits 3:0: the XCD this wave runs on
  const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & (kDegCopies - 1);
  unsigned *deg = deg8 + (size_t)xcc * (size_t)ncols;
  __syncthreads();
  // sample > 1: every sample-th run of 256 entries only (the order of the columns needs degree classes, not degrees)
  const int64_t nruns = (nnz + 255) / 256;
  for (int64_t run = (int64_t)blockIdx.x * sample; run < nruns; run += (int64_t)gridDim.x * sample) {
    const int64_t e = run * 256 + threadIdx.x;
    if (e >= nnz) continue;
    const int c = col[e];
    const int slot = (int)(((unsigned)c * 2654435761u) >> 19);   // top 13 bits
    int owner = skey[slot];
    if (owner == -1) {
      owner = atomicCAS(&skey[slot], -1, c);
      if (owner == -1) owner = c;
    }
    if (owner == c)
      atomicAdd(&scnt[slot], 1u);
    else
      __hip_atomic_fetch_add(&deg[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kDegSlots; i += blockDim.x)
    if (scnt[i]) __hip_atomic_fetch_add(&deg[skey[i]], scnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

static __global__ void col_degree_sum_kernel(int64_t ncols, const unsigned *__restrict__ deg8, unsigned *__restrict__ deg) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * blockDim.x) {
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < kDegCopies; k++) s += deg8[(size_t)k * (size_t)ncols + c];
    deg[c] = s;
  }
}

static __global__ void col_degree_check_kernel(int64_t nnz, const int *__restrict__ col, unsigned *__restrict__ deg) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&deg[col[e]], 1u);
}

// key 0 = hottest: 4 classes per power of two of the degree (two mantissa bits), columns nobody refers to last
static __global__ void col_class_kernel(int64_t ncols, const unsigned *__restrict__ deg, int *__restrict__ key, int *__restrict__ id,
                                        double *__restrict__ dummy) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * blockDim.x) {
    const unsigned d = deg[c];
    int k = 0;
    if (d > 0) {
      const int e = 31 - __clz(d);
      const int m = e >= 2 ? (int)((d >> (e - 2)) & 3u) : (int)((d << (2 - e)) & 3u);
      k = 4 * e + m + 1;   // 1 .. 128
    }
    key[c] = 128 - k;
    id[c] = (int)c;
    dummy[c] = 0.0;
  }
}

static __global__ void invert_order_kernel(int64_t ncols, const int *__restrict__ order, int *__restrict__ perm) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < ncols; j += (int64_t)gridDim.x * blockDim.x) perm[order[j]] = (int)j;
}

static __global__ void relabel_cols_kernel(int64_t nnz, const int *__restrict__ col, const int *__restrict__ perm, int *__restrict__ col2) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) col2[e] = perm[col[e]];
}

int build_column_relabel(mk_matrix *A) {
  Context &c = ctx();
  const int64_t nc = A->ncols, nnz = A->nnz;
  unsigned *deg = nullptr, *deg8 = nullptr;
  int *perm = nullptr;
  mk_coo d;
  d.nrows = 129; d.ncols = nc; d.nnz = nc;
  int rc = dev_alloc(&deg, (size_t)nc);
  if (rc == MK_OK) rc = dev_alloc(&deg8, (size_t)nc * kDegCopies);
  if (rc == MK_OK) rc = dev_alloc(&perm, (size_t)nc);
  if (rc == MK_OK) rc = dev_alloc(&d.rows, (size_t)nc);
  if (rc == MK_OK) rc = dev_alloc(&d.cols, (size_t)nc);
  if (rc == MK_OK) rc = dev_alloc(&d.vals, (size_t)nc);
  mk_matrix M;   // "CSR" of (class, column id): its col array is the new order of the columns
  if (rc == MK_OK) {
    (void)hipMemsetAsync(deg8, 0, sizeof(unsigned) * (size_t)nc * kDegCopies, c.stream);
    const int be = (int)std::max<int64_t>(1, std::min<int64_t>((nnz + 255) / 256, 8192));
    const int bc = (int)std::max<int64_t>(1, std::min<int64_t>((nc + 255) / 256, 4096));
    static const int env_sample = getenv("MK_RELABEL_SAMPLE") ? atoi(getenv("MK_RELABEL_SAMPLE")) : 0;
    const bool checking = getenv("MK_RELABEL_CHECK") && atoi(getenv("MK_RELABEL_CHECK"));
    // default: every 8th run above 64 M entries, every 4th above 16 M.  R-MAT scale 24 (2^28 entries), from the end of the
    // conversion to the end of the first product / steady product: all entries 17.7 / 2.39 ms, 1 in 4 8.6 / 2.39, 1 in 8
    // 7.0 / 2.41, 1 in 16 6.1 / 2.41, 1 in 64 5.6 / 2.47
    const int sample = checking ? 1 : env_sample > 0 ? env_sample : nnz >= ((int64_t)1 << 26) ? 8 : nnz >= ((int64_t)1 << 24) ? 4 : 1;
    hipLaunchKernelGGL(col_degree_kernel, dim3(std::min(be, 2048)), dim3(256), 0, c.stream, nnz, nc, A->col, deg8, sample);
    hipLaunchKernelGGL(col_degree_sum_kernel, dim3(bc), dim3(256), 0, c.stream, nc, deg8, deg);
    if (getenv("MK_RELABEL_CHECK") && atoi(getenv("MK_RELABEL_CHECK"))) {   // tests: the degrees against device-scope atomics
      (void)hipMemsetAsync(deg8, 0, sizeof(unsigned) * (size_t)nc, c.stream);
      hipLaunchKernelGGL(col_degree_check_kernel, dim3(be), dim3(256), 0, c.stream, nnz, A->col, deg8);
      std::vector<unsigned> h1((size_t)nc), h2((size_t)nc);
      hipError_t e1 = hipMemcpyAsync(h1.data(), deg, sizeof(unsigned) * (size_t)nc, hipMemcpyDeviceToHost, c.stream);
      hipError_t e2 = hipMemcpyAsync(h2.data(), deg8, sizeof(unsigned) * (size_t)nc, hipMemcpyDeviceToHost, c.stream);
      (void)hipStreamSynchronize(c.stream);
      if (e1 != hipSuccess || e2 != hipSuccess || h1 != h2) rc = fail(MK_ERR_HIP, "column relabelling: per-XCD degree counts differ from the device-scope count");
    }
    hipLaunchKernelGGL(col_class_kernel, dim3(bc), dim3(256), 0, c.stream, nc, deg, d.rows, d.cols, d.vals);
    if (rc == MK_OK) rc = device_coo_to_csr(&d, &M, nullptr);   // stable: columns of one class keep their order
    if (rc == MK_OK) rc = dev_alloc(&A->col2, (size_t)nnz);
    if (rc == MK_OK) rc = dev_alloc(&A->xg, (size_t)nc);
    if (rc == MK_OK) {
      hipLaunchKernelGGL(invert_order_kernel, dim3(bc), dim3(256), 0, c.stream, nc, M.col, perm);
      hipLaunchKernelGGL(relabel_cols_kernel, dim3(be), dim3(256), 0, c.stream, nnz, A->col, perm, A->col2);
      A->xorder = M.col;
      M.col = nullptr;
      hipError_t e = hipStreamSynchronize(c.stream);
      if (e == hipSuccess) e = hipGetLastError();
      if (e != hipSuccess) rc = fail(MK_ERR_HIP, "column relabelling: %s", hipGetErrorString(e));
    }
  }
  dev_free(deg); dev_free(deg8); dev_free(perm); dev_free(d.rows); dev_free(d.cols); dev_free(d.vals);
  dev_free(M.rowptr); dev_free(M.col); dev_free(M.val);
  if (rc != MK_OK) { dev_free(A->col2); dev_free(A->xg); dev_free(A->xorder); A->col2 = nullptr; A->xg = nullptr; A->xorder = nullptr; }
  return rc;
}

}  // namespace mk
