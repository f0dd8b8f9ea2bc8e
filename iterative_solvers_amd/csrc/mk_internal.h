// mk_internal.h -- shared host/device plumbing of libmi355_krylov.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mk_krylov.h"

namespace mk {

// ----------------------------------------------------------------------------- errors
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define MK_HIP(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return mk::fail(MK_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,            \
                      hipGetErrorString(e_));                                              \
  } while (0)

#define MK_TRY(expr)                      \
  do {                                    \
    int rc_ = (expr);                     \
    if (rc_ != MK_OK) return rc_;         \
  } while (0)

#define MK_REQUIRE_INIT()                                                         \
  do {                                                                            \
    if (!mk::ctx().ready) return mk::fail(MK_ERR_NOT_INIT, "mk_init() not called"); \
  } while (0)

// ---------------------------------------------------------------------------- context
struct Context {
  bool ready = false;
  int device = -1;
  int num_cus = 256;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // pinned staging for small read-backs (solver state)
  void *pinned = nullptr;
  size_t pinned_bytes = 0;
  // scratch for stand-alone fused calls (mk_spmv_dot): partial slab + 2 results
  double *slab = nullptr;
  double *scalars = nullptr;
};
Context &ctx();

// Device memory comes from a caching allocator (context.hip): hipFree costs 0.2 ms a call on this stack whatever the size
// (it waits for the device, then unmaps), and a conversion or a solver setup frees half a dozen arrays.  Blocks of 1 MB and
// more go to a size-ordered free list instead and are handed out again to the next request they fit (at most 1/8 larger);
// INVARIANT the reuse rests on: a block's old user and its new user are ordered on ctx().stream.  Work on any other
// stream (RCCL communication streams, the local world's side stream) must be joined back into ctx().stream by events
// before the API call that issued it returns; such streams are registered (register_side_stream) and cached_free waits
// for any of them that is still busy before it keeps a block for reuse.  Host copies into fresh blocks go through
// ctx().stream (hipMemcpyAsync + synchronize), never the null stream.  The cache holds at most a quarter of the device's memory (MK_ALLOC_CACHE_MB), gives everything back
// before an allocation would fail and at mk_finalize(); MK_NO_ALLOC_CACHE=1 switches it off.
void *cached_alloc(size_t bytes);
void cached_free(void *p);
void cache_release_all();
void register_side_stream(hipStream_t s);
void unregister_side_stream(hipStream_t s);
template <typename T>
int dev_alloc(T **p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;  // keep pointers valid for empty objects
  // 256 B of slack: the LDS-staged kernels read whole 16-byte vectors around a chunk
  *p = static_cast<T *>(cached_alloc(count * sizeof(T) + 256));
  if (!*p) return fail(MK_ERR_ALLOC, "hipMalloc(%zu bytes) failed", count * sizeof(T));
  return MK_OK;
}
inline void dev_free(void *p) {
  if (p) cached_free(p);
}

// kernel argument (by value) of spmv_sweep_kernel: the one base pattern every row is a sub-list of,
// each offset split as a * P + b (P = sweep stride, e.g. one grid plane; |b| small), and the geometry
struct SweepArgs {
  int len;        // entries of the base pattern
  int P;          // rows between consecutive sweep steps of a block (even); == C in contiguous mode
  int C;          // rows per chunk (512 or 1024 = two per lane of a 256- or 512-thread block)
  int ncp;        // chunks of C rows per plane
  int nseg;       // segments the K planes are cut into (grid = ncp * nseg blocks, set per launch)
  int wlo, sw;    // window: rows [r0 - wlo, r0 - wlo + sw) of a plane per slab (both even)
  int K;          // planes = ceil(nrows / P) (per launch: planes of the launch's range)
  int kb;         // first plane of the launch (0 unless a caller sweeps a sub-range of the planes)
  int diag;       // index of the last entry with offset 0, or -1
  double dv;      // its value
  int gap0, gap1; // MODE 6: local rows [gap0, gap1) have a zero `extra` (set per launch)
  int rpl;        // vsweep kernel: rows per lane (2; 1 = long rows: C = T = 512, one block per CU)
};
struct SweepEntry {   // one entry of the base pattern.  Device table: 4 copies x 32 entries (padded with zeros);
  int off;            // copy c serves the steps whose plane sits in ring slot c: off = byte offset in the LDS ring
  int off1;           // of the x entry this matrix entry multiplies, for the lane that owns row r0 of the chunk (off1: the
                      // operand of the lane's second row, spmv_sweep_kernel's de-interleaved slabs only)
  double v;
};

// solver state of the fused CG iteration, one record in device memory (krylov_kernels.h, spmv.hip)
struct PcgState {
  double gamma, gamma_old, alpha, pp, i_prod, bi_prod, eps, sdotp;
  long long iter;
  int done;        // 0 running, 1 converged, 2 breakdown
  int converged;
  int breakdown;   // 1 zero <s,p>, 2 subnormal alpha, 3 subnormal gamma
  int x_pending;   // deferred solution update (vsweep CG): x += alpha p of the last completed step has not been applied yet
  int pbuf;        // ... and which of the two p buffers that p is in
  int pad;
  double sr_alpha[2], sr_gamma[2];   // single-reduction step: alpha and gamma of the last two steps by parity (read [parity ^ 1], write [parity])
  double beta_last, xx_last;         // halved solution update (sweep MODE 7 / 8): beta and <x,x> of the step's K3
};

// grid-level reduction slabs: component k of block b lives at slab[k * stride + b]
constexpr int kMaxParts = 4;
constexpr int kSlabStride = 4096;  // >= any reduction grid we launch
constexpr int kSplitChunk = 2048;  // entries per chunk of the split SpMV kernel

// ---------------------------------------------------------------------------- objects
}  // namespace mk

// opaque C types (global namespace, matching the header's forward declarations)
struct mk_vector {
  int64_t n = 0;      // logical length (owned rows)
  int64_t cap = 0;    // allocated entries (>= n; distributed vectors carry their ghost entries behind the owned ones)
  double *d = nullptr;
};

struct mk_coo {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  int *rows = nullptr;
  int *cols = nullptr;
  double *vals = nullptr;
};

struct mk_matrix {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  int *rowptr = nullptr;   // nrows+1
  int *col = nullptr;      // nnz
  double *val = nullptr;   // nnz
  double *dinv = nullptr;  // 1/diag, built on demand (Jacobi preconditioner)
  // selector state
  mk_matrix_info info{};
  bool analysed = false;
  int kernel = MK_SPMV_AUTO;
  int lanes = 0;
  // stream kernel (fixed rows per chunk): lanes per row and LDS capacity in entries
  int stream_lanes = 0;
  int stream_cap = 0;
  mutable int stream_bpc[3] = {0, 0, 0};
  mutable int pattern_bpc[3] = {0, 0, 0};   // resident blocks per CU of the stream kernel per MODE (occupancy query, cached)
  // column-index compression for the stream kernel: rows that share the same list of
  // (column - row) offsets share one table entry; the kernel then reads 2 B per row instead
  // of 4 B per entry.  pat == nullptr -> not compressed.
  unsigned short *pat = nullptr;   // pattern id per row
  int *ptab = nullptr;             // npat x (pml+1) ints: length, offsets
  int *stab = nullptr;             // stream kernel's two-level copy when patterns share shapes: [shapes: length, offsets - first offset][npat x (shape start, first offset)]
  int stab_shape_ints = 0, stab_ints = 0;
  int *colc = nullptr;             // stream kernel, rows that repeat the column list of the row before them (the unknowns of one mesh node):
  int *cptr = nullptr;             // colc = the lists stored once, cptr[row] = where row's list starts in colc
  int64_t colc_len = 0;
  // row-group kernel (group.hip): runs of up to 4 consecutive rows with one column list; grow[g] = first row of group g
  // (ngroups + 1 entries), gcol[g] = start of its list in colc
  int *grow = nullptr, *gcol = nullptr;
  int *gdesc = nullptr, *cdesc = nullptr;   // 16 bytes per group / per chunk: what a lane / a block loads of them (group.hip)
  int ngroups = 0, group_lanes = 0, group_cap = 0, group_ccap = 0, group_chunks = 0;
  mutable int group_bpc[2] = {0, 0};
  // band kernel (band.hip): chunks of band_rows consecutive rows whose columns lie within band_window entries behind the running
  // maximum column; band_desc = 16 bytes per chunk (first entry, running maximum, first new column, 0), band_bstart = first chunk
  // of each of the band_blocks blocks
  int *band_desc = nullptr, *band_bstart = nullptr;
  int band_rows = 0, band_chunks = 0, band_blocks = 0, band_window = 0, band_lanes = 0, band_cap = 0;
  // ... or, when one window would not fit the LDS, the matrix cut into nstrips diagonal strips (CSR matrices of their own, each
  // with its band plan; band.hip build_strips)
  mk_matrix *strips[12] = {};
  int nstrips = 0;
  double *vtab = nullptr;          // npat x pml doubles when the rows of a pattern also share their values
                                   // (pattern kernel: neither col[] nor val[] nor rowptr[] is read)
  int *otab8 = nullptr;            // pattern kernel's padded copies of ptab / vtab (row stride p8 + 8 ints, p8 doubles)
  double *vtab8 = nullptr;
  int p8 = 0;
  mk::SweepArgs *sweep = nullptr;  // host copy of the sweep kernel's plan (nullptr: not eligible)
  unsigned *pmask = nullptr;       // device, npat: which entries of the base pattern each pattern keeps
  mk::SweepEntry *etab = nullptr;  // device, 32 entries
  bool sweep_auto = false;         // enough parallelism for the sweep kernel to be the default
  int *rowpair = nullptr;          // vsweep kernel: rowptr[0], rowptr[2], ... (first value of every lane's row pair)
  bool sweep_streamed = false;     // the plan matches column offsets only: values are streamed from val[] (spmv_vsweep_kernel)
  unsigned short *sweep_perm = nullptr;   // sweep kernel, rows of more than 8 entries: row pair each lane of a chunk owns (ncp x C/2)
  mutable int sweep_bpc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool sweep_diag_uniform = false; // every row has the (single) diagonal entry of the base pattern
  int npat = 0, pml = 0;
  int max_off = 0;                 // largest |column - row| over all patterns
  // split kernel: chunk c covers entries [c*kSplitChunk, (c+1)*kSplitChunk); chunk_row[c] = first row starting in it
  // split kernel on matrices whose x outgrows the caches: columns renumbered hot-first (convert.hip build_column_relabel)
  int *col2 = nullptr;       // perm[col[e]]
  int *xorder = nullptr;     // new position -> old column
  double *xg = nullptr;      // x in the new order, rebuilt by every product
  int relabel_tried = 0;
  int *chunk_row = nullptr;
  int nchunks = 0;
  double *carry = nullptr;   // 2 per chunk: partial of the row entering from the left, partial of the row leaving to the right
  // long rows (split over several blocks)
  int *long_rows = nullptr;
  int n_long = 0;
};

namespace mk {

// nonzero: the SpMV launchers work out their grid (*nblocks) and return without enqueueing anything
int &dry_launch();
int analyse_matrix(mk_matrix *A);
int init_static_scratch(int device);   // spmv.hip: address of the file-scope scratch array of mk_spmv_dot_static on this GPU
int ensure_dinv(mk_matrix *A);

// launchers shared between translation units (all enqueue on ctx().stream)
// y = alpha*A*x + beta*y
// `skip` (may be null): device flag, nonzero -> the kernel returns at once
int launch_spmv(const mk_matrix *A, double alpha, const double *x, double beta, double *y, const int *skip = nullptr);
// y = A*x, partial sums of <y,x>,<x,x> into slab (components 0,1); returns blocks used.
// `done` (may be null) is a device flag: nonzero -> kernel returns at once.
int launch_spmv_dot(const mk_matrix *A, const double *x, double *y, double *slab, const int *done, int *nblocks);

// xnew = xold + D^-1 (rhs - A xold - extra) with the partial of the squared residual in slab component 0;
// MK_ERR_UNSUPPORTED (no error text) when A is not served by the stream kernel
int launch_jacobi_sweep(const mk_matrix *A, const double *rhs, const double *extra, const double *xold, double *xnew,
                        double *slab, const int *done, int *nblocks, int extra_gap0 = 0, int extra_gap1 = 0,
                        int plane_begin = 0, int plane_end = 0, int max_blocks = kSlabStride);
int jacobi_sweep_blocks(const mk_matrix *A, int plane_begin, int plane_end, int max_blocks);
// planes of the sweep plan (rows per plane, number of planes); 0 when the matrix is not served by the sweep kernel
int sweep_planes(const mk_matrix *A, int *rows_per_plane, int *nplanes);

// fused CG on matrices served by the sweep kernel: two more modes of it, the scalar logic of the step (alpha,
// breakdown tests, iteration counter) runs in the second one's prologue.
//   dots   : partials of <A p, p>, <p,p> into slab_out, nothing stored
//   resid_x: r -= alpha * A p with alpha = gamma / <A p,p> from slab_in (stride_in between its two components: 1 for
//            an all-reduced pair), x += alpha p, partials of <r,z>, <r,r>, <x,x>
int launch_sweep_cg_dots(const mk_matrix *A, const double *p, double *slab_out, PcgState *st, int *nblocks);
//   pdots  : K3 folded into the window fill of `dots`: prologue = K3's (gamma', convergence test, beta from slab_in, 3
//            components), p_new = zscale * r + beta * p_old written for the block's own rows and used as the window,
//            then the partials of <A p_new, p_new>, <p_new, p_new>.  zscale = 1, or 1/diag when sweep_uniform_diag().
int launch_sweep_cg_pdots(const mk_matrix *A, const double *r, const double *p_old, double *p_new, double zscale,
                          const double *slab_in, int nblocks_in, int stride_in, double *slab_out, PcgState *st, int two_norm,
                          int rel_change, double *norms, int *nblocks);
int sweep_uniform_diag(const mk_matrix *A, double *inv_diag);
//   single : the single-reduction step (opt-in): both halves of the step in one sweep behind ONE reduction of five values
//            (<A r, r>, <r,z>, <r,r>, <x,x>, <p,p>); see spmv_sweep_kernel MODE 6.  r and p are ping-ponged by the caller.
int sweep_single_ok(const mk_matrix *A);   // MODE 6 is available for this matrix's sweep plan
int launch_sweep_cg_single(const mk_matrix *A, const double *r_old, double *r_new, const double *p_old, double *p_new, double *x,
                           double zscale, const double *extra, int gap0, int gap1, const double *sr_in, double *slab_out,
                           PcgState *st, int two_norm, int rel_change, int first, int parity, double *norms, int *nblocks, int xmode = 0,
                           int pbuf = 0);
//            xmode 0: x += alpha p every step.  1 / 2 (alternating, from step 0 on): the halved solution update -- step 2k leaves
//            x alone (pending in the state record), step 2k + 1 adds both steps' updates (see spmv_sweep_kernel MODE 7 / 8);
//            parity = step & 1, pbuf = which of the caller's two p buffers `p` is (for pcg_apply_pending_x_kernel)
int launch_sweep_cg_resid_x(const mk_matrix *A, const double *p, double *r, double *x, const double *dinv, const double *slab_in,
                            int nblocks_in, int stride_in, double *slab_out, PcgState *st, int *nblocks, int xmode = 0, int parity = 0,
                            int pbuf = 0);

// sweep over grid matrices with per-row values (vsweep.hip); mode 0: y = alpha A x + beta y, 1: y = A x + partials of
// <y,x>, <x,x>, 2: Jacobi sweep (as launch_jacobi_sweep)
int launch_vsweep(const mk_matrix *A, int mode, double alpha, const double *x, double beta, double *y, double *slab, const int *done,
                  int *nblocks, const double *rhs, const double *extra);

// CG step on vsweep matrices: K3 (p' = z + beta p, z = r or r * dinv) rides in the window fill of the product;
// s = A p', partials of <s,p'>, <p',p'>; prologue = K3's convergence test on the 3 components of slab_in
int build_vsweep_rowpair(mk_matrix *A);
int launch_vsweep_cg_pdots(const mk_matrix *A, const double *r, const double *p_old, double *p_new, const double *dinv, double *s,
                           const double *slab_in, int nblocks_in, int stride_in, double *slab_out, PcgState *st, int two_norm,
                           int rel_change, double *norms, int *nblocks, double *xupd = nullptr);

// rows that repeat their neighbour's column list: groups of such rows, served by spmv_group_kernel (group.hip)
int build_row_groups(mk_matrix *A);
int launch_group(const mk_matrix *A, int mode, double alpha, const double *x, double beta, double *y, double *slab, const int *done, int *nblocks);

// rows that stay within a bounded distance of the diagonal: x window of a row panel in LDS, served by spmv_band_kernel (band.hip)
int build_band_plan(mk_matrix *A);
int launch_band(const mk_matrix *A, int mode, double alpha, const double *x, double beta, double *y, double *slab, const int *done, int *nblocks);

int exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *total_out_dev);

// device COO -> CSR (stable by row).  Takes ownership of rows/cols/vals buffers of `coo`
// (they are freed or become the CSR arrays).  sorted_rows_out (optional) receives the
// sorted row keys (device buffer of nnz ints owned by caller afterwards).
int device_coo_to_csr(mk_coo *coo, mk_matrix *A, int **sorted_rows_out);
int build_column_relabel(mk_matrix *A);
int build_transpose(const mk_matrix *A, mk_matrix **out);   // A^T through the same conversion (cgnr.hip)

}  // namespace mk

// ------------------------------------------------------------------- device helpers
#if defined(__HIPCC__)
namespace mk {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// sum within aligned groups of G lanes (G power of two <= 64); result valid in the group's lane 0
template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_down(v, off, G);
  return v;
}

// Block-wide sums of NP values; every thread gets the totals in out[].  Fixed order:
// lanes by shuffle tree, then waves 0..nw-1 left to right -> bitwise reproducible.
template <int NP, int THREADS>
__device__ __forceinline__ void block_sum(double (&v)[NP], double *smem /* NP*(THREADS/64) */) {
  constexpr int NW = THREADS / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NP; k++) {
    double s = wave_sum(v[k]);
    if (lane == 0) smem[k * NW + w] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NP; k++) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; i++) s += smem[k * NW + i];
    v[k] = s;
  }
  __syncthreads();
}

// Block-uniform read of a stop flag.  The leader of block 0 may raise the flag inside the same launch (breakdown,
// convergence); waves of one block that read it themselves could then disagree, one leaving while its siblings go
// on into a block-wide reduction over LDS slots that were never written.  One thread reads, everybody takes its
// answer.  Call at most once per kernel, before any other use of shared memory is in flight.
__device__ __forceinline__ bool block_flag_raised(const int *flag) {
  __shared__ int sh_flag;
  if (threadIdx.x == 0) sh_flag = *flag;
  __syncthreads();
  return sh_flag != 0;
}

// Vectors of this many rows no longer stay in the 256 MB Infinity Cache between the kernels of an iteration
// (4 vectors x 8 B x 10 M rows = 320 MB): single-use streams then go nontemporal.  Measured crossover on the sweep
// path and on the K2/K3 vector kernels alike: 8.4 M rows default policy 2 % ahead, 16.7 M rows nontemporal 5-9 % ahead.
static inline int nt_force() {
  static const int force = getenv("MK_NT") ? atoi(getenv("MK_NT")) : -1;   // 0 / 1: never / always (A/B, tests)
  return force;
}
static inline bool nt_rows(int64_t n) { return nt_force() >= 0 ? nt_force() != 0 : n > 10 * 1000 * 1000; }
// the matrix streams (val + col) of the stream and split kernels: nontemporal once they exceed the cache
static inline bool nt_matrix(double bytes) { return nt_force() >= 0 ? nt_force() != 0 : bytes > 200e6; }

// Sum the per-block partials a previous kernel left in a slab (launch-boundary reduce:
// no in-launch hand-off, so no fences).  Every block computes the same bits.
template <int NP, int THREADS>
__device__ __forceinline__ void reduce_slab(const double *slab, int nblocks, int stride, double (&out)[NP], double *smem) {
#pragma unroll
  for (int k = 0; k < NP; k++) out[k] = 0.0;
  // four loads per component in flight (a serial chain of dependent loads made the one-block finalize kernel of
  // the row-block solver 7 us long); added in index order, so the bits do not depend on the batching
  for (int i0 = threadIdx.x; i0 < nblocks; i0 += 4 * THREADS) {
    double t[NP][4];
#pragma unroll
    for (int k = 0; k < NP; k++)
#pragma unroll
      for (int u = 0; u < 4; u++) t[k][u] = slab[k * stride + min(i0 + u * THREADS, nblocks - 1)];
#pragma unroll
    for (int k = 0; k < NP; k++)
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (i0 + u * THREADS < nblocks) out[k] += t[k][u];
  }
  block_sum<NP, THREADS>(out, smem);
}

}  // namespace mk
#endif
