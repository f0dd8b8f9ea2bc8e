// group.hip -- CSR SpMV for matrices whose consecutive rows repeat one column list (row-group kernel).
//
// Stands in for hypre_ParCSRMatrixMatvec on the matrices the reference's authors benchmark (scripts/bench_all_gpu:2:
// Bump_2911, Serena, Geo_1438, Flan_1565 ...: finite-element meshes with three unknowns per node).  The rows of one node
// couple to the same nodes, so they carry the same column list; try_share_columns (spmv.hip) finds such rows and stores
// each list once.  The stream kernel still walked every row for itself: it gathered x once per ROW, and what bounds it
// on these matrices is not HBM but the L2 -> L1 line traffic of those gathers.  Here the lanes that walk a
// list gather x ONCE per list and multiply it with the values of every row that shares the list:
//   group  = up to kGroupMax consecutive rows with one column list (a longer run is cut into several groups);
//   chunk  = 256 / G consecutive groups; their values are one contiguous span of val[] (the rows are consecutive and of
//            equal length inside a group), staged in LDS by coalesced 16-byte loads as in the stream kernel; the column
//            lists of the chunk are one contiguous span of the compact list array;
//   lanes  = G per group: lane g takes list entries g, g + G, ...; per entry one gather of x and one multiply-add per row.
// HBM bytes per entry: 8 (value) + 4 / rows-per-group (column); per row y, x; per group 8 B of group tables.  Gather
// requests per entry: 1 / rows-per-group of the stream kernel's.  Sums are formed per lane over its entries in list order,
// then across the G lanes by a shuffle tree: the stream kernel's order for the same G, so results agree with it bit for bit
// where both run with equal G, and with the serial reference to 1e-12 (tests).
// Algorithmic bytes: SURVEY 8(d)'s 12 nnz + 20 n + 4.
#include "mk_internal.h"

namespace mk {

constexpr int kGThreads = 256;
constexpr int kGroupMax = 4;   // rows per group (sums a lane keeps)
typedef double gd2_t __attribute__((ext_vector_type(2)));
typedef int gi4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------- building the groups
// start[r] = 1 where a group begins: at every row that does not repeat its predecessor's list, and every kGroupMax rows
// inside a longer run.  One thread per run leader walks its run (runs are a few rows long).
__global__ void group_mark_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ cptr, uint32_t *__restrict__ start) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    const int len = rowptr[r + 1] - rowptr[r];
    const bool follower = r > 0 && len > 0 && cptr[r] == cptr[r - 1] && rowptr[r] - rowptr[r - 1] == len;
    if (follower) continue;
    start[r] = 1u;
    int k = 1;
    for (int64_t q = r + 1; q < nrows; q++, k++) {
      const int lq = rowptr[q + 1] - rowptr[q];
      if (!(lq > 0 && lq == len && cptr[q] == cptr[r])) break;
      start[q] = (k % kGroupMax == 0) ? 1u : 0u;
    }
  }
}

__global__ void group_fill_kernel(int64_t nrows, const uint32_t *__restrict__ start, const uint32_t *__restrict__ scan, uint32_t ngroups,
                                  const int *__restrict__ cptr, int colc_len, int *__restrict__ grow, int *__restrict__ gcol) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= nrows; r += (int64_t)gridDim.x * blockDim.x) {
    if (r == nrows) { grow[ngroups] = (int)nrows; gcol[ngroups] = colc_len; continue; }
    if (start[r]) { grow[scan[r]] = (int)r; gcol[scan[r]] = cptr[r]; }
  }
}

// What a lane needs of its group in ONE 16-byte load (row0, rows | list length << 8, first value, list start), and what a block
// needs of a chunk in one scalar load (value span, list span): the kernel fetches both one chunk ahead, so no chain of
// dependent loads (group -> row -> row pointer) sits between the barrier and the gathers.
__global__ void group_desc_kernel(int ngroups, const int *__restrict__ rowptr, const int *__restrict__ grow, const int *__restrict__ gcol,
                                  gi4_t *__restrict__ gdesc) {
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x) {
    const int r0 = grow[g], cnt = grow[g + 1] - r0;
    const int p0 = rowptr[r0], len = rowptr[r0 + 1] - p0;
    gi4_t d;
    d.x = r0; d.y = cnt | (len << 8); d.z = p0; d.w = gcol[g];
    gdesc[g] = d;
  }
}
// A base chunk is R = 256 / G consecutive groups.  One whose values do not fit the LDS stage is cut into 2, 4, ... equal
// runs of groups (the lanes of the missing groups idle in those chunks): nsub[c] = pieces of base chunk c, 0 = a single
// group alone exceeds the stage (the matrix then stays on the stream kernel).
__global__ void chunk_split_kernel(int nbase, int R, int ngroups, int cap, const int *__restrict__ rowptr, const int *__restrict__ grow,
                                   uint32_t *__restrict__ nsub, int *__restrict__ fail) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nbase; c += gridDim.x * blockDim.x) {
    const int g0 = c * R, g1 = min(g0 + R, ngroups);
    int n = 1;
    for (; n <= R; n *= 2) {
      const int per = R / n;
      bool ok = true;
      for (int a = g0; a < g1 && ok; a += per) ok = rowptr[grow[min(a + per, g1)]] - rowptr[grow[a]] <= cap;
      if (ok) break;
    }
    if (n > R) { n = 0; *fail = 1; }
    else n = (g1 - g0 + R / n - 1) / (R / n);   // pieces the (possibly partial) base chunk really has
    nsub[c] = (uint32_t)n;
  }
}
// two int4 per chunk: {first value, end of values, first list entry, end of lists}, {first group, groups, 0, 0}; the largest
// value / list spans go to stats[0], stats[1]
__global__ void chunk_desc_kernel(int nbase, int R, int ngroups, const int *__restrict__ rowptr, const int *__restrict__ grow,
                                  const int *__restrict__ gcol, const uint32_t *__restrict__ nsub, const uint32_t *__restrict__ first_of,
                                  gi4_t *__restrict__ cdesc, int *__restrict__ stats) {
  int vm = 0, cm = 0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nbase; c += gridDim.x * blockDim.x) {
    const int g0 = c * R, g1 = min(g0 + R, ngroups);
    const int n = (int)nsub[c], per = R / max(n, 1);
    int k = (int)first_of[c];
    for (int a = g0; a < g1; a += per, k++) {
      const int z = min(a + per, g1);
      const int ra = grow[a], rz = grow[z], rl = grow[z - 1];
      gi4_t d, e;
      d.x = rowptr[ra];
      d.y = rowptr[rz];
      d.z = gcol[a];
      d.w = gcol[z - 1] + (d.y - rowptr[rl]) / max(rz - rl, 1);
      e.x = a; e.y = z - a; e.z = 0; e.w = 0;
      cdesc[2 * k] = d;
      cdesc[2 * k + 1] = e;
      vm = max(vm, d.y - d.x);
      cm = max(cm, d.w - d.z);
    }
  }
  if (vm > 0) atomicMax(&stats[0], vm);
  if (cm > 0) atomicMax(&stats[1], cm);
}
// Every chunk descriptor the product kernel will trust, checked once: spans inside the arrays and inside the stage, groups
// inside the table, consecutive chunks covering consecutive groups.  A violation switches the row-group path off.
__global__ void chunk_check_kernel(int nchunks, int ngroups, int R, int cap, int64_t nnz, int64_t ncl, const gi4_t *__restrict__ cdesc,
                                   const gi4_t *__restrict__ gdesc, int *__restrict__ bad) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nchunks; k += gridDim.x * blockDim.x) {
    const gi4_t d = cdesc[2 * k], e = cdesc[2 * k + 1];
    bool ok = d.x >= 0 && d.x <= d.y && d.y <= nnz && d.y - d.x <= cap && d.z >= 0 && d.z <= d.w && d.w <= ncl && d.w - d.z <= cap;
    ok = ok && e.x >= 0 && e.y >= 1 && e.y <= R && e.x + e.y <= ngroups;
    if (ok) {
      const int nextg = (k + 1 < nchunks) ? cdesc[2 * (k + 1) + 1].x : ngroups;
      ok = (e.x + e.y == nextg) && (k > 0 || e.x == 0);
      for (int q = e.x; q < e.x + e.y && ok; q++) {
        const gi4_t gq = gdesc[q];
        const int cnt = gq.y & 0xff, len = gq.y >> 8;
        ok = cnt >= 1 && cnt <= kGroupMax && len >= 0 && gq.z >= d.x && gq.z + cnt * len <= d.y && gq.w >= d.z && gq.w + len <= d.w && gq.x >= 0;
      }
    }
    if (!ok) *bad = 1;
  }
}

static int group_max_cap() {   // value entries per chunk: 4088 = 32 KB of LDS, three blocks of 134 VGPRs per CU (MK_GROUP_MAX_CAP: up to 8184 for A/B runs)
  static const int v = getenv("MK_GROUP_MAX_CAP") ? std::max(256, std::min(atoi(getenv("MK_GROUP_MAX_CAP")), 8160)) : 4064;
  return v;
}
#define kGroupMaxCap group_max_cap()

int build_row_groups(mk_matrix *A) {
  static const int off = getenv("MK_NO_ROW_GROUPS") ? atoi(getenv("MK_NO_ROW_GROUPS")) : 0;
  if (off || !A->cptr || !A->colc || A->grow || A->nrows < 2 || A->nnz >= ((int64_t)1 << 31) - 65536) return MK_OK;
  Context &c = ctx();
  uint32_t *start = nullptr, *scan = nullptr, *d_total = nullptr;
  int rc = MK_OK;
  auto done = [&](int code) {
    dev_free(start); dev_free(scan); dev_free(d_total);
    if (code != MK_OK || A->group_lanes == 0) {   // an optimisation only: without it the stream kernel serves the matrix
      dev_free(A->grow); dev_free(A->gcol); dev_free(A->gdesc); dev_free(A->cdesc);
      A->grow = nullptr; A->gcol = nullptr; A->gdesc = nullptr; A->cdesc = nullptr; A->ngroups = 0; A->group_lanes = 0;
    }
    return MK_OK;
  };
  if ((rc = dev_alloc(&start, (size_t)A->nrows)) != MK_OK) return done(rc);
  if ((rc = dev_alloc(&scan, (size_t)A->nrows)) != MK_OK) return done(rc);
  if ((rc = dev_alloc(&d_total, 1)) != MK_OK) return done(rc);
  MK_HIP(hipMemsetAsync(start, 0, sizeof(uint32_t) * (size_t)A->nrows, c.stream));
  const int blocks = (int)std::min<int64_t>((A->nrows + 255) / 256, 4096);
  hipLaunchKernelGGL(group_mark_kernel, dim3(blocks), dim3(256), 0, c.stream, A->nrows, A->rowptr, A->cptr, start);
  if ((rc = exclusive_scan_u32(start, scan, (size_t)A->nrows, d_total)) != MK_OK) return done(rc);
  uint32_t ng = 0;
  MK_HIP(hipMemcpyAsync(&ng, d_total, sizeof(ng), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  // worth it from an average of 1.5 rows per group on
  if (ng < 1 || (double)A->nrows < 1.5 * (double)ng) return done(MK_OK);
  if ((rc = dev_alloc(&A->grow, (size_t)ng + 1)) != MK_OK) return done(rc);
  if ((rc = dev_alloc(&A->gcol, (size_t)ng + 1)) != MK_OK) return done(rc);
  A->ngroups = (int)ng;
  hipLaunchKernelGGL(group_fill_kernel, dim3(blocks), dim3(256), 0, c.stream, A->nrows, start, scan, ng, A->cptr, (int)A->colc_len, A->grow, A->gcol);
  MK_HIP(hipGetLastError());
  // lanes per group: about 8 list entries per lane (one batch of gathers)
  const double mean_list = (double)A->colc_len / (double)ng;
  static const int env_g = getenv("MK_GROUP_LANES") ? atoi(getenv("MK_GROUP_LANES")) : 0;
  int G = 4;
  while (G < 64 && 8.0 * G < mean_list) G *= 2;
  // ... and the TYPICAL chunk of 256 / G groups must fit the stage (cutting is for the few fullest ones: a cut chunk leaves half
  // of the block's lanes idle)
  const double mean_group = (double)A->nnz / (double)ng;
  while (G < 64 && (double)(kGThreads / G) * mean_group > 0.97 * kGroupMaxCap) G *= 2;
  if (env_g >= 4 && env_g <= 64 && !(env_g & (env_g - 1))) G = env_g;
  if ((rc = dev_alloc(&A->gdesc, (size_t)ng * 4 + 16)) != MK_OK) return done(rc);
  MK_HIP(hipMemsetAsync(A->gdesc + (size_t)ng * 4, 0, 16 * sizeof(int), c.stream));
  hipLaunchKernelGGL(group_desc_kernel, dim3(std::min<int>((int)((ng + 255) / 256), 4096)), dim3(256), 0, c.stream, (int)ng, A->rowptr, A->grow, A->gcol,
                     reinterpret_cast<gi4_t *>(A->gdesc));
  // chunks: R groups each; a base chunk whose values exceed the stage is cut into equal runs of groups
  const int R = kGThreads / G, nbase = (int)((ng + R - 1) / R);
  uint32_t *nsub = nullptr, *first_of = nullptr;
  int *stats = nullptr, h_stats[4] = {0, 0, 0, 0};
  auto done2 = [&](int code) { dev_free(nsub); dev_free(first_of); dev_free(stats); return done(code); };
  if ((rc = dev_alloc(&nsub, (size_t)nbase)) != MK_OK) return done2(rc);
  if ((rc = dev_alloc(&first_of, (size_t)nbase)) != MK_OK) return done2(rc);
  if ((rc = dev_alloc(&stats, 4)) != MK_OK) return done2(rc);
  MK_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(int), c.stream));
  const int cblocks = std::min((nbase + 255) / 256, 4096);
  hipLaunchKernelGGL(chunk_split_kernel, dim3(cblocks), dim3(256), 0, c.stream, nbase, R, (int)ng, kGroupMaxCap, A->rowptr, A->grow, nsub, stats + 2);
  if ((rc = exclusive_scan_u32(nsub, first_of, (size_t)nbase, d_total)) != MK_OK) return done2(rc);
  uint32_t nchunks = 0;
  MK_HIP(hipMemcpyAsync(&nchunks, d_total, sizeof(nchunks), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipMemcpyAsync(h_stats, stats, sizeof(h_stats), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  if (h_stats[2] != 0 || nchunks < 1) return done2(MK_OK);   // a group larger than the stage: the stream kernel keeps the matrix
  if ((rc = dev_alloc(&A->cdesc, (size_t)nchunks * 8 + 16)) != MK_OK) return done2(rc);
  MK_HIP(hipMemsetAsync(A->cdesc, 0, ((size_t)nchunks * 8 + 16) * sizeof(int), c.stream));
  hipLaunchKernelGGL(chunk_desc_kernel, dim3(cblocks), dim3(256), 0, c.stream, nbase, R, (int)ng, A->rowptr, A->grow, A->gcol, nsub, first_of,
                     reinterpret_cast<gi4_t *>(A->cdesc), stats);
  hipLaunchKernelGGL(chunk_check_kernel, dim3(std::min<int>((int)((nchunks + 255) / 256), 4096)), dim3(256), 0, c.stream, (int)nchunks, (int)ng, R,
                     kGroupMaxCap, A->nnz, A->colc_len, reinterpret_cast<const gi4_t *>(A->cdesc), reinterpret_cast<const gi4_t *>(A->gdesc), stats + 3);
  MK_HIP(hipMemcpyAsync(h_stats, stats, sizeof(h_stats), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  MK_HIP(hipGetLastError());
  if (h_stats[3] != 0) return done2(MK_OK);   // (never expected: the kernel would read outside its arrays, so it does not run)
  A->group_lanes = G;
  A->group_chunks = (int)nchunks;
  A->group_cap = (std::max(64, h_stats[0]) + 3) & ~3;
  A->group_ccap = (std::max(64, h_stats[1]) + 3) & ~3;
  return done2(MK_OK);
}

// ------------------------------------------------------------------------------------------ the kernel
// MODE 0: y = alpha A x + beta y.  MODE 1: y = A x, per-block partials of <y,x>, <x,x> (the CG step's K1).
// U: capacity class of the value stage (cap <= 1024 U - 8), as in spmv_stream_kernel; the column stage holds ccap <= cap.
template <int G, int MODE, int U, bool NT>
__global__ __launch_bounds__(kGThreads) void spmv_group_kernel(
    int64_t nrows, int ngroups, int nchunks, const gi4_t *__restrict__ gdesc, const gi4_t *__restrict__ cdesc, const int *__restrict__ colc,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, double alpha,
    double beta, double *__restrict__ slab, const int *__restrict__ done, int cap, int ccap) {
  if (done != nullptr && *done != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // (values and lists are staged from the last cache line boundary in front of the chunk's first value / list entry -- 16
  // values, 32 columns --, so the stage holds up to 15 / 31 entries more than the chunk has; see spmv.hip kStreamAlignV)
  double *sval = reinterpret_cast<double *>(smem_raw);                      // cap + 18 doubles of stage, then a zero (below)
  int *scol = reinterpret_cast<int *>(smem_raw + (size_t)(cap + 20) * 8);   // ccap + 40 ints
  // Groups of EMPTY rows (and the lanes of a cut chunk that own no group) multiply nothing, but their lanes run the
  // multiply-adds of their wave with xv = 0: they must read a finite number, not whatever lies behind the staged span.
  if (threadIdx.x == 0) sval[cap + 18] = 0.0;   // (visible after the first barrier of the loop; never overwritten by the stage)
  constexpr int NV = 2 * U, NI = U;
  const int tid = threadIdx.x;
  const int g = tid % G, rl = tid / G;
  // blocks of one XCD (equal blockIdx % 8) take a contiguous range of chunks, interleaved among themselves
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int per_xcd = (nchunks + 7) / 8;
  const int first = xcd * per_xcd, last = min(first + per_xcd, nchunks);
  double acc_yx = 0.0, acc_xx = 0.0;
  gd2_t rv[NV];
  gi4_t ri[NI];
  int chunk = first + slot;
  // Staging addresses: a block-uniform base (scalar registers) plus the thread's own 32-bit byte offset, so that a load or
  // an LDS store costs one add and one compare.
  const unsigned toff = (unsigned)tid * 16u;
  auto stage_loads = [&](gi4_t cd) {   // cd: value span [x, y), list span [z, w)
    const int vb_ = cd.x & ~15, ib_ = cd.z & ~31;
    const char *vbase = reinterpret_cast<const char *>(val + vb_);
    const char *ibase = reinterpret_cast<const char *>(colc + ib_);
    const unsigned vbytes = (unsigned)(cd.y - vb_) * 8u, ibytes = (unsigned)(cd.w - ib_) * 4u;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const unsigned o = toff + (unsigned)i * 4096u;
      if (o < vbytes) rv[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const gd2_t *>(vbase + o)) : *reinterpret_cast<const gd2_t *>(vbase + o);
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const unsigned o = toff + (unsigned)i * 4096u;
      if (o < ibytes) ri[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const gi4_t *>(ibase + o)) : *reinterpret_cast<const gi4_t *>(ibase + o);
    }
  };
  // chunk and group descriptors run one chunk ahead of the data they describe; both arrays have slack behind their end
  // (chunk descriptors two chunks ahead, so that the group descriptor load of the next chunk, which needs that chunk's first
  // group, does not wait for a scalar load issued in the same step)
  gi4_t cd = {0, 0, 0, 0}, ce2 = {0, 0, 0, 0}, gd = {0, 0, 0, 0}, ncd = cd, nce2 = cd;
  if (chunk < last) {
    cd = cdesc[2 * chunk];
    ce2 = cdesc[2 * chunk + 1];
    const int nx = (chunk + slots < last) ? chunk + slots : chunk;
    ncd = cdesc[2 * nx];
    nce2 = cdesc[2 * nx + 1];
    gd = gdesc[min(ce2.x + rl, ngroups)];
    stage_loads(cd);
  }
  while (chunk < last) {
    const int cb = cd.x, ce = cd.y, ccb = cd.z, cce = cd.w;
    const int vb = cb & ~15, ib = ccb & ~31;
    {
      const unsigned vbytes = (unsigned)(ce - vb) * 8u, ibytes = (unsigned)(cce - ib) * 4u;
      char *sv8 = reinterpret_cast<char *>(sval), *sc8 = reinterpret_cast<char *>(scol);
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const unsigned o = toff + (unsigned)i * 4096u;
        if (o < vbytes) *reinterpret_cast<gd2_t *>(sv8 + o) = rv[i];
      }
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const unsigned o = toff + (unsigned)i * 4096u;
        if (o < ibytes) *reinterpret_cast<gi4_t *>(sc8 + o) = ri[i];
      }
    }
    const bool have = rl < ce2.y;
    const int row0 = gd.x, cnt = have ? (gd.y & 0xff) : 0, len = have ? (gd.y >> 8) : 0;
    const int b0 = have ? gd.z - cb : 0, cs = have ? gd.w - ib : 0;   // (the entry behind the last group is slack)
    // descriptors of the next chunk: requested now, used after this chunk's gathers have been issued
    const int next = chunk + slots;
    const bool more = next < last;
    const int next2 = (next + slots < last) ? next + slots : (more ? next : chunk);
    const gi4_t nncd = cdesc[2 * next2];
    const gi4_t nnce2 = cdesc[2 * next2 + 1];
    const gi4_t ngd = gdesc[min(nce2.x + rl, ngroups)];
    __syncthreads();
    const double *sv = (len > 0) ? sval + (cb - vb) + b0 : sval + cap + 18;   // sv[m * len + j]: value of list entry j in the group's m-th row
    const int *sc = scol + cs;
    // The lane's first 8 list entries: all gathers issued before anything waits.  Slots no lane of the wave needs are skipped
    // by wave-uniform branches (an exec-masked slot would still take its issue cycles: with 78-entry lists on 16 lanes five
    // of the eight slots are live, and the kernel is bound by instruction issue, not by memory).
    const int lastj = max(len - 1, 0);
    int cj[8];
    double xj[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
      cj[t] = 0;
      if (__any(g + t * G < len)) cj[t] = (g < len) ? sc[min(g + t * G, lastj)] : 0;
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
      xj[t] = 0.0;
      if (__any(g + t * G < len)) xj[t] = x[cj[t]];
    }
    // the lane that will hold row m's sum after the reduction below: the first lane of quarter m of the group
    constexpr int Q = G / 4;
    const int mq = g / Q;
    const bool writer = (g % Q == 0) && mq < cnt;
    const double xr = (MODE == 1 && writer) ? x[row0 + mq] : 0.0;
    const double yold = (MODE == 0 && writer && beta != 0.0) ? y[row0 + mq] : 0.0;
    // prefetch the next chunk (stays in flight while this one is reduced)
    if (more) stage_loads(ncd);
    double sum[kGroupMax];
#pragma unroll
    for (int m = 0; m < kGroupMax; m++) sum[m] = 0.0;
    // rows beyond the group's count read the last row's values (clamped) and are never stored
    const int mo1 = min(1, cnt - 1) * len, mo2 = min(2, cnt - 1) * len, mo3 = min(3, cnt - 1) * len;
    const bool w2 = __any(cnt > 1), w3 = __any(cnt > 2), w4 = __any(cnt > 3);   // wave-uniform
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int j = g + t * G;
      if (__any(j < len)) {
        const double xv = (j < len) ? xj[t] : 0.0;
        const double *p = sv + min(j, lastj);
        sum[0] += p[0] * xv;
        if (w2) sum[1] += p[mo1] * xv;
        if (w3) sum[2] += p[mo2] * xv;
        if (w4) sum[3] += p[mo3] * xv;
      }
    }
    for (int j0 = g + 8 * G; __any(j0 < len); j0 += 4 * G) {   // lists longer than 8 lanes' worth: batches of 4
      int c4[4];
      double x4[4];
#pragma unroll
      for (int t = 0; t < 4; t++) c4[t] = (g < len) ? sc[min(j0 + t * G, lastj)] : 0;
#pragma unroll
      for (int t = 0; t < 4; t++) x4[t] = x[c4[t]];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int j = j0 + t * G;
        const double xv = (j < len) ? x4[t] : 0.0;
        const double *p = sv + min(j, lastj);
        sum[0] += p[0] * xv;
        if (w2) sum[1] += p[mo1] * xv;
        if (w3) sum[2] += p[mo2] * xv;
        if (w4) sum[3] += p[mo3] * xv;
      }
    }
    // Four sums per lane over G lanes: a transposing butterfly (the halves of a group swap the two sums the other half keeps,
    // then the quarters, then a plain butterfly inside each quarter): 3 + log2(G / 4) exchanges instead of 4 log2(G).
    // Row m's total ends up in every lane of quarter m.
    double tot;
    {
      const bool lo = g < G / 2;
      const double r0 = __shfl_xor(lo ? sum[2] : sum[0], G / 2, G);
      const double r1 = __shfl_xor(lo ? sum[3] : sum[1], G / 2, G);
      const double a0 = (lo ? sum[0] : sum[2]) + r0;   // lower half: m = 0, 1; upper half: m = 2, 3
      const double a1 = (lo ? sum[1] : sum[3]) + r1;
      const bool q0 = (g & Q) == 0;
      const double rb = __shfl_xor(q0 ? a1 : a0, Q, G);
      tot = (q0 ? a0 : a1) + rb;
#pragma unroll
      for (int off = Q / 2; off > 0; off >>= 1) tot += __shfl_xor(tot, off, G);
    }
    if (writer) {
      double yv;
      if (MODE == 0) {
        yv = (beta == 0.0) ? alpha * tot : alpha * tot + beta * yold;
      } else {
        yv = tot;
        acc_yx += tot * xr;
        acc_xx += xr * xr;
      }
      if (NT) __builtin_nontemporal_store(yv, y + row0 + mq); else y[row0 + mq] = yv;
    }
    __syncthreads();
    chunk = next; cd = ncd; ce2 = nce2; gd = ngd; ncd = nncd; nce2 = nnce2;
  }
  if (MODE != 0) {
    double *red = reinterpret_cast<double *>(smem_raw);
    double v[2] = {acc_yx, acc_xx};
    block_sum<2, kGThreads>(v, red);
    if (threadIdx.x == 0) {
      slab[0 * kSlabStride + blockIdx.x] = v[0];
      slab[1 * kSlabStride + blockIdx.x] = v[1];
    }
  }
}

template <int MODE, int U, bool NT>
static int launch_group_u(const mk_matrix *A, double alpha, const double *x, double beta, double *y, double *slab, const int *done, int *nblocks) {
  Context &c = ctx();
  const int G = A->group_lanes, cap = A->group_cap, ccap = A->group_ccap;
  const size_t lds = (size_t)(cap + 20) * 8 + (size_t)(ccap + 40) * 4;
  const int nchunks = A->group_chunks;
  static const int env_bpc = getenv("MK_GROUP_BPC") ? atoi(getenv("MK_GROUP_BPC")) : 0;
  int per_cu = env_bpc > 0 ? env_bpc : A->group_bpc[MODE];
  if (per_cu <= 0) {
    int occ = 0;
    hipError_t oe = hipErrorInvalidValue;
#define MK_OCC_G(GG) case GG: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spmv_group_kernel<GG, MODE, U, NT>, kGThreads, lds); break;
    switch (G) { MK_OCC_G(4) MK_OCC_G(8) MK_OCC_G(16) MK_OCC_G(32) MK_OCC_G(64) default: break; }
#undef MK_OCC_G
    per_cu = (oe == hipSuccess && occ > 0) ? std::min(occ, 8) : 4;
    A->group_bpc[MODE] = per_cu;
  }
  int64_t blocks = std::min<int64_t>(((int64_t)nchunks + 7) / 8 * 8, (int64_t)c.num_cus * per_cu);
  blocks = std::max<int64_t>(8, blocks / 8 * 8);
  if (blocks > kSlabStride) blocks = kSlabStride;
  if (nblocks) *nblocks = (int)blocks;
  if (dry_launch()) return MK_OK;
#define MK_LAUNCH_GR(GG)                                                                                                       \
  case GG: {                                                                                                                   \
    auto kern = spmv_group_kernel<GG, MODE, U, NT>;                                                                            \
    static size_t lds_max = 48 * 1024;   /* dynamic LDS above 48 KB: one opt-in per kernel, only ever raised */                \
    if (lds > lds_max) {                                                                                                       \
      MK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      lds_max = lds;                                                                                                           \
    }                                                                                                                          \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kGThreads), lds, c.stream, A->nrows, A->ngroups, nchunks,          \
                       reinterpret_cast<const gi4_t *>(A->gdesc), reinterpret_cast<const gi4_t *>(A->cdesc), A->colc, A->val, \
                       x, y, alpha, beta, slab, done, cap, ccap);                                                               \
  } break;
  switch (G) {
    MK_LAUNCH_GR(4) MK_LAUNCH_GR(8) MK_LAUNCH_GR(16) MK_LAUNCH_GR(32) MK_LAUNCH_GR(64)
    default: return fail(MK_ERR_INVALID, "spmv: row-group lanes %d invalid", G);
  }
#undef MK_LAUNCH_GR
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// mode 0: y = alpha A x + beta y; mode 1: y = A x + partials of <y,x>, <x,x>
int launch_group(const mk_matrix *A, int mode, double alpha, const double *x, double beta, double *y, double *slab, const int *done, int *nblocks) {
  if (!A->grow || A->group_lanes < 4) return fail(MK_ERR_UNSUPPORTED, "spmv: the row-group kernel needs consecutive rows that share their column lists");
  const bool nt = nt_matrix((double)A->nnz * 8.0);
  const int cap = A->group_cap;
#define MK_GROUP_PICK(UU)                                                                                      \
  if (cap <= 1024 * UU - 32) {                                                                                 \
    if (mode == 0) return nt ? launch_group_u<0, UU, true>(A, alpha, x, beta, y, slab, done, nblocks)          \
                             : launch_group_u<0, UU, false>(A, alpha, x, beta, y, slab, done, nblocks);        \
    return nt ? launch_group_u<1, UU, true>(A, alpha, x, beta, y, slab, done, nblocks)                         \
              : launch_group_u<1, UU, false>(A, alpha, x, beta, y, slab, done, nblocks);                       \
  }
  MK_GROUP_PICK(1)
  MK_GROUP_PICK(2)
  MK_GROUP_PICK(4)
  MK_GROUP_PICK(8)
#undef MK_GROUP_PICK
  return fail(MK_ERR_UNSUPPORTED, "spmv: row-group chunk of %d entries exceeds the LDS stage", cap);
}

}  // namespace mk
