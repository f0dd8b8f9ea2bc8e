// dist.hip -- row-block distributed SpMV and CG: one rank per GPU, RCCL over xGMI.
//
// Reference counterpart: solvers.c:214-281 partitions the row-sorted COO into contiguous
// nnz-balanced row blocks, one per MPI rank, and HYPRE's ParCSR format then keeps, per rank,
// a "diag" block (columns it owns), an "offd" block (columns owned by others, renumbered
// into a compact ghost range) and a communication package; its matvec posts the halo
// exchange, multiplies the diag block, waits, multiplies the offd block, and every inner
// product ends in MPI_Allreduce of one double (all inside HYPRE, not in /root/reference).
// Here:
//   - the reference broadcasts the WHOLE COO to every rank first (solvers.c:263-265); this
//     library never does: every rank builds or uploads only its own rows;
//   - halo: pack kernel + grouped ncclSend/ncclRecv on a communication stream, received
//     straight into the ghost tail of the vector, overlapped with the diag-block SpMV;
//   - inner products: the fused kernels leave per-block partials, one small kernel folds
//     them, ONE ncclAllReduce carries both dots of a step (2 and 3 doubles per iteration);
//   - two communicators (point-to-point and collective) so that the halo of the next step
//     may run beside an all-reduce without two kernels of one communicator in flight.
// RCCL is resolved with dlopen at first use, so single-GPU users need no RCCL at all.
//
// For tests on one GPU the same code drives several logical ranks inside one process
// ("local world"): the exchange becomes device-to-device copies and the all-reduce a small
// kernel, everything else (partition, ghost maps, diag/offd split, solver) is identical.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>

#include <cfloat>

#include "krylov_kernels.h"
#include "dist_internal.h"

namespace mk {

// --------------------------------------------------------------------------- RCCL binding
struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};

static Rccl g_rccl;

static int load_rccl() {
  if (g_rccl.handle) return MK_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail(MK_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
#define MK_SYM(field, name)                                                         \
  *(void **)(&g_rccl.field) = dlsym(h, name);                                       \
  if (!g_rccl.field) return fail(MK_ERR_COMM, "librccl lacks symbol %s", name);
  MK_SYM(GetUniqueId, "ncclGetUniqueId")
  MK_SYM(CommInitRank, "ncclCommInitRank")
  MK_SYM(CommDestroy, "ncclCommDestroy")
  MK_SYM(GetErrorString, "ncclGetErrorString")
  MK_SYM(AllReduce, "ncclAllReduce")
  MK_SYM(AllGather, "ncclAllGather")
  MK_SYM(Send, "ncclSend")
  MK_SYM(Recv, "ncclRecv")
  MK_SYM(GroupStart, "ncclGroupStart")
  MK_SYM(GroupEnd, "ncclGroupEnd")
#undef MK_SYM
  g_rccl.handle = h;
  return MK_OK;
}

#define MK_NCCL(call)                                                                                 \
  do {                                                                                                \
    ncclResult_t r_ = (call);                                                                         \
    if (r_ != ncclSuccess)                                                                            \
      return mk::fail(MK_ERR_COMM, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
  } while (0)

struct LocalWorld;

}  // namespace mk

namespace mk {

struct LocalWorld {
  int nranks = 0;
  std::vector<mk_comm *> comms;
  std::vector<mk_dist_matrix *> mats;   // registered by rank
  // all-reduce emulation: device tables of nranks pointers each.  A solver all-reduces the same few buffer sets over and over,
  // so a table is uploaded once and found again by its contents (no copy, no host wait per reduction: with one the
  // local world stalled the GPU twice per CG iteration, which no RCCL run does)
  static constexpr int kPtrTabs = 4;
  double **d_ptrs = nullptr;            // kPtrTabs x nranks
  std::vector<double *> h_ptrs[kPtrTabs];
  int next_tab = 0;
  // the local world's stand-in for the communication stream: halo copies and the off-rank products of the fused CG step
  // run here, beside the diag sweeps on the main stream, exactly as they do on the RCCL communication stream
  hipStream_t side = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
};

// ------------------------------------------------------------------------------ kernels
__global__ void flag_ghost_cols_kernel(int64_t nnz, const int *__restrict__ col, int rb, int re, uint32_t *__restrict__ flags) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
    const int c = col[k];
    if (c < rb || c >= re) flags[c] = 1u;
  }
}

__global__ void compact_ghosts_kernel(int64_t n_global, const uint32_t *__restrict__ scanned, uint32_t total, int *__restrict__ ghost) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_global; c += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t here = scanned[c];
    const uint32_t next = (c + 1 < n_global) ? scanned[c + 1] : total;
    if (next != here) ghost[here] = (int)c;
  }
}

__global__ void split_count_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col, int rb, int re,
                                   uint32_t *__restrict__ cnt_d, uint32_t *__restrict__ cnt_o, uint32_t *__restrict__ has_o) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    uint32_t d = 0, o = 0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; k++) {
      const int c = col[k];
      if (c >= rb && c < re) d++; else o++;
    }
    cnt_d[r] = d;
    cnt_o[r] = o;
    has_o[r] = o ? 1u : 0u;
  }
}

__global__ void split_fill_kernel(int64_t nrows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                  const double *__restrict__ val, int rb, int re, const uint32_t *__restrict__ ghost_pos,
                                  const int *__restrict__ d_rowptr, int *__restrict__ d_col, double *__restrict__ d_val,
                                  const uint32_t *__restrict__ o_off, const uint32_t *__restrict__ o_rowpos,
                                  const uint32_t *__restrict__ has_o, int *__restrict__ o_rowptr, int *__restrict__ o_rows,
                                  int *__restrict__ o_col, double *__restrict__ o_val) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    int dp = d_rowptr[r];
    uint32_t op = o_off[r];
    if (has_o[r]) {
      o_rowptr[o_rowpos[r]] = (int)op;
      o_rows[o_rowpos[r]] = (int)r;
    }
    for (int k = rowptr[r]; k < rowptr[r + 1]; k++) {
      const int c = col[k];
      if (c >= rb && c < re) {
        d_col[dp] = c - rb;
        d_val[dp++] = val[k];
      } else {
        o_col[op] = (int)ghost_pos[c];
        o_val[op++] = val[k];
      }
    }
  }
}

__global__ void set_int_kernel(int *p, int v) { *p = v; }
__global__ void add_int_kernel(int *p, int64_t n, int delta) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] += delta;
}

__global__ void pack_kernel(int n, const int *__restrict__ idx, const double *__restrict__ v, double *__restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = v[idx[i]];
}

// What the neighbours need of p' = z + beta p BEFORE the sweep that forms p' has run (K3 folded into that sweep):
// the rows of the send list, with beta worked out from the all-reduced <r,z>, <r,r>, <x,x> exactly as K3's
// prologue does (same expression as the sweep -> the ghosts hold the owners' bits).  Reads only state fields that
// the sweep's prologue, which may run at the same time on the other stream, does not write.
struct HaloFuse {
  const double *r = nullptr, *p_old = nullptr, *buf3 = nullptr;
  const PcgState *st = nullptr;
  double zscale = 1.0;
  int flags = 0;   // bit 0 two_norm, bit 1 rel_change
};

__global__ void pack_new_p_kernel(int n, const int *__restrict__ idx, HaloFuse f, double *__restrict__ out) {
  const PcgState *st = f.st;
  if (st->done != 0) return;
  const double gamma_new = f.buf3[0];
  const double i_prod = (f.flags & 1) ? f.buf3[1] : f.buf3[0];
  const double xx = f.buf3[2];
  bool conv = (i_prod / st->bi_prod < st->eps);
  if (conv && (f.flags & 2) && i_prod > 0.0) {
    const double ratio = st->alpha * st->alpha * st->pp / xx;
    if (ratio >= st->eps) conv = false;
  }
  const bool tiny_gamma = (gamma_new < DBL_MIN) && (-gamma_new < DBL_MIN);
  if (conv || tiny_gamma) return;
  const double beta = gamma_new / st->gamma_old;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int row = idx[i];
    double z = f.r[row];
    z *= f.zscale;
    out[i] = beta * f.p_old[row] + z;
  }
}

// y[rows[i]] += alpha * sum_k val[k] * xghost[col[k]]  over the offd block; with DOT the
// per-block partial of sum_i (contribution_i * xlocal[rows[i]]) goes to slab component 0
// (component 1 gets 0) so that <A p, p> can be completed after the diag pass.
// COMPACT: y is a vector over the compressed list of rows (t[i]) instead of over the local rows (y[rows[i]])
// FUSE (fused CG step from its second iteration on, the kernel runs BESIDE the sweep that forms p' = z + beta p): the
// local operand of the dot is not read from xlocal (the sweep is writing it) but formed from r and the old p with
// K3's own tests and the sweep's own expression, as pack_new_p_kernel does for the send rows.
template <int G, bool DOT, bool SET, bool COMPACT = false, bool FUSE = false>
__global__ __launch_bounds__(256) void offd_spmv_kernel(int nrows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                        const double *__restrict__ val, const int *__restrict__ rows,
                                                        const double *__restrict__ xghost, const double *__restrict__ xlocal,
                                                        double *__restrict__ y, double alpha, double *__restrict__ slab,
                                                        const int *__restrict__ done, HaloFuse f) {
  if (done != nullptr && *done != 0) return;
  double beta = 0.0;
  if (FUSE) {
    const PcgState *st = f.st;
    const double gamma_new = f.buf3[0];
    const double i_prod = (f.flags & 1) ? f.buf3[1] : f.buf3[0];
    const double xx = f.buf3[2];
    bool conv = (i_prod / st->bi_prod < st->eps);
    if (conv && (f.flags & 2) && i_prod > 0.0) {
      const double ratio = st->alpha * st->alpha * st->pp / xx;
      if (ratio >= st->eps) conv = false;
    }
    const bool tiny_gamma = (gamma_new < DBL_MIN) && (-gamma_new < DBL_MIN);
    if (conv || tiny_gamma) return;
    beta = gamma_new / st->gamma_old;
  }
  constexpr int ROWS = 256 / G;
  const int g = threadIdx.x % G, rl = threadIdx.x / G;
  double acc = 0.0;
  for (int base = blockIdx.x * ROWS; base < nrows; base += gridDim.x * ROWS) {
    const int i = base + rl;
    double sum = 0.0;
    if (i < nrows)
      for (int k = rowptr[i] + g; k < rowptr[i + 1]; k += G) sum += val[k] * xghost[col[k]];
    sum = group_sum<G>(sum);
    if (g == 0 && i < nrows) {
      const int r = rows[i];
      const double add = alpha * sum;
      if (COMPACT) y[i] = add;
      else if (SET) y[r] = add;
      else y[r] += add;
      if (DOT) {
        double xl;
        if (FUSE) {
          double z = f.r[r];
          z *= f.zscale;
          xl = beta * f.p_old[r] + z;
        } else {
          xl = xlocal[r];
        }
        acc += add * xl;
      }
    }
  }
  if (DOT) {
    __shared__ double smem[256 / 64];
    double v[1] = {acc};
    block_sum<1, 256>(v, smem);
    if (threadIdx.x == 0) {
      slab[0 * kSlabStride + blockIdx.x] = v[0];
      slab[1 * kSlabStride + blockIdx.x] = 0.0;
    }
  }
}

__global__ void local_allreduce_kernel(double **bufs, int nranks, int count) {
  const int k = threadIdx.x;
  if (k >= count) return;
  double s = 0.0;
  for (int r = 0; r < nranks; r++) s += bufs[r][k];
  for (int r = 0; r < nranks; r++) bufs[r][k] = s;
}

static int grid1(int64_t n, int cap = 4096) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, cap)); }

static void release_dist(mk_dist_matrix *D) {
  auto freem = [](mk_matrix &M) {
    dev_free(M.rowptr); dev_free(M.col); dev_free(M.val); dev_free(M.dinv); dev_free(M.chunk_row); dev_free(M.long_rows); dev_free(M.carry); dev_free(M.pat); dev_free(M.ptab); dev_free(M.stab); dev_free(M.colc); dev_free(M.cptr); dev_free(M.rowpair); dev_free(M.grow); dev_free(M.gcol); dev_free(M.gdesc); dev_free(M.cdesc); dev_free(M.vtab); dev_free(M.otab8); dev_free(M.vtab8); dev_free(M.pmask); dev_free(M.etab); dev_free(M.sweep_perm); delete M.sweep;
    dev_free(M.col2); dev_free(M.xg); dev_free(M.xorder);
    M = mk_matrix();
  };
  freem(D->diag);
  freem(D->offd);
  dev_free(D->offd_rows); dev_free(D->ghost_global); dev_free(D->send_idx); dev_free(D->sendbuf);
  if (D->diagT) mk_matrix_destroy(D->diagT);
  if (D->offdT) mk_matrix_destroy(D->offdT);
  D->diagT = D->offdT = nullptr;
  dev_free(D->ghost_out); dev_free(D->recvbuf);
  D->ghost_out = D->recvbuf = nullptr;
}

// split a local row block with GLOBAL column ids into diag + offd and list its ghosts
static int split_block(mk_matrix *A, mk_dist_matrix *D) {
  Context &c = ctx();
  const int64_t nl = D->n_local, ng = D->n_global;
  const int rb = (int)D->row_begin, re = (int)D->row_end;
  uint32_t *flags = nullptr, *cnt_d = nullptr, *cnt_o = nullptr, *has_o = nullptr, *d_tot = nullptr;
  MK_TRY(dev_alloc(&flags, (size_t)ng));
  MK_TRY(dev_alloc(&cnt_d, (size_t)nl));
  MK_TRY(dev_alloc(&cnt_o, (size_t)nl));
  MK_TRY(dev_alloc(&has_o, (size_t)nl));
  MK_TRY(dev_alloc(&d_tot, 4));
  MK_HIP(hipMemsetAsync(flags, 0, sizeof(uint32_t) * (size_t)ng, c.stream));
  if (A->nnz) hipLaunchKernelGGL(flag_ghost_cols_kernel, dim3(grid1(A->nnz, 8192)), dim3(256), 0, c.stream, A->nnz, A->col, rb, re, flags);
  MK_TRY(exclusive_scan_u32(flags, flags, (size_t)ng, d_tot + 0));
  if (nl) hipLaunchKernelGGL(split_count_kernel, dim3(grid1(nl)), dim3(256), 0, c.stream, nl, A->rowptr, A->col, rb, re, cnt_d, cnt_o, has_o);
  uint32_t h_tot[4] = {0, 0, 0, 0};
  MK_TRY(dev_alloc(&D->diag.rowptr, (size_t)nl + 1));
  MK_TRY(exclusive_scan_u32(cnt_d, reinterpret_cast<uint32_t *>(D->diag.rowptr), (size_t)nl, d_tot + 1));
  MK_TRY(exclusive_scan_u32(cnt_o, cnt_o, (size_t)nl, d_tot + 2));        // cnt_o -> offset of the row's offd entries
  uint32_t *o_rowpos = cnt_d;                                             // reuse: position among offd rows
  MK_TRY(exclusive_scan_u32(has_o, o_rowpos, (size_t)nl, d_tot + 3));
  MK_HIP(hipMemcpyAsync(h_tot, d_tot, sizeof(h_tot), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  D->n_ghost = h_tot[0];
  D->n_offd_rows = (int)h_tot[3];
  D->diag.nrows = nl; D->diag.ncols = nl; D->diag.nnz = h_tot[1];
  D->offd.nrows = D->n_offd_rows; D->offd.ncols = D->n_ghost; D->offd.nnz = h_tot[2];
  if ((int64_t)h_tot[1] + h_tot[2] != A->nnz) return fail(MK_ERR_INVALID, "dist split: entry counts do not add up");
  MK_TRY(dev_alloc(&D->diag.col, (size_t)D->diag.nnz));
  MK_TRY(dev_alloc(&D->diag.val, (size_t)D->diag.nnz));
  MK_TRY(dev_alloc(&D->offd.rowptr, (size_t)D->n_offd_rows + 1));
  MK_TRY(dev_alloc(&D->offd.col, (size_t)D->offd.nnz));
  MK_TRY(dev_alloc(&D->offd.val, (size_t)D->offd.nnz));
  MK_TRY(dev_alloc(&D->offd_rows, (size_t)D->n_offd_rows));
  MK_TRY(dev_alloc(&D->ghost_global, (size_t)D->n_ghost));
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, c.stream, D->diag.rowptr + nl, (int)D->diag.nnz);
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, c.stream, D->offd.rowptr + D->n_offd_rows, (int)D->offd.nnz);
  if (D->n_ghost) hipLaunchKernelGGL(compact_ghosts_kernel, dim3(grid1(ng, 8192)), dim3(256), 0, c.stream, ng, flags, h_tot[0], D->ghost_global);
  if (nl) hipLaunchKernelGGL(split_fill_kernel, dim3(grid1(nl)), dim3(256), 0, c.stream, nl, A->rowptr, A->col, A->val, rb, re, flags,
                             D->diag.rowptr, D->diag.col, D->diag.val, cnt_o, o_rowpos, has_o, D->offd.rowptr, D->offd_rows,
                             D->offd.col, D->offd.val);
  D->h_ghost.resize((size_t)D->n_ghost);
  if (D->n_ghost) MK_HIP(hipMemcpyAsync(D->h_ghost.data(), D->ghost_global, sizeof(int) * D->n_ghost, hipMemcpyDeviceToHost, c.stream));
  std::vector<int> h_rows((size_t)D->n_offd_rows);
  if (D->n_offd_rows) MK_HIP(hipMemcpyAsync(h_rows.data(), D->offd_rows, sizeof(int) * D->n_offd_rows, hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  MK_HIP(hipGetLastError());
  {   // longest run of rows between two consecutive rows that touch ghosts (the list is ascending)
    int prev = -1;
    D->gap0 = 0; D->gap1 = 0;
    for (int i = 0; i <= D->n_offd_rows; i++) {
      const int next = (i < D->n_offd_rows) ? h_rows[i] : (int)nl;
      if (next - (prev + 1) > D->gap1 - D->gap0) { D->gap0 = prev + 1; D->gap1 = next; }
      prev = next;
    }
  }
  dev_free(flags); dev_free(cnt_d); dev_free(cnt_o); dev_free(has_o); dev_free(d_tot);
  // receive plan: ghosts are sorted by global id, owners are contiguous row blocks (host logic shared with the CPU tests)
  D->recv_cnt.assign(D->nranks, 0);
  D->recv_off.assign(D->nranks + 1, 0);
  MK_TRY(mk_halo_plan_recv(D->nranks, D->rank, D->row_start.data(), D->h_ghost.data(), D->n_ghost, D->recv_cnt.data(), D->recv_off.data()));
  const double mean_o = D->n_offd_rows ? (double)D->offd.nnz / D->n_offd_rows : 0.0;
  // lanes per row: up to 8 entries per lane (a 27-point slab face has 9 per row: 2 lanes; 16 lanes left most of the
  // chip idle on the 0.5 M short rows: 37 us against 12 us)
  int g = 1;
  while (g < 64 && 8 * g < mean_o) g *= 2;
  D->offd_lanes = g;
  return MK_OK;
}

// recv_all[q * nranks + r] = what rank q receives from rank r
static int finish_send_plan(mk_dist_matrix *D, const std::vector<int> &recv_all) {
  D->send_cnt.assign(D->nranks, 0);
  D->send_off.assign(D->nranks + 1, 0);
  MK_TRY(mk_halo_plan_send(D->nranks, D->rank, recv_all.data(), D->send_cnt.data(), D->send_off.data()));
  D->n_send = D->send_off[D->nranks];
  MK_TRY(dev_alloc(&D->send_idx, (size_t)D->n_send));
  MK_TRY(dev_alloc(&D->sendbuf, (size_t)D->n_send));
  return MK_OK;
}

// build the send side once every rank knows what it needs (collective)
static int build_plan_rccl(mk_dist_matrix *D) {
  Context &c = ctx();
  mk_comm *cm = D->comm;
  const int P = D->nranks;
  int *d_cnt = nullptr, *d_all = nullptr;
  MK_TRY(dev_alloc(&d_cnt, (size_t)P));
  MK_TRY(dev_alloc(&d_all, (size_t)P * P));
  MK_HIP(hipMemcpyAsync(d_cnt, D->recv_cnt.data(), sizeof(int) * P, hipMemcpyHostToDevice, c.stream));
  MK_NCCL(g_rccl.AllGather(d_cnt, d_all, (size_t)P, ncclInt32, cm->coll, c.stream));
  std::vector<int> all((size_t)P * P);
  MK_HIP(hipMemcpyAsync(all.data(), d_all, sizeof(int) * P * P, hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  dev_free(d_cnt); dev_free(d_all);
  MK_TRY(finish_send_plan(D, all));   // what every q needs from me
  MK_NCCL(g_rccl.GroupStart());
  for (int q = 0; q < P; q++) {
    if (D->recv_cnt[q]) MK_NCCL(g_rccl.Send(D->ghost_global + D->recv_off[q], (size_t)D->recv_cnt[q], ncclInt32, q, cm->p2p, c.stream));
    if (D->send_cnt[q]) MK_NCCL(g_rccl.Recv(D->send_idx + D->send_off[q], (size_t)D->send_cnt[q], ncclInt32, q, cm->p2p, c.stream));
  }
  MK_NCCL(g_rccl.GroupEnd());
  if (D->n_send) hipLaunchKernelGGL(add_int_kernel, dim3(grid1(D->n_send)), dim3(256), 0, c.stream, D->send_idx, (int64_t)D->n_send, -(int)D->row_begin);
  MK_HIP(hipStreamSynchronize(c.stream));
  D->plan_ready = true;
  return MK_OK;
}

static int build_plan_local(LocalWorld *W) {
  Context &c = ctx();
  const int P = W->nranks;
  std::vector<int> all((size_t)P * P);   // the table an all-gather of recv_cnt gives over RCCL
  for (int q = 0; q < P; q++) std::copy(W->mats[q]->recv_cnt.begin(), W->mats[q]->recv_cnt.end(), all.begin() + (size_t)q * P);
  for (int r = 0; r < P; r++) {
    mk_dist_matrix *D = W->mats[r];
    MK_TRY(finish_send_plan(D, all));
    for (int q = 0; q < P; q++)
      if (D->send_cnt[q])
        MK_HIP(hipMemcpyAsync(D->send_idx + D->send_off[q], W->mats[q]->ghost_global + W->mats[q]->recv_off[r],
                              sizeof(int) * D->send_cnt[q], hipMemcpyDeviceToDevice, c.stream));
    if (D->n_send) hipLaunchKernelGGL(add_int_kernel, dim3(grid1(D->n_send)), dim3(256), 0, c.stream, D->send_idx, (int64_t)D->n_send, -(int)D->row_begin);
    D->plan_ready = true;
  }
  MK_HIP(hipStreamSynchronize(c.stream));
  return MK_OK;
}

// ------------------------------------------------------------------- communication phases
// v[r] has n_local owned entries followed by n_ghost ghost entries
static int halo_exchange(int R, mk_dist_matrix **A, double **v, bool overlap, const HaloFuse *fuse = nullptr) {
  Context &c = ctx();
  if (A[0]->comm == nullptr) return MK_OK;
  if (A[0]->comm->local) {
    LocalWorld *W = A[0]->comm->world;
    hipStream_t ls = overlap ? W->side : c.stream;
    if (overlap) {
      MK_HIP(hipEventRecord(W->ev_ready, c.stream));
      MK_HIP(hipStreamWaitEvent(ls, W->ev_ready, 0));
    }
    for (int r = 0; r < R; r++)
      if (A[r]->n_send) {
        if (fuse && fuse[r].st) hipLaunchKernelGGL(pack_new_p_kernel, dim3(grid1(A[r]->n_send, 1024)), dim3(256), 0, ls, A[r]->n_send, A[r]->send_idx, fuse[r], A[r]->sendbuf);
        else hipLaunchKernelGGL(pack_kernel, dim3(grid1(A[r]->n_send, 1024)), dim3(256), 0, ls, A[r]->n_send, A[r]->send_idx, v[r], A[r]->sendbuf);
      }
    for (int r = 0; r < R; r++)
      for (int q = 0; q < R; q++)
        if (A[r]->recv_cnt[q])
          MK_HIP(hipMemcpyAsync(v[r] + A[r]->n_local + A[r]->recv_off[q], A[q]->sendbuf + A[q]->send_off[r],
                                sizeof(double) * A[r]->recv_cnt[q], hipMemcpyDeviceToDevice, ls));
    if (overlap) MK_HIP(hipEventRecord(W->ev_done, ls));
    return MK_OK;
  }
  mk_dist_matrix *D = A[0];
  mk_comm *cm = D->comm;
  hipStream_t s = overlap ? cm->comm_stream : c.stream;
  if (overlap) {
    MK_HIP(hipEventRecord(cm->ev_ready, c.stream));
    MK_HIP(hipStreamWaitEvent(s, cm->ev_ready, 0));
  }
  if (D->n_send) {
    if (fuse && fuse[0].st) hipLaunchKernelGGL(pack_new_p_kernel, dim3(grid1(D->n_send, 1024)), dim3(256), 0, s, D->n_send, D->send_idx, fuse[0], D->sendbuf);
    else hipLaunchKernelGGL(pack_kernel, dim3(grid1(D->n_send, 1024)), dim3(256), 0, s, D->n_send, D->send_idx, v[0], D->sendbuf);
  }
  MK_NCCL(g_rccl.GroupStart());
  for (int q = 0; q < D->nranks; q++) {
    if (D->send_cnt[q]) MK_NCCL(g_rccl.Send(D->sendbuf + D->send_off[q], (size_t)D->send_cnt[q], ncclDouble, q, cm->p2p, s));
    if (D->recv_cnt[q]) MK_NCCL(g_rccl.Recv(v[0] + D->n_local + D->recv_off[q], (size_t)D->recv_cnt[q], ncclDouble, q, cm->p2p, s));
  }
  MK_NCCL(g_rccl.GroupEnd());
  if (overlap) MK_HIP(hipEventRecord(cm->ev_halo, s));
  return MK_OK;
}

static int halo_wait(mk_dist_matrix *D, bool overlap) {
  if (D->comm == nullptr || !overlap) return MK_OK;
  if (D->comm->local) MK_HIP(hipStreamWaitEvent(ctx().stream, D->comm->world->ev_done, 0));
  else MK_HIP(hipStreamWaitEvent(ctx().stream, D->comm->ev_halo, 0));
  return MK_OK;
}

// the stream the halo of this matrix arrives on, for work that is to run there behind the exchange; halo_mark() moves the
// event halo_wait() waits for behind that work
static hipStream_t halo_stream(const mk_dist_matrix *D) {
  if (D->comm == nullptr) return ctx().stream;
  return D->comm->local ? D->comm->world->side : D->comm->comm_stream;
}
static int halo_mark(mk_dist_matrix *D) {
  if (D->comm == nullptr) return MK_OK;
  if (D->comm->local) MK_HIP(hipEventRecord(D->comm->world->ev_done, D->comm->world->side));
  else MK_HIP(hipEventRecord(D->comm->ev_halo, D->comm->comm_stream));
  return MK_OK;
}

int dist_all_reduce(int R, mk_dist_matrix **A, double **bufs, int count) {
  Context &c = ctx();
  if (A[0]->comm == nullptr) return MK_OK;
  if (A[0]->comm->local) {
    LocalWorld *W = A[0]->comm->world;
    int tab = -1;
    for (int t = 0; t < LocalWorld::kPtrTabs && tab < 0; t++)
      if ((int)W->h_ptrs[t].size() == R && std::equal(bufs, bufs + R, W->h_ptrs[t].begin())) tab = t;
    if (tab < 0) {   // a new set of buffers: take the next table (whoever used it last has long finished: drain to be sure)
      tab = W->next_tab;
      W->next_tab = (W->next_tab + 1) % LocalWorld::kPtrTabs;
      MK_HIP(hipStreamSynchronize(c.stream));
      W->h_ptrs[tab].assign(bufs, bufs + R);
      MK_HIP(hipMemcpyAsync(W->d_ptrs + (size_t)tab * R, W->h_ptrs[tab].data(), sizeof(double *) * R, hipMemcpyHostToDevice, c.stream));
      MK_HIP(hipStreamSynchronize(c.stream));
    }
    hipLaunchKernelGGL(local_allreduce_kernel, dim3(1), dim3(64), 0, c.stream, W->d_ptrs + (size_t)tab * R, R, count);
    return MK_OK;
  }
  MK_NCCL(g_rccl.AllReduce(bufs[0], bufs[0], (size_t)count, ncclDouble, ncclSum, A[0]->comm->coll, c.stream));
  return MK_OK;
}

// max_blocks: room left in the slab the partials go to (they sit behind the diag kernel's partials)
template <bool DOT, bool SET = false>
static int launch_offd(const mk_dist_matrix *D, const double *xext, double *y, double alpha, double *slab, const int *done, int *nblocks,
                       int max_blocks = 1024, hipStream_t stream = nullptr) {
  Context &c = ctx();
  if (stream == nullptr) stream = c.stream;
  const int G = D->offd_lanes;
  const int rows_per_block = 256 / G;
  if (max_blocks < 1) return fail(MK_ERR_UNSUPPORTED, "row-block CG: the diag block's %d reduction partials leave no room for the off-rank product's", kSlabStride - max_blocks);
  int blocks = std::max(1, std::min((D->n_offd_rows + rows_per_block - 1) / rows_per_block, std::min(1024, max_blocks)));
  if (nblocks) *nblocks = blocks;
  if (D->n_offd_rows == 0 && !DOT) return MK_OK;
#define MK_OFFD(GG) case GG: hipLaunchKernelGGL((offd_spmv_kernel<GG, DOT, SET>), dim3(blocks), dim3(256), 0, stream, D->n_offd_rows, \
    D->offd.rowptr, D->offd.col, D->offd.val, D->offd_rows, xext + D->n_local, xext, y, alpha, slab, done, HaloFuse()); break;
  switch (G) { MK_OFFD(1) MK_OFFD(2) MK_OFFD(4) MK_OFFD(8) MK_OFFD(16) MK_OFFD(32) MK_OFFD(64) default: return fail(MK_ERR_INVALID, "offd lanes"); }
#undef MK_OFFD
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// t[i] = (offd x_ghost)[i] over the rows that touch ghosts, partial of <t, x_local> (fused row-block CG step).
// stream: where the ghosts arrive (communication / side stream); fuse: the local operand is formed from r and the old p
static int launch_offd_compact(const mk_dist_matrix *D, const double *xext, double *t, double *slab, const int *done, int *nblocks,
                               int max_blocks, hipStream_t stream, const HaloFuse *fuse) {
  const int G = D->offd_lanes;
  const int rows_per_block = 256 / G;
  if (max_blocks < 1) return fail(MK_ERR_UNSUPPORTED, "row-block CG: the diag block's %d reduction partials leave no room for the off-rank product's", kSlabStride - max_blocks);
  int blocks = std::max(1, std::min((D->n_offd_rows + rows_per_block - 1) / rows_per_block, std::min(1024, max_blocks)));
  if (nblocks) *nblocks = blocks;
#define MK_OFFD(GG) case GG: \
    if (fuse && fuse->st) hipLaunchKernelGGL((offd_spmv_kernel<GG, true, true, true, true>), dim3(blocks), dim3(256), 0, stream, D->n_offd_rows, \
      D->offd.rowptr, D->offd.col, D->offd.val, D->offd_rows, xext + D->n_local, xext, t, 1.0, slab, done, *fuse); \
    else hipLaunchKernelGGL((offd_spmv_kernel<GG, true, true, true, false>), dim3(blocks), dim3(256), 0, stream, D->n_offd_rows, \
      D->offd.rowptr, D->offd.col, D->offd.val, D->offd_rows, xext + D->n_local, xext, t, 1.0, slab, done, HaloFuse()); \
    break;
  switch (G) { MK_OFFD(1) MK_OFFD(2) MK_OFFD(4) MK_OFFD(8) MK_OFFD(16) MK_OFFD(32) MK_OFFD(64) default: return fail(MK_ERR_INVALID, "offd lanes"); }
#undef MK_OFFD
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// r[rows[i]] -= alpha t[i] with alpha = gamma / <A p, p> (all-reduced pair in buf); the sweep that follows does
// the diag part of the same update and owns the breakdown bookkeeping, this kernel only mirrors its tests
static __global__ void offd_resid_kernel(int nrows, const int *__restrict__ rows, const double *__restrict__ t, double *__restrict__ r,
                                         const double *__restrict__ buf, const PcgState *st) {
  if (st->done != 0) return;
  const double sdotp = buf[0];
  if (sdotp == 0.0) return;
  const double alpha = st->gamma / sdotp;
  if (!(alpha > DBL_MIN)) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += gridDim.x * blockDim.x) r[rows[i]] -= alpha * t[i];
}

// ---- single-reduction step (opt-in, mk_pcg_set_single_reduction): small kernels around spmv_sweep_kernel MODE 6
// out[0] = sum of component 0 of slab_a (<A r, r>: diag sweep + off-rank product); out[1..4] = the four components the last
// MODE 6 sweep left in slab_b (<r,z>, <r,r>, <x,x>, <p,p>; nb = 0 before the first step: zeros)
static __global__ __launch_bounds__(kVecThreads) void single_fold_kernel(const double *slab_a, int na, const double *slab_b, int nb, double *out) {
  __shared__ double smem[4 * (kVecThreads / 64)];
  double a[1], b[4];
  reduce_slab<1, kVecThreads>(slab_a, na, kSlabStride, a, smem);
  reduce_slab<4, kVecThreads>(slab_b, nb, kSlabStride, b, smem);
  if (threadIdx.x == 0) {
    out[0] = a[0];
    for (int k = 0; k < 4; k++) out[1 + k] = b[k];
  }
}
// tpf[rows[i]] = zscale t[i] + beta tpf[rows[i]]: the off-rank part of A p' follows p' = z + beta p (t = off-rank part of
// A r; beta as MODE 6's prologue forms it from the same all-reduced values)
static __global__ void single_offd_kernel(int nrows, const int *__restrict__ rows, const double *__restrict__ t, double *__restrict__ tpf,
                                          const double *__restrict__ buf, const PcgState *st, double zscale, int first, int parity) {
  if (st->done != 0) return;
  const double beta = first ? 0.0 : buf[1] / st->sr_gamma[parity ^ 1];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += gridDim.x * blockDim.x) {
    const int r = rows[i];
    tpf[r] = zscale * t[i] + beta * tpf[r];
  }
}
// the convergence test the next step's prologue would have made (after the last step of a solve)
static __global__ void single_final_test_kernel(const double *__restrict__ buf, PcgState *st, int flags, double *__restrict__ norms, int parity) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || st->done != 0) return;
  const double gamma_new = buf[1], i_prod = (flags & 1) ? buf[2] : buf[1], al = st->sr_alpha[parity ^ 1];
  const double xx = (buf[3] != 0.0) ? buf[3] : st->xx_last;   // (a last step that left x alone handed on zeros)
  bool conv = (i_prod / st->bi_prod < st->eps);
  if (conv && (flags & 2) && i_prod > 0.0) {
    const double ratio = al * al * buf[4] / xx;
    if (ratio >= st->eps) conv = false;
  }
  st->gamma = gamma_new;
  st->i_prod = i_prod;
  if (norms) norms[st->iter] = sqrt(i_prod / st->bi_prod);
  if (conv) { st->converged = 1; st->done = 1; }
}

struct DistWork {
  double *r = nullptr, *p = nullptr, *s = nullptr, *slab1 = nullptr, *slab2 = nullptr, *buf = nullptr, *norms = nullptr;
  double *t = nullptr;   // fused step: off-rank part of A p over the rows that touch ghosts
  double *p2 = nullptr;  // fused step: the other p buffer (K3 folded into the first sweep writes p' there)
  double *tpf = nullptr; // single-reduction step: off-rank part of A p over the local rows (zero where a row touches no ghost)
  PcgState *state = nullptr;
  void release() {
    dev_free(t); dev_free(p2); dev_free(tpf);
    dev_free(r); dev_free(p); dev_free(s); dev_free(slab1); dev_free(slab2); dev_free(buf); dev_free(state); dev_free(norms);
  }
};

// ------------------------------------------------------------------- transposed product
static __global__ void map_cols_kernel(int64_t nnz, int *__restrict__ col, const int *__restrict__ map) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) col[k] = map[col[k]];
}

// y[rows[i]] += alpha v[i]; the rows one neighbour contributes to are distinct, so no two threads meet
static __global__ void scatter_add_kernel(int n, const int *__restrict__ rows, const double *__restrict__ v, double *__restrict__ y, double alpha,
                                          const int *__restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[rows[i]] += alpha * v[i];
}

static int build_dist_transpose(mk_dist_matrix *D) {
  Context &c = ctx();
  if (D->diagT) return MK_OK;
  MK_TRY(build_transpose(&D->diag, &D->diagT));
  MK_TRY(analyse_matrix(D->diagT));
  if (D->n_ghost > 0 && D->offd.nnz > 0) {
    MK_TRY(build_transpose(&D->offd, &D->offdT));   // n_ghost x n_offd_rows: columns are compressed row numbers
    hipLaunchKernelGGL(map_cols_kernel, dim3(grid1(D->offdT->nnz, 4096)), dim3(256), 0, c.stream, D->offdT->nnz, D->offdT->col, D->offd_rows);
    D->offdT->ncols = D->n_local;                   // offd_rows ascends: the mapped columns of a row stay sorted
    MK_TRY(analyse_matrix(D->offdT));
    MK_TRY(dev_alloc(&D->ghost_out, (size_t)D->n_ghost));
  }
  if (D->n_send > 0) MK_TRY(dev_alloc(&D->recvbuf, (size_t)D->n_send));
  MK_HIP(hipStreamSynchronize(c.stream));
  return MK_OK;
}

int dist_spmv_transpose(int R, mk_dist_matrix **A, double **x, double **y, double alpha, double beta, const int *const *skip) {
  Context &c = ctx();
  for (int r = 0; r < R; r++) MK_TRY(build_dist_transpose(A[r]));
  const bool comm = A[0]->comm != nullptr;
  // what my rows contribute to the others' rows
  for (int r = 0; r < R; r++)
    if (A[r]->offdT) MK_TRY(launch_spmv(A[r]->offdT, 1.0, x[r], 0.0, A[r]->ghost_out, skip ? skip[r] : nullptr));
  // ... goes back along the plan (send and receive sides swapped) on the communication stream, beside diag^T x
  if (comm && A[0]->comm->local) {
    LocalWorld *W = A[0]->comm->world;
    MK_HIP(hipEventRecord(W->ev_ready, c.stream));
    MK_HIP(hipStreamWaitEvent(W->side, W->ev_ready, 0));
    for (int r = 0; r < R; r++)
      for (int q = 0; q < R; q++)
        if (A[r]->send_cnt[q])
          MK_HIP(hipMemcpyAsync(A[r]->recvbuf + A[r]->send_off[q], A[q]->ghost_out + A[q]->recv_off[r], sizeof(double) * A[r]->send_cnt[q],
                                hipMemcpyDeviceToDevice, W->side));
    MK_HIP(hipEventRecord(W->ev_done, W->side));
  } else if (comm) {
    mk_dist_matrix *D = A[0];
    mk_comm *cm = D->comm;
    MK_HIP(hipEventRecord(cm->ev_ready, c.stream));
    MK_HIP(hipStreamWaitEvent(cm->comm_stream, cm->ev_ready, 0));
    MK_NCCL(g_rccl.GroupStart());
    for (int q = 0; q < D->nranks; q++) {
      if (D->recv_cnt[q]) MK_NCCL(g_rccl.Send(D->ghost_out + D->recv_off[q], (size_t)D->recv_cnt[q], ncclDouble, q, cm->p2p, cm->comm_stream));
      if (D->send_cnt[q]) MK_NCCL(g_rccl.Recv(D->recvbuf + D->send_off[q], (size_t)D->send_cnt[q], ncclDouble, q, cm->p2p, cm->comm_stream));
    }
    MK_NCCL(g_rccl.GroupEnd());
    MK_HIP(hipEventRecord(cm->ev_halo, cm->comm_stream));
  }
  for (int r = 0; r < R; r++) MK_TRY(launch_spmv(A[r]->diagT, alpha, x[r], beta, y[r], skip ? skip[r] : nullptr));
  for (int r = 0; r < R; r++) {
    MK_TRY(halo_wait(A[r], true));
    mk_dist_matrix *D = A[r];
    for (int q = 0; q < D->nranks; q++)
      if (D->plan_ready && D->send_cnt[q])
        hipLaunchKernelGGL(scatter_add_kernel, dim3(grid1(D->send_cnt[q], 1024)), dim3(256), 0, c.stream, D->send_cnt[q], D->send_idx + D->send_off[q],
                           D->recvbuf + D->send_off[q], y[r], alpha, skip ? skip[r] : nullptr);
  }
  MK_HIP(hipGetLastError());
  return MK_OK;
}

int dist_spmv(int R, mk_dist_matrix **A, double **x, double **y, double alpha, double beta, const int *const *skip) {
  MK_TRY(halo_exchange(R, A, x, true));
  for (int r = 0; r < R; r++) MK_TRY(launch_spmv(&A[r]->diag, alpha, x[r], beta, y[r], skip ? skip[r] : nullptr));
  for (int r = 0; r < R; r++) {
    MK_TRY(halo_wait(A[r], true));
    MK_TRY(launch_offd<false>(A[r], x[r], y[r], alpha, nullptr, skip ? skip[r] : nullptr, nullptr));
  }
  return MK_OK;
}

}  // namespace mk

using namespace mk;

extern "C" {

// ------------------------------------------------------------------------ communicators
int mk_comm_get_unique_id(void *id256) {
  if (!id256) return fail(MK_ERR_INVALID, "mk_comm_get_unique_id: null buffer");
  MK_TRY(load_rccl());
  ncclUniqueId a, b;
  MK_NCCL(g_rccl.GetUniqueId(&a));
  MK_NCCL(g_rccl.GetUniqueId(&b));
  memcpy(id256, &a, 128);
  memcpy((char *)id256 + 128, &b, 128);
  return MK_OK;
}

int mk_comm_create(const void *id256, int nranks, int rank, mk_comm **comm) {
  MK_REQUIRE_INIT();
  if (!id256 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(MK_ERR_INVALID, "mk_comm_create: bad argument");
  MK_TRY(load_rccl());
  mk_comm *cm = new mk_comm();
  cm->nranks = nranks; cm->rank = rank;
  ncclUniqueId a, b;
  memcpy(&a, id256, 128);
  memcpy(&b, (const char *)id256 + 128, 128);
  MK_NCCL(g_rccl.CommInitRank(&cm->coll, nranks, a, rank));
  MK_NCCL(g_rccl.CommInitRank(&cm->p2p, nranks, b, rank));
  MK_HIP(hipStreamCreateWithFlags(&cm->comm_stream, hipStreamNonBlocking));
  register_side_stream(cm->comm_stream);
  MK_HIP(hipEventCreateWithFlags(&cm->ev_ready, hipEventDisableTiming));
  MK_HIP(hipEventCreateWithFlags(&cm->ev_halo, hipEventDisableTiming));
  *comm = cm;
  return MK_OK;
}

int mk_comm_create_local(int nranks, mk_comm **comms) {
  MK_REQUIRE_INIT();
  if (!comms || nranks < 1 || nranks > 64) return fail(MK_ERR_INVALID, "mk_comm_create_local: bad argument");
  LocalWorld *W = new LocalWorld();
  W->nranks = nranks;
  W->mats.assign(nranks, nullptr);
  MK_TRY(dev_alloc(&W->d_ptrs, (size_t)nranks * LocalWorld::kPtrTabs));
  MK_HIP(hipStreamCreateWithFlags(&W->side, hipStreamNonBlocking));
  register_side_stream(W->side);
  MK_HIP(hipEventCreateWithFlags(&W->ev_ready, hipEventDisableTiming));
  MK_HIP(hipEventCreateWithFlags(&W->ev_done, hipEventDisableTiming));
  for (int r = 0; r < nranks; r++) {
    mk_comm *cm = new mk_comm();
    cm->nranks = nranks; cm->rank = r; cm->local = true; cm->world = W;
    W->comms.push_back(cm);
    comms[r] = cm;
  }
  return MK_OK;
}

__global__ void selftest_fill_kernel(double *a, int n, int rank) {
  if (threadIdx.x < n) a[threadIdx.x] = 1000.0 * rank + threadIdx.x;
}

// exercises every RCCL entry point the solver uses (grouped send/recv to itself, all-gather,
// all-reduce) on a few doubles and checks the values; collective
int mk_comm_self_test(mk_comm *cm) {
  MK_REQUIRE_INIT();
  if (!cm) return fail(MK_ERR_INVALID, "mk_comm_self_test: null communicator");
  if (cm->local) return MK_OK;
  Context &c = ctx();
  const int n = 8, P = cm->nranks;
  double *a = nullptr, *b = nullptr, *g = nullptr;
  MK_TRY(dev_alloc(&a, n));
  MK_TRY(dev_alloc(&b, n));
  MK_TRY(dev_alloc(&g, (size_t)n * P));
  hipLaunchKernelGGL(selftest_fill_kernel, dim3(1), dim3(64), 0, c.stream, a, n, cm->rank);
  MK_HIP(hipMemsetAsync(b, 0, sizeof(double) * n, c.stream));
  // ring: send to the next rank, receive from the previous one (to itself when alone)
  const int next = (cm->rank + 1) % P, prev = (cm->rank + P - 1) % P;
  MK_NCCL(g_rccl.GroupStart());
  MK_NCCL(g_rccl.Send(a, n, ncclDouble, next, cm->p2p, c.stream));
  MK_NCCL(g_rccl.Recv(b, n, ncclDouble, prev, cm->p2p, c.stream));
  MK_NCCL(g_rccl.GroupEnd());
  MK_NCCL(g_rccl.AllGather(a, g, n, ncclDouble, cm->coll, c.stream));
  MK_NCCL(g_rccl.AllReduce(a, a, n, ncclDouble, ncclSum, cm->coll, c.stream));
  std::vector<double> ha(n), hb(n), hg((size_t)n * P);
  MK_HIP(hipMemcpyAsync(ha.data(), a, sizeof(double) * n, hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipMemcpyAsync(hb.data(), b, sizeof(double) * n, hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipMemcpyAsync(hg.data(), g, sizeof(double) * n * P, hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  dev_free(a); dev_free(b); dev_free(g);
  for (int k = 0; k < n; k++) {
    const double want_sum = 1000.0 * (P * (P - 1) / 2) + (double)P * k;
    if (hb[k] != 1000.0 * prev + k) return fail(MK_ERR_COMM, "mk_comm_self_test: send/recv delivered %g, expected %g", hb[k], 1000.0 * prev + k);
    if (ha[k] != want_sum) return fail(MK_ERR_COMM, "mk_comm_self_test: all-reduce gave %g, expected %g", ha[k], want_sum);
    for (int q = 0; q < P; q++)
      if (hg[(size_t)q * n + k] != 1000.0 * q + k) return fail(MK_ERR_COMM, "mk_comm_self_test: all-gather slot wrong");
  }
  return MK_OK;
}

int mk_comm_all_reduce_host(mk_comm *cm, double *values, int count, int op) {
  MK_REQUIRE_INIT();
  if (!values || count < 1 || count > 64 || (op != 0 && op != 1)) return fail(MK_ERR_INVALID, "mk_comm_all_reduce_host: bad argument");
  if (!cm || cm->local || cm->nranks == 1) return MK_OK;
  Context &c = ctx();
  double *d = nullptr;
  MK_TRY(dev_alloc(&d, (size_t)count));
  hipError_t e = hipMemcpyAsync(d, values, sizeof(double) * count, hipMemcpyHostToDevice, c.stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) r = g_rccl.AllReduce(d, d, (size_t)count, ncclDouble, op == 1 ? ncclMax : ncclSum, cm->coll, c.stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(values, d, sizeof(double) * count, hipMemcpyDeviceToHost, c.stream);
  const hipError_t e2 = hipStreamSynchronize(c.stream);
  dev_free(d);
  if (r != ncclSuccess) return fail(MK_ERR_COMM, "mk_comm_all_reduce_host: %s", g_rccl.GetErrorString(r));
  if (e != hipSuccess || e2 != hipSuccess) return fail(MK_ERR_HIP, "mk_comm_all_reduce_host: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return MK_OK;
}

static void comm_release(mk_comm *cm) {
  if (cm->local) {
    LocalWorld *W = cm->world;
    W->comms[cm->rank] = nullptr;
    bool last = true;
    for (mk_comm *o : W->comms) if (o) last = false;
    if (last) {
      dev_free(W->d_ptrs);
      if (W->ev_ready) (void)hipEventDestroy(W->ev_ready);
      if (W->ev_done) (void)hipEventDestroy(W->ev_done);
      if (W->side) { unregister_side_stream(W->side); (void)hipStreamDestroy(W->side); }
      delete W;
    }
  } else {
    if (cm->coll) (void)g_rccl.CommDestroy(cm->coll);
    if (cm->p2p) (void)g_rccl.CommDestroy(cm->p2p);
    if (cm->ev_ready) (void)hipEventDestroy(cm->ev_ready);
    if (cm->ev_halo) (void)hipEventDestroy(cm->ev_halo);
    if (cm->comm_stream) { unregister_side_stream(cm->comm_stream); (void)hipStreamDestroy(cm->comm_stream); }
  }
  delete cm;
}

int mk_comm_destroy(mk_comm *cm) {
  if (!cm) return MK_OK;
  if (ctx().ready) (void)hipDeviceSynchronize();
  if (cm->users > 0) {   // matrices still refer to it
    cm->dead = true;
    return MK_OK;
  }
  comm_release(cm);
  return MK_OK;
}

// ---------------------------------------------------------------------- distributed matrix
int mk_dist_matrix_create(mk_matrix *A, int64_t n_global, const int64_t *row_start, mk_comm *comm, mk_dist_matrix **out) {
  MK_REQUIRE_INIT();
  if (!A || !row_start || !out) return fail(MK_ERR_INVALID, "mk_dist_matrix_create: null argument");
  const int P = comm ? comm->nranks : 1, me = comm ? comm->rank : 0;
  if (comm && !comm->local && !comm->coll) return fail(MK_ERR_INVALID, "mk_dist_matrix_create: communicator not initialised");
  MK_TRY(mk_halo_check_row_start(P, row_start, n_global));
  if (A->nrows != row_start[me + 1] - row_start[me] || A->ncols != n_global)
    return fail(MK_ERR_INVALID, "mk_dist_matrix_create: block is %lld x %lld, expected %lld x %lld", (long long)A->nrows,
                (long long)A->ncols, (long long)(row_start[me + 1] - row_start[me]), (long long)n_global);
  mk_dist_matrix *D = new mk_dist_matrix();
  D->comm = comm; D->nranks = P; D->rank = me;
  if (comm) comm->users++;
  D->n_global = n_global;
  D->row_start.assign(row_start, row_start + P + 1);
  D->row_begin = row_start[me]; D->row_end = row_start[me + 1];
  D->n_local = D->row_end - D->row_begin;
  int rc = split_block(A, D);
  if (rc == MK_OK) {
    if (comm == nullptr) {
      if (P != 1) rc = fail(MK_ERR_INVALID, "mk_dist_matrix_create: more than one rank needs a communicator");
      if (rc == MK_OK) rc = finish_send_plan(D, std::vector<int>(1, 0));
      D->plan_ready = true;
    } else if (comm->local) {
      LocalWorld *W = comm->world;
      W->mats[me] = D;
      bool all = true;
      for (mk_dist_matrix *m : W->mats) if (!m) all = false;
      if (all) rc = build_plan_local(W);
    } else {
      rc = build_plan_rccl(D);
    }
  }
  if (rc != MK_OK) {
    if (comm && comm->local && comm->world && comm->world->mats[me] == D) comm->world->mats[me] = nullptr;
    if (comm) comm->users--;
    release_dist(D);
    delete D;
    return rc;
  }
  mk_matrix_destroy(A);   // the block with global column ids is consumed
  *out = D;
  return MK_OK;
}

int mk_dist_matrix_destroy(mk_dist_matrix *D) {
  if (!D) return MK_OK;
  if (ctx().ready) (void)hipDeviceSynchronize();
  mk_comm *cm = D->comm;
  if (cm && cm->local && cm->world) cm->world->mats[D->rank] = nullptr;
  release_dist(D);
  delete D;
  if (cm && --cm->users == 0 && cm->dead) comm_release(cm);
  return MK_OK;
}

int mk_dist_matrix_get_info(const mk_dist_matrix *D, int64_t *n_local, int64_t *n_ghost, int64_t *n_offd_rows, int64_t *nnz_diag,
                            int64_t *nnz_offd, int *n_neighbors) {
  if (!D) return fail(MK_ERR_INVALID, "mk_dist_matrix_get_info: null matrix");
  if (n_local) *n_local = D->n_local;
  if (n_ghost) *n_ghost = D->n_ghost;
  if (n_offd_rows) *n_offd_rows = D->n_offd_rows;
  if (nnz_diag) *nnz_diag = D->diag.nnz;
  if (nnz_offd) *nnz_offd = D->offd.nnz;
  if (n_neighbors) {
    int k = 0;
    for (int q = 0; q < D->nranks; q++) if (D->recv_cnt[q] || (D->plan_ready && D->send_cnt[q])) k++;
    *n_neighbors = k;
  }
  return MK_OK;
}

int mk_dist_matrix_set_spmv_kernel(mk_dist_matrix *D, int kernel, int lanes_per_row) {
  if (!D) return fail(MK_ERR_INVALID, "mk_dist_matrix_set_spmv_kernel: null matrix");
  return mk_matrix_set_spmv_kernel(&D->diag, kernel, lanes_per_row);
}

int mk_dist_matrix_get_diag_info(mk_dist_matrix *D, mk_matrix_info *info) {
  if (!D) return fail(MK_ERR_INVALID, "mk_dist_matrix_get_diag_info: null matrix");
  return mk_matrix_get_info(&D->diag, info);
}

int mk_dist_vector_create(const mk_dist_matrix *D, mk_vector **v) {
  MK_REQUIRE_INIT();
  if (!D || !v) return fail(MK_ERR_INVALID, "mk_dist_vector_create: null argument");
  mk_vector *p = new mk_vector();
  p->n = D->n_local;
  p->cap = D->n_local + D->n_ghost;
  int rc = dev_alloc(&p->d, (size_t)p->cap);
  if (rc != MK_OK) { delete p; return rc; }
  *v = p;
  return MK_OK;
}

}  // extern "C"

int mk::dist_check_ranks(const char *who, int R, mk_dist_matrix **A) {
  if (R < 1 || !A) return fail(MK_ERR_INVALID, "%s: bad argument", who);
  for (int r = 0; r < R; r++) {
    if (!A[r]) return fail(MK_ERR_INVALID, "%s: null matrix", who);
    if (!A[r]->plan_ready) return fail(MK_ERR_INVALID, "%s: exchange plan incomplete (create the matrices of all ranks first)", who);
  }
  const bool local = A[0]->comm && A[0]->comm->local;
  if (local) {
    if (R != A[0]->nranks) return fail(MK_ERR_INVALID, "%s: a local world of %d ranks must be driven with all of them", who, A[0]->nranks);
    for (int r = 0; r < R; r++) if (A[r]->rank != r) return fail(MK_ERR_INVALID, "%s: matrices must be passed in rank order", who);
  } else if (R != 1) {
    return fail(MK_ERR_INVALID, "%s: one rank per process over RCCL", who);
  }
  return MK_OK;
}

extern "C" {

// y = A x on every rank this process drives (collective over RCCL)
int mk_dist_spmv(int R, mk_dist_matrix **A, mk_vector **x, mk_vector **y) {
  MK_REQUIRE_INIT();
  MK_TRY(dist_check_ranks("mk_dist_spmv", R, A));
  if (!x || !y) return fail(MK_ERR_INVALID, "mk_dist_spmv: null argument");
  std::vector<double *> xv(R);
  for (int r = 0; r < R; r++) {
    if (!x[r] || !y[r] || x[r]->cap < A[r]->n_local + A[r]->n_ghost || y[r]->n < A[r]->n_local)
      return fail(MK_ERR_INVALID, "mk_dist_spmv: vectors must come from mk_dist_vector_create of the same matrix");
    xv[r] = x[r]->d;
  }
  std::vector<double *> yv(R);
  for (int r = 0; r < R; r++) yv[r] = y[r]->d;
  return dist_spmv(R, A, xv.data(), yv.data(), 1.0, 0.0, nullptr);
}

// HYPRE_ParCSRPCGSetup + Solve over the row blocks (solvers.c:747-754 with num_procs > 1)
int mk_dist_pcg_solve(int R, mk_pcg **solvers, mk_dist_matrix **A, mk_vector **b, mk_vector **x) {
  MK_REQUIRE_INIT();
  MK_TRY(dist_check_ranks("mk_dist_pcg_solve", R, A));
  if (!solvers || !b || !x) return fail(MK_ERR_INVALID, "mk_dist_pcg_solve: null argument");
  Context &c = ctx();
  mk_pcg *S0 = solvers[0];
  for (int r = 0; r < R; r++) {
    if (!solvers[r] || !b[r] || !x[r]) return fail(MK_ERR_INVALID, "mk_dist_pcg_solve: null argument");
    if (b[r]->n != A[r]->n_local || x[r]->n != A[r]->n_local || x[r]->cap < A[r]->n_local + A[r]->n_ghost)
      return fail(MK_ERR_INVALID, "mk_dist_pcg_solve: vectors must come from mk_dist_vector_create of the same matrix");
    if (solvers[r]->max_iter != S0->max_iter || solvers[r]->tol != S0->tol || solvers[r]->precond != S0->precond)
      return fail(MK_ERR_INVALID, "mk_dist_pcg_solve: ranks disagree on solver settings");
  }
  const bool want_norms = S0->logging > 0 || S0->print_level > 0;
  std::vector<DistWork> W(R);
  std::vector<double *> pv(R), xv(R), bufs(R), pold(R, nullptr);
  std::vector<int> vb(R), nb1(R), nb2(R), fusedv(R, 0), pfuse(R, 0);
  std::vector<double> zs(R, 1.0);
  std::vector<HaloFuse> hf(R);
  static const int no_fused = getenv("MK_NO_FUSED_CG") ? atoi(getenv("MK_NO_FUSED_CG")) : 0;
  static const int no_pfuse = getenv("MK_NO_FUSED_K3") ? atoi(getenv("MK_NO_FUSED_K3")) : 0;
  static const int no_beside = getenv("MK_NO_OFFD_OVERLAP") ? atoi(getenv("MK_NO_OFFD_OVERLAP")) : 0;   // A/B: off-rank product after the sweep
  static const int env_single = getenv("MK_CG_SINGLE_REDUCE") ? atoi(getenv("MK_CG_SINGLE_REDUCE")) : -1;   // tests / A/B: force the variant on or off
  const bool want_single = env_single >= 0 ? env_single != 0 : S0->single_reduction != 0;
  int rc = MK_OK;
  auto cleanup = [&]() { (void)hipDeviceSynchronize(); for (auto &w : W) w.release(); };
#define MK_D(expr) do { rc = (expr); if (rc != MK_OK) { cleanup(); return rc; } } while (0)
#define MK_DH(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(MK_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); } } while (0)
  for (int r = 0; r < R; r++) {
    mk_dist_matrix *D = A[r];
    const size_t nl = (size_t)D->n_local, ne = (size_t)(D->n_local + D->n_ghost);
    MK_D(dev_alloc(&W[r].r, want_single ? ne : nl));   // (single-reduction step: r is the vector that is exchanged)
    MK_D(dev_alloc(&W[r].p, ne));
    MK_D(dev_alloc(&W[r].s, want_single ? ne : nl));
    MK_D(dev_alloc(&W[r].slab1, (size_t)kMaxParts * kSlabStride));
    MK_D(dev_alloc(&W[r].slab2, (size_t)kMaxParts * kSlabStride));
    MK_D(dev_alloc(&W[r].buf, 8));
    MK_D(dev_alloc(&W[r].state, 1));
    if (want_norms) MK_D(dev_alloc(&W[r].norms, (size_t)S0->max_iter + 1));
    MK_D(analyse_matrix(&D->diag));
    if (S0->precond == MK_PRECOND_JACOBI) MK_D(ensure_dinv(&D->diag));   // the diagonal lives in the diag block
    pv[r] = W[r].p; xv[r] = x[r]->d; bufs[r] = W[r].buf;
    vb[r] = vec_blocks(D->n_local);
    // diag blocks served by the sweep kernel take the fused step (A p recomputed instead of stored; 76 B per row
    // and iteration instead of 90).  Decided per rank: both variants all-reduce the same 2 + 3 values.
    fusedv[r] = (D->diag.kernel == MK_SPMV_SWEEP) && !no_fused && D->n_local > 0;
    if (fusedv[r]) MK_D(dev_alloc(&W[r].t, (size_t)std::max(D->n_offd_rows, 1)));
    // ... and fold K3 into the first sweep when 1/diag is one number (or there is no preconditioner)
    zs[r] = 1.0;
    pfuse[r] = fusedv[r] && !no_pfuse && (S0->precond != MK_PRECOND_JACOBI || sweep_uniform_diag(&D->diag, &zs[r]));
    if (pfuse[r]) MK_D(dev_alloc(&W[r].p2, ne));
    solvers[r]->iterations = 0; solvers[r]->rel_res_norm = 0.0; solvers[r]->converged = 0; solvers[r]->breakdown = 0;
  }
  // The single-reduction step needs the sweep kernel with K3 in its window fill on EVERY rank (its one reduction carries
  // other values than the default step's two, so ranks cannot mix): agree on it, else everybody takes the default step.
  bool single = false;
  if (want_single) {
    double bad = 0.0;
    for (int r = 0; r < R; r++) if (!pfuse[r] || !sweep_single_ok(&A[r]->diag)) bad += 1.0;
    if (A[0]->comm != nullptr && !A[0]->comm->local) {
      MK_DH(hipMemcpyAsync(W[0].buf, &bad, sizeof(double), hipMemcpyHostToDevice, c.stream));
      MK_D(dist_all_reduce(R, A, bufs.data(), 1));
      MK_DH(hipMemcpyAsync(&bad, W[0].buf, sizeof(double), hipMemcpyDeviceToHost, c.stream));
      MK_DH(hipStreamSynchronize(c.stream));
    }
    single = (bad == 0.0);
    if (single)
      for (int r = 0; r < R; r++)
        if (A[r]->n_offd_rows > 0) {
          MK_D(dev_alloc(&W[r].tpf, (size_t)A[r]->n_local));
          MK_DH(hipMemsetAsync(W[r].tpf, 0, sizeof(double) * (size_t)A[r]->n_local, c.stream));
        }
  }
  for (int r = 0; r < R; r++) solvers[r]->single_reduction_used = single ? 1 : 0;
  PcgState *hst = static_cast<PcgState *>(c.pinned);
  const int two_norm = S0->two_norm, rel_change = S0->rel_change;

  // r = b - A x;  p = C r;  gamma = <r,p>;  bi_prod = <b,b>   (each dot all-reduced)
  MK_D(halo_exchange(R, A, xv.data(), false));
  for (int r = 0; r < R; r++) {
    mk_dist_matrix *D = A[r];
    const int64_t nl = D->n_local;
    if (nl) hipLaunchKernelGGL(copy_kernel, dim3(vb[r]), dim3(256), 0, c.stream, W[r].r, b[r]->d, nl);
    MK_D(launch_spmv(&D->diag, -1.0, x[r]->d, 1.0, W[r].r));
    MK_D(launch_offd<false>(D, x[r]->d, W[r].r, -1.0, nullptr, nullptr, nullptr));
    const double *dinv = (S0->precond == MK_PRECOND_JACOBI) ? D->diag.dinv : nullptr;
    hipLaunchKernelGGL(pcg_init_vectors_kernel, dim3(vb[r]), dim3(kVecThreads), 0, c.stream, nl, b[r]->d, W[r].r, dinv, W[r].p, two_norm, W[r].slab2);
    hipLaunchKernelGGL((finalize_slab_kernel<3>), dim3(1), dim3(kVecThreads), 0, c.stream, W[r].slab2, vb[r], W[r].buf);
  }
  MK_D(dist_all_reduce(R, A, bufs.data(), 3));
  for (int r = 0; r < R; r++)
    hipLaunchKernelGGL(pcg_init_scalars_kernel, dim3(1), dim3(kVecThreads), 0, c.stream, W[r].buf, 1, 1, S0->tol, two_norm, W[r].state, W[r].norms);
  MK_DH(hipMemcpyAsync(&hst[0], W[0].state, sizeof(PcgState), hipMemcpyDeviceToHost, c.stream));
  MK_DH(hipStreamSynchronize(c.stream));
  if (!std::isfinite(hst[0].bi_prod) || hst[0].bi_prod < 0.0 || !std::isfinite(hst[0].i_prod)) {
    // hypre_PCGSolve's ieee_check: NaN or Inf in b, x0 or A is an error, not a zero right-hand side
    for (int r = 0; r < R; r++) solvers[r]->breakdown = 4;
    cleanup();
    return fail(MK_ERR_INVALID, "mk_dist_pcg_solve: <b,b> = %g, <r,r> = %g: NaN or Inf in the right-hand side, initial guess or matrix", hst[0].bi_prod, hst[0].i_prod);
  }
  const bool zero_rhs = (hst[0].bi_prod == 0.0);
  if (zero_rhs) {
    for (int r = 0; r < R; r++)
      if (A[r]->n_local) hipLaunchKernelGGL(copy_kernel, dim3(vb[r]), dim3(256), 0, c.stream, x[r]->d, b[r]->d, A[r]->n_local);
  }

  bool stop = zero_rhs;
  // ---- single-reduction step: per iteration ONE exchange (of r, no scalar in front of it), ONE all-reduce of five values, two
  // sweeps.  K_A: <A r, r> (diag sweep, dots only) beside pack + exchange + off-rank product on the communication stream;
  // fold + all-reduce; K_B (MODE 6): test of the step before, beta, alpha by the Chronopoulos-Gear identity, p' = z + beta p in
  // the window fill, r' = r - alpha A p', x += alpha p', partials for the next reduction.  r and p ping-pong (W.s and W.p2
  // are the second buffers).  60 B per row and iteration against the default step's 68.
  if (single) {
    std::vector<double *> rcur(R), rnew(R), pnew(R);
    std::vector<int> nbk2(R, 0);
    const int tflags = (two_norm ? 1 : 0) | (rel_change ? 2 : 0);
    // the solution update every second step with both steps' terms (sweep MODE 9 / 10, as MODE 7 / 8 of the single-matrix solve):
    // 52 B per row and iteration on average
    static const int no_halfx = getenv("MK_NO_HALVED_X") ? atoi(getenv("MK_NO_HALVED_X")) : 0;
    const bool halfx = !no_halfx && S0->max_iter > 1;
    for (int r = 0; r < R; r++) { rcur[r] = W[r].r; rnew[r] = W[r].s; pv[r] = W[r].p; pnew[r] = W[r].p2; }
    int it = 0;
    for (; it < S0->max_iter && !stop; it++) {
      const int first = (it == 0), par = it & 1;
      MK_D(halo_exchange(R, A, rcur.data(), true));
      for (int r = 0; r < R; r++) MK_D(launch_sweep_cg_dots(&A[r]->diag, rcur[r], W[r].slab1, W[r].state, &nb1[r]));
      for (int r = 0; r < R; r++)
        MK_D(launch_offd_compact(A[r], rcur[r], W[r].t, W[r].slab1 + nb1[r], &W[r].state->done, &nb2[r], kSlabStride - nb1[r], halo_stream(A[r]), nullptr));
      MK_D(halo_mark(A[0]));
      for (int r = 0; r < R; r++) {
        MK_D(halo_wait(A[r], true));
        hipLaunchKernelGGL(single_fold_kernel, dim3(1), dim3(kVecThreads), 0, c.stream, W[r].slab1, nb1[r] + nb2[r], W[r].slab2, nbk2[r], W[r].buf);
      }
      MK_D(dist_all_reduce(R, A, bufs.data(), 5));
      for (int r = 0; r < R; r++) {
        if (W[r].tpf)
          hipLaunchKernelGGL(single_offd_kernel, dim3(grid1(A[r]->n_offd_rows, 1024)), dim3(256), 0, c.stream, A[r]->n_offd_rows, A[r]->offd_rows,
                             W[r].t, W[r].tpf, W[r].buf, W[r].state, zs[r], first, par);
        MK_D(launch_sweep_cg_single(&A[r]->diag, rcur[r], rnew[r], pv[r], pnew[r], x[r]->d, zs[r], W[r].tpf, W[r].tpf ? A[r]->gap0 : 0,
                                    W[r].tpf ? A[r]->gap1 : 0, W[r].buf, W[r].slab2, W[r].state, two_norm, rel_change, first, par, W[r].norms, &nbk2[r],
                                    halfx ? 1 + (it & 1) : 0, pnew[r] == W[r].p2 ? 1 : 0));
        std::swap(rcur[r], rnew[r]);
        std::swap(pv[r], pnew[r]);
      }
      if ((it + 1) % S0->check_every == 0) {
        MK_DH(hipMemcpyAsync(&hst[0], W[0].state, sizeof(PcgState), hipMemcpyDeviceToHost, c.stream));
        MK_DH(hipStreamSynchronize(c.stream));
        if (hst[0].done != 0) stop = true;
      }
    }
    if (halfx && it > 0)         // a solve that ended on a step that left x alone: its update is still pending
      for (int r = 0; r < R; r++) {
        hipLaunchKernelGGL(pcg_apply_pending_x_kernel, dim3(vb[r]), dim3(kVecThreads), 0, c.stream, A[r]->n_local, x[r]->d, W[r].p, W[r].p2, W[r].state);
        hipLaunchKernelGGL(pcg_clear_pending_kernel, dim3(1), dim3(1), 0, c.stream, W[r].state);
      }
    if (it > 0 && !zero_rhs) {   // the test of the last step (one more reduction, once per solve)
      for (int r = 0; r < R; r++)
        hipLaunchKernelGGL(single_fold_kernel, dim3(1), dim3(kVecThreads), 0, c.stream, W[r].slab1, 0, W[r].slab2, nbk2[r], W[r].buf);
      MK_D(dist_all_reduce(R, A, bufs.data(), 5));
      for (int r = 0; r < R; r++)
        hipLaunchKernelGGL(single_final_test_kernel, dim3(1), dim3(64), 0, c.stream, W[r].buf, W[r].state, tflags, W[r].norms, it & 1);
    }
    stop = true;   // (the default loop below does not run)
  }
  for (int it = 0; it < S0->max_iter && !stop; it++) {
    // ranks with K3 folded into the first sweep: from the second iteration on p' = z + beta p is formed inside that
    // sweep, into the other p buffer; its send rows are formed ahead by the pack kernel so the halo can start now
    for (int r = 0; r < R; r++) {
      hf[r] = HaloFuse();
      if (pfuse[r] && it > 0) {
        hf[r].r = W[r].r; hf[r].p_old = pv[r]; hf[r].buf3 = W[r].buf + 4; hf[r].st = W[r].state;
        hf[r].zscale = zs[r]; hf[r].flags = (two_norm ? 1 : 0) | (rel_change ? 2 : 0);
        pold[r] = pv[r];
        pv[r] = (pv[r] == W[r].p) ? W[r].p2 : W[r].p;
      }
    }
    MK_D(halo_exchange(R, A, pv.data(), true, hf.data()));   // on the communication stream (local world: its side stream)
    for (int r = 0; r < R; r++) {                                                 // K1a: diag block, overlaps the halo
      if (pfuse[r] && it > 0)
        MK_D(launch_sweep_cg_pdots(&A[r]->diag, W[r].r, pold[r], pv[r], zs[r], W[r].buf + 4, 1, 1, W[r].slab1, W[r].state, two_norm,
                                   rel_change, W[r].norms, &nb1[r]));
      else if (fusedv[r]) MK_D(launch_sweep_cg_dots(&A[r]->diag, pv[r], W[r].slab1, W[r].state, &nb1[r]));
      else MK_D(launch_spmv_dot(&A[r]->diag, pv[r], W[r].s, W[r].slab1, &W[r].state->done, &nb1[r]));
    }
    // K1b, fused ranks: the off-rank product into the compact buffer t runs on the stream the ghosts arrive on, BESIDE the
    // diag sweep (it writes t and its own partials only; where the sweep is still forming p' the kernel forms the local
    // operand of its dot itself).  The off-rank partials sit behind the diag kernel's in the same slab: nb1 + nb2 <=
    // kSlabStride, or component 0 of the off-rank part would run into component 1 of the diag part.
    bool beside = false;
    for (int r = 0; r < R; r++)
      if (fusedv[r] && !no_beside) {
        const HaloFuse *f = (pfuse[r] && it > 0) ? &hf[r] : nullptr;
        MK_D(launch_offd_compact(A[r], pv[r], W[r].t, W[r].slab1 + nb1[r], &W[r].state->done, &nb2[r], kSlabStride - nb1[r], halo_stream(A[r]), f));
        beside = true;
      }
    if (beside) MK_D(halo_mark(A[0]));
    for (int r = 0; r < R; r++) {                                                 // the rest once the ghosts are in
      MK_D(halo_wait(A[r], true));
      if (fusedv[r] && no_beside)
        MK_D(launch_offd_compact(A[r], pv[r], W[r].t, W[r].slab1 + nb1[r], &W[r].state->done, &nb2[r], kSlabStride - nb1[r], c.stream, nullptr));
      else if (!fusedv[r]) MK_D(launch_offd<true>(A[r], pv[r], W[r].s, 1.0, W[r].slab1 + nb1[r], &W[r].state->done, &nb2[r], kSlabStride - nb1[r]));
      hipLaunchKernelGGL((finalize_slab_kernel<2>), dim3(1), dim3(kVecThreads), 0, c.stream, W[r].slab1, nb1[r] + nb2[r], W[r].buf);
    }
    MK_D(dist_all_reduce(R, A, bufs.data(), 2));                                       // <s,p>, <p,p>
    for (int r = 0; r < R; r++) {
      const int64_t nl = A[r]->n_local;
      double *buf3 = W[r].buf + 4;
      int nbk2 = vb[r];
      if (fusedv[r]) {
        if (A[r]->n_offd_rows > 0)
          hipLaunchKernelGGL(offd_resid_kernel, dim3(grid1(A[r]->n_offd_rows, 1024)), dim3(256), 0, c.stream, A[r]->n_offd_rows,
                             A[r]->offd_rows, W[r].t, W[r].r, W[r].buf, W[r].state);
        MK_D(launch_sweep_cg_resid_x(&A[r]->diag, pv[r], W[r].r, x[r]->d, S0->precond == MK_PRECOND_JACOBI ? A[r]->diag.dinv : nullptr,
                                     W[r].buf, 1, 1, W[r].slab2, W[r].state, &nbk2));
      } else if (S0->precond == MK_PRECOND_JACOBI)
        hipLaunchKernelGGL((pcg_update_xr_kernel<true>), dim3(vb[r]), dim3(kVecThreads), 0, c.stream, nl, pv[r], W[r].s, A[r]->diag.dinv, x[r]->d, W[r].r, W[r].buf, 1, 1, W[r].slab2, W[r].state, vec_flags(nl));
      else
        hipLaunchKernelGGL((pcg_update_xr_kernel<false>), dim3(vb[r]), dim3(kVecThreads), 0, c.stream, nl, pv[r], W[r].s, (const double *)nullptr, x[r]->d, W[r].r, W[r].buf, 1, 1, W[r].slab2, W[r].state, vec_flags(nl));
      hipLaunchKernelGGL((finalize_slab_kernel<3>), dim3(1), dim3(kVecThreads), 0, c.stream, W[r].slab2, nbk2, buf3);
      bufs[r] = buf3;
    }
    MK_D(dist_all_reduce(R, A, bufs.data(), 3));                                       // <r,z>, <r,r>, <x,x>
    for (int r = 0; r < R; r++) {
      const int64_t nl = A[r]->n_local;
      bufs[r] = W[r].buf;
      if (pfuse[r]) continue;   // K3 rides in the next iteration's first sweep
      if (S0->precond == MK_PRECOND_JACOBI)
        hipLaunchKernelGGL((pcg_update_p_kernel<true>), dim3(vb[r]), dim3(kVecThreads), 0, c.stream, nl, W[r].r, A[r]->diag.dinv, pv[r], W[r].buf + 4, 1, 1, two_norm, rel_change, W[r].state, W[r].norms, vec_flags(nl));
      else
        hipLaunchKernelGGL((pcg_update_p_kernel<false>), dim3(vb[r]), dim3(kVecThreads), 0, c.stream, nl, W[r].r, (const double *)nullptr, pv[r], W[r].buf + 4, 1, 1, two_norm, rel_change, W[r].state, W[r].norms, vec_flags(nl));
      bufs[r] = W[r].buf;
    }
    if ((it + 1) % S0->check_every == 0) {
      MK_DH(hipMemcpyAsync(&hst[0], W[0].state, sizeof(PcgState), hipMemcpyDeviceToHost, c.stream));
      MK_DH(hipStreamSynchronize(c.stream));
      if (hst[0].done != 0) stop = true;   // the same on every rank: it derives from all-reduced values only
    }
  }
  for (int r = 0; r < R; r++)   // the test the next iteration's first sweep would have made (K3's prologue; n = 0: no p update)
    if (pfuse[r] && S0->max_iter > 0 && !zero_rhs && !single)
      hipLaunchKernelGGL((pcg_update_p_kernel<false>), dim3(1), dim3(kVecThreads), 0, c.stream, (int64_t)0, W[r].r, (const double *)nullptr, pv[r], W[r].buf + 4, 1, 1, two_norm, rel_change, W[r].state, W[r].norms, 0);
  MK_DH(hipGetLastError());
  MK_DH(hipMemcpyAsync(&hst[0], W[0].state, sizeof(PcgState), hipMemcpyDeviceToHost, c.stream));
  MK_DH(hipDeviceSynchronize());
  for (int r = 0; r < R; r++) {
    solvers[r]->iterations = hst[0].iter;
    solvers[r]->converged = hst[0].converged;
    solvers[r]->breakdown = hst[0].breakdown;
    solvers[r]->rel_res_norm = zero_rhs ? 0.0 : sqrt(hst[0].i_prod / hst[0].bi_prod);
    // hand the residual history to the solver object so mk_pcg_get_residual_history works
    if (want_norms) {
      dev_free(solvers[r]->norms);
      solvers[r]->norms = W[r].norms;
      solvers[r]->norms_cap = S0->max_iter + 1;
      W[r].norms = nullptr;
    }
    if (solvers[r]->print_level > 1 && solvers[r]->norms && !zero_rhs && hst[0].iter <= S0->max_iter) {
      std::vector<double> h((size_t)hst[0].iter + 1);
      if (hipMemcpyAsync(h.data(), solvers[r]->norms, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c.stream) == hipSuccess &&
          hipStreamSynchronize(c.stream) == hipSuccess)
        print_pcg_table(two_norm, h.data(), hst[0].iter, sqrt(hst[0].bi_prod));
    }
  }
  cleanup();
#undef MK_D
#undef MK_DH
  return MK_OK;
}


// plain Jacobi on the distributed operator (BASELINE config 5).  Per iteration: halo of x,
// off-rank part of A x into a scratch vector (rows that touch ghosts only), then the fused
// Jacobi pass over the diag block.  The residual norm is all-reduced only at the iterations
// where it is tested (every check_every).
int mk_dist_jacobi_solve(int R, mk_jacobi **solvers, mk_dist_matrix **A, mk_vector **b, mk_vector **x) {
  MK_REQUIRE_INIT();
  MK_TRY(dist_check_ranks("mk_dist_jacobi_solve", R, A));
  if (!solvers || !b || !x) return fail(MK_ERR_INVALID, "mk_dist_jacobi_solve: null argument");
  static const int no_gap = getenv("MK_NO_EXTRA_GAP") ? atoi(getenv("MK_NO_EXTRA_GAP")) : 0;   // A/B: the sweep reads the whole off-rank vector
  Context &c = ctx();
  mk_jacobi *S0 = solvers[0];
  for (int r = 0; r < R; r++) {
    if (!solvers[r] || !b[r] || !x[r]) return fail(MK_ERR_INVALID, "mk_dist_jacobi_solve: null argument");
    if (b[r]->n != A[r]->n_local || x[r]->n != A[r]->n_local || x[r]->cap < A[r]->n_local + A[r]->n_ghost)
      return fail(MK_ERR_INVALID, "mk_dist_jacobi_solve: vectors must come from mk_dist_vector_create of the same matrix");
    if (solvers[r]->max_iter != S0->max_iter || solvers[r]->tol != S0->tol || solvers[r]->check_every != S0->check_every)
      return fail(MK_ERR_INVALID, "mk_dist_jacobi_solve: ranks disagree on solver settings");
  }
  struct JW { double *xalt = nullptr, *extra = nullptr, *slab = nullptr, *buf = nullptr; JacobiState *state = nullptr; };
  std::vector<JW> W(R);
  std::vector<double *> cur(R), nxt(R), bufs(R);
  std::vector<int> blocks(R), vb(R), split(R, 0), pb(R, 0), pe(R, 0), nplanes(R, 0), imax(R, kSlabStride);
  // split sweep (see the loop): default over RCCL, where the exchange has a latency worth hiding; in a local world (one GPU,
  // nothing to hide, two more launches per rank) only on request.  MK_JACOBI_SPLIT=0 / 1 forces it off / on.
  static const int env_split = getenv("MK_JACOBI_SPLIT") ? atoi(getenv("MK_JACOBI_SPLIT")) : -1;
  const bool want_split = env_split >= 0 ? env_split != 0 : (A[0]->comm != nullptr && !A[0]->comm->local);
  auto cleanup = [&]() {
    (void)hipDeviceSynchronize();
    for (auto &w : W) { dev_free(w.xalt); dev_free(w.extra); dev_free(w.slab); dev_free(w.buf); dev_free(w.state); }
  };
  int rc = MK_OK;
#define MK_D(expr) do { rc = (expr); if (rc != MK_OK) { cleanup(); return rc; } } while (0)
#define MK_DH(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(MK_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); } } while (0)
  for (int r = 0; r < R; r++) {
    mk_dist_matrix *D = A[r];
    const size_t nl = (size_t)D->n_local;
    MK_D(dev_alloc(&W[r].xalt, nl + (size_t)D->n_ghost));
    MK_D(dev_alloc(&W[r].extra, nl));
    MK_D(dev_alloc(&W[r].slab, (size_t)kSlabStride));
    MK_D(dev_alloc(&W[r].buf, 2));
    MK_D(dev_alloc(&W[r].state, 1));
    MK_DH(hipMemsetAsync(W[r].extra, 0, sizeof(double) * nl, c.stream));
    MK_D(analyse_matrix(&D->diag));
    cur[r] = x[r]->d; nxt[r] = W[r].xalt; bufs[r] = W[r].buf;
    vb[r] = vec_blocks(D->n_local);
    const int G = D->diag.lanes;
    blocks[r] = (int)std::max<int64_t>(1, std::min<int64_t>((D->n_local + 256 / G - 1) / (256 / G), std::min<int64_t>((int64_t)c.num_cus * 8, kSlabStride)));
    solvers[r]->iterations = 0; solvers[r]->rel_res_norm = 0.0; solvers[r]->converged = 0;
    int P = 0, K = 0;
    if (want_split && !no_gap && A[0]->comm != nullptr && sweep_planes(&D->diag, &P, &K) && D->n_offd_rows > 0) {
      nplanes[r] = K;
      pb[r] = (int)((D->gap0 + P - 1) / P);     // first plane wholly inside the ghost-free run of rows
      pe[r] = (int)(D->gap1 / P);               // one past the last such plane
      if (D->gap1 >= (int)D->n_local) pe[r] = K;
      split[r] = pe[r] - pb[r] >= 8 && (pb[r] > 0 || pe[r] < K);
      if (split[r]) {
        // The partials of the interior launch and of the (at most two) boundary launches share one slab of kSlabStride
        // slots: decided HERE, never in the loop.  The boundary launches take what their planes need (one segment each),
        // the interior launch gets the rest (fewer segments if need be); a grid that does not fit keeps the one-launch sweep.
        const int b0 = pb[r] > 0 ? jacobi_sweep_blocks(&D->diag, 0, pb[r], kSlabStride) : 0;
        const int b1 = pe[r] < K ? jacobi_sweep_blocks(&D->diag, pe[r], K, kSlabStride) : 0;
        const bool edges_ok = (pb[r] == 0 || b0 > 0) && (pe[r] >= K || b1 > 0);
        imax[r] = kSlabStride - b0 - b1;
        if (!edges_ok || imax[r] < 8 || jacobi_sweep_blocks(&D->diag, pb[r], pe[r], imax[r]) <= 0) split[r] = 0;
      }
    }
  }
  JacobiState *hst = static_cast<JacobiState *>(c.pinned);
  // ||b||^2
  for (int r = 0; r < R; r++) {
    hipLaunchKernelGGL(dot_self_kernel, dim3(vb[r]), dim3(256), 0, c.stream, A[r]->n_local, b[r]->d, W[r].slab);
    hipLaunchKernelGGL((finalize_slab_kernel<1>), dim3(1), dim3(kVecThreads), 0, c.stream, W[r].slab, vb[r], W[r].buf);
  }
  MK_D(dist_all_reduce(R, A, bufs.data(), 1));
  for (int r = 0; r < R; r++) hipLaunchKernelGGL(jacobi_init_kernel, dim3(1), dim3(256), 0, c.stream, W[r].buf, 1, 1, S0->tol, W[r].state);
  MK_DH(hipMemcpyAsync(hst, W[0].state, sizeof(JacobiState), hipMemcpyDeviceToHost, c.stream));
  MK_DH(hipStreamSynchronize(c.stream));
  if (!std::isfinite(hst->bb) || hst->bb < 0.0) {
    cleanup();
    return fail(MK_ERR_INVALID, "mk_dist_jacobi_solve: <b,b> = %g: NaN or Inf in the right-hand side", hst->bb);
  }
  if (hst->bb == 0.0) {
    for (int r = 0; r < R; r++)
      if (A[r]->n_local) hipLaunchKernelGGL(copy_kernel, dim3(vb[r]), dim3(256), 0, c.stream, x[r]->d, b[r]->d, A[r]->n_local);
    cleanup();
    return MK_OK;
  }
  long long k = 0;
  int checks = 0;
  for (;;) {
    // Split sweep (diag block on the sweep kernel, enough planes inside the ghost-free run of rows): the planes whose rows
    // touch no ghost are swept while the halo of x travels and the off-rank product forms on the halo's stream; the few
    // boundary planes follow once both are in.  Same kernel, same rows, same arithmetic as the one-launch sweep.
    bool all_split = true;
    for (int r = 0; r < R; r++) all_split = all_split && split[r];
    if (all_split) {
      MK_D(halo_exchange(R, A, cur.data(), true));
      for (int r = 0; r < R; r++)
        MK_D((launch_offd<false, true>(A[r], cur[r], W[r].extra, 1.0, nullptr, &W[r].state->done, nullptr, 1024, halo_stream(A[r]))));
      MK_D(halo_mark(A[0]));
      for (int r = 0; r < R; r++) {   // interior planes
        int nb = 0;
        MK_D(launch_jacobi_sweep(&A[r]->diag, b[r]->d, W[r].extra, cur[r], nxt[r], W[r].slab, &W[r].state->done, &nb, A[r]->gap0, A[r]->gap1,
                                 pb[r], pe[r], imax[r]));
        blocks[r] = nb;
      }
      for (int r = 0; r < R; r++) {   // boundary planes once ghosts and off-rank product are in
        MK_D(halo_wait(A[r], true));
        const int ranges[2][2] = {{0, pb[r]}, {pe[r], nplanes[r]}};
        for (int q = 0; q < 2; q++) {
          if (ranges[q][1] <= ranges[q][0]) continue;
          int nb = 0;
          // fits by construction (set-up sized the interior launch around the boundary launches' needs); the launcher checks again
          MK_D(launch_jacobi_sweep(&A[r]->diag, b[r]->d, W[r].extra, cur[r], nxt[r], W[r].slab + blocks[r], &W[r].state->done, &nb, A[r]->gap0,
                                   A[r]->gap1, ranges[q][0], ranges[q][1], kSlabStride - blocks[r]));
          blocks[r] += nb;
        }
      }
    } else {
    MK_D(halo_exchange(R, A, cur.data(), false));
    for (int r = 0; r < R; r++) {
      mk_dist_matrix *D = A[r];
      MK_D((launch_offd<false, true>(D, cur[r], W[r].extra, 1.0, nullptr, &W[r].state->done, nullptr)));
      const mk_matrix &M = D->diag;
      int sweep_blocks = 0;
      const int src = launch_jacobi_sweep(&D->diag, b[r]->d, W[r].extra, cur[r], nxt[r], W[r].slab, &W[r].state->done, &sweep_blocks,
                                          no_gap ? 0 : D->gap0, no_gap ? 0 : D->gap1);
      if (src == MK_OK) {
        blocks[r] = sweep_blocks;
      } else if (src == MK_ERR_UNSUPPORTED) {
#define MK_JSTEP(GG) case GG: hipLaunchKernelGGL((jacobi_step_kernel<GG>), dim3(blocks[r]), dim3(256), 0, c.stream, D->n_local, M.rowptr, M.col, M.val, b[r]->d, cur[r], nxt[r], W[r].slab, W[r].state, (const double *)W[r].extra); break;
      switch (M.lanes) { MK_JSTEP(1) MK_JSTEP(2) MK_JSTEP(4) MK_JSTEP(8) MK_JSTEP(16) MK_JSTEP(32) MK_JSTEP(64)
        default: cleanup(); return fail(MK_ERR_INVALID, "jacobi: bad lanes per row %d", M.lanes); }
#undef MK_JSTEP
      } else {
        cleanup();
        return src;
      }
    }
    }
    const bool last = (k >= S0->max_iter);
    if (k % S0->check_every == 0 || last) {
      for (int r = 0; r < R; r++)
        hipLaunchKernelGGL((finalize_slab_kernel<1>), dim3(1), dim3(kVecThreads), 0, c.stream, W[r].slab, blocks[r], W[r].buf);
      MK_D(dist_all_reduce(R, A, bufs.data(), 1));
      for (int r = 0; r < R; r++)
        hipLaunchKernelGGL(jacobi_check_kernel, dim3(1), dim3(256), 0, c.stream, W[r].buf, 1, 1, k, W[r].state);
      checks++;
      if (last || checks % 4 == 0) {
        MK_DH(hipMemcpyAsync(hst, W[0].state, sizeof(JacobiState), hipMemcpyDeviceToHost, c.stream));
        MK_DH(hipStreamSynchronize(c.stream));
        if (hst->done != 0 || last) break;
      }
    }
    for (int r = 0; r < R; r++) std::swap(cur[r], nxt[r]);
    k++;
  }
  MK_DH(hipGetLastError());
  for (int r = 0; r < R; r++) {
    solvers[r]->iterations = hst->iter;
    solvers[r]->converged = hst->converged;
    solvers[r]->rel_res_norm = sqrt(hst->rr / hst->bb);
    if ((hst->iter & 1) && A[r]->n_local)
      hipLaunchKernelGGL(copy_kernel, dim3(vb[r]), dim3(256), 0, c.stream, x[r]->d, W[r].xalt, A[r]->n_local);
  }
  cleanup();
#undef MK_D
#undef MK_DH
  return MK_OK;
}

}  // extern "C"
