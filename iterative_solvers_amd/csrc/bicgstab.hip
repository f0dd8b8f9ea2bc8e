// bicgstab.hip -- BiCGSTAB (solver id 6) and CGNR (solver id 40) on one GPU.
//
// Stands in for HYPRE_ParCSRBiCGSTABSetup/Solve as the reference drives them for solver id 6
// (solvers.c:517-564: PrintLevel, Logging, Tol, MaxIter, identity preconditioner).
// hypre_BiCGSTABSolve is not in /root/reference; the loop follows its published algorithm
// (oracle/oracle.c orc_bicgstab lists the steps).  PARITY UNPINNED.
//
// Per iteration: two SpMVs (any of the kernels of spmv.hip), five vector kernels, and two
// conditional kernels that only run when the recursive residual passes the test (HYPRE then
// recomputes r = b - A x and tests again).  All scalars (alpha, gamma, beta, norms, flags) live
// in a device record; every reduction is finished in the next kernel's prologue from a slab of
// per-block partials (fixed order, no atomics); the host polls the stop flag every check_every
// iterations.  Algorithmic bytes per iteration: 2 SpMV + 160 n (the dots step 3 needs ride in step 2).
#include <cfloat>
#include <cmath>

#include <vector>

#include "dist_internal.h"
#include "krylov_kernels.h"

namespace mk {

struct BicgState {
  // resv: <r0,r> ping-pong.  Iteration `it` reads resv[it & 1] (steps 1 and 3) and step 3 writes resv[(it + 1) & 1]:
  // no kernel reads a field that a block of the same launch may already have rewritten.
  double resv[2], alpha, gamma, r_norm, b_norm, eps;
  long long iter;
  int done;        // 0 running, 1 converged, 2 breakdown
  int converged;
  int breakdown;   // 1 <r0,q> ~ 0, 2 res ~ 0, 3 gamma ~ 0
  int skip_true;   // 1: the true-residual kernels of this iteration return at once
};

constexpr double kEpsMac = 1.e-128;

// partials of <a1,b1> (component 0) and, when a2 != null, <a2,b2> (component 1)
static __global__ __launch_bounds__(kVecThreads) void dot2_kernel(int64_t n, const double *__restrict__ a1, const double *__restrict__ b1,
                                                                  const double *__restrict__ a2, const double *__restrict__ b2,
                                                                  double *__restrict__ slab, const int *__restrict__ done, int nt) {
  if (done != nullptr && *done != 0) return;
  double v[2] = {0.0, 0.0};
  const int64_t n2 = n >> 1;
  auto rows = [&](auto tag) {
    constexpr bool NT = decltype(tag)::value;
    for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kVecThreads) {
      const double2 x1 = vec_ld<NT>(reinterpret_cast<const double2 *>(a1) + i), y1 = vec_ld<NT>(reinterpret_cast<const double2 *>(b1) + i);
      v[0] += x1.x * y1.x + x1.y * y1.y;
      if (a2) {
        const double2 x2 = vec_ld<NT>(reinterpret_cast<const double2 *>(a2) + i), y2 = vec_ld<NT>(reinterpret_cast<const double2 *>(b2) + i);
        v[1] += x2.x * y2.x + x2.y * y2.y;
      }
    }
  };
  if (nt) rows(std::true_type{}); else rows(std::false_type{});
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    v[0] += a1[n - 1] * b1[n - 1];
    if (a2) v[1] += a2[n - 1] * b2[n - 1];
  }
  __shared__ double smem[2 * (kVecThreads / 64)];
  block_sum<2, kVecThreads>(v, smem);
  if (threadIdx.x == 0) {
    slab[blockIdx.x] = v[0];
    slab[kSlabStride + blockIdx.x] = v[1];
  }
}

// after r0 = b - A x: r = p = r0, partials of <b,b>, <r0,r0>
static __global__ __launch_bounds__(kVecThreads) void bicg_init_vectors_kernel(int64_t n, const double *__restrict__ b, const double *__restrict__ r0,
                                                                               double *__restrict__ r, double *__restrict__ p, double *__restrict__ slab) {
  double v[2] = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kVecThreads) {
    const double bi = b[i], ri = r0[i];
    r[i] = ri;
    p[i] = ri;
    v[0] += bi * bi;
    v[1] += ri * ri;
  }
  __shared__ double smem[2 * (kVecThreads / 64)];
  block_sum<2, kVecThreads>(v, smem);
  if (threadIdx.x == 0) {
    slab[blockIdx.x] = v[0];
    slab[kSlabStride + blockIdx.x] = v[1];
  }
}

static __global__ __launch_bounds__(kVecThreads) void bicg_init_scalars_kernel(const double *slab, int nblocks, int stride, double tol, BicgState *st, double *norms) {
  __shared__ double smem[2 * (kVecThreads / 64)];
  double v[2];
  reduce_slab<2, kVecThreads>(slab, nblocks, stride, v, smem);
  if (threadIdx.x == 0) {
    BicgState s;
    memset(&s, 0, sizeof(s));
    s.b_norm = sqrt(v[0]);
    s.resv[0] = v[1];
    s.r_norm = sqrt(v[1]);
    s.eps = tol * (s.b_norm > 0.0 ? s.b_norm : s.r_norm);
    s.done = (s.r_norm == 0.0) ? 1 : 0;   // hypre returns at once, zero iterations
    s.skip_true = 1;
    *st = s;
    if (norms) norms[0] = s.r_norm;
  }
}

// alpha = res / <r0,q>;  x += alpha p;  r -= alpha q
static __global__ __launch_bounds__(kVecThreads) void bicg_step1_kernel(int64_t n, const double *__restrict__ p, const double *__restrict__ q,
                                                                        double *__restrict__ x, double *__restrict__ r,
                                                                        const double *__restrict__ slab_in, int nblocks_in, BicgState *st, int nt,
                                                                        int parity) {
  if (block_flag_raised(&st->done)) return;
  __shared__ double smem[kVecThreads / 64];
  double d[1];
  reduce_slab<1, kVecThreads>(slab_in, nblocks_in, kSlabStride, d, smem);
  const double temp = d[0], res = st->resv[parity];
  const bool leader = (blockIdx.x == 0 && threadIdx.x == 0);
  if (!(fabs(temp) >= kEpsMac)) {
    if (leader) { st->iter += 1; st->breakdown = 1; st->done = 2; }
    return;
  }
  const double alpha = res / temp;
  const int64_t n2 = n >> 1;
  auto rows = [&](auto tag) {
    constexpr bool NT = decltype(tag)::value;
    for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kVecThreads) {
      const double2 pv = vec_ld<NT>(reinterpret_cast<const double2 *>(p) + i), qv = vec_ld<NT>(reinterpret_cast<const double2 *>(q) + i);
      double2 xv = vec_ld<NT>(reinterpret_cast<const double2 *>(x) + i), rv = vec_ld<NT>(reinterpret_cast<const double2 *>(r) + i);
      xv.x += alpha * pv.x; xv.y += alpha * pv.y;
      rv.x -= alpha * qv.x; rv.y -= alpha * qv.y;
      vec_st<NT>(reinterpret_cast<double2 *>(x) + i, xv);
      vec_st<NT>(reinterpret_cast<double2 *>(r) + i, rv);
    }
  };
  if (nt) rows(std::true_type{}); else rows(std::false_type{});
  if ((n & 1) && leader) {
    x[n - 1] += alpha * p[n - 1];
    r[n - 1] -= alpha * q[n - 1];
  }
  if (leader) { st->alpha = alpha; st->iter += 1; }
}

// gamma = <r,s>/<s,s> (0/0 := 0);  x += gamma r;  r -= gamma s;  partials of <r0,r>, <r,r> (what step 3 needs, unless
// the true residual replaces r in between: then the dot kernel behind it overwrites them)
static __global__ __launch_bounds__(kVecThreads) void bicg_step2_kernel(int64_t n, const double *__restrict__ s, const double *__restrict__ r0, double *__restrict__ x,
                                                                        double *__restrict__ r, const double *__restrict__ slab_in,
                                                                        int nblocks_in, int stride_in, double *__restrict__ slab_out, BicgState *st, int nt) {
  if (st->done != 0) return;
  __shared__ double smem[2 * (kVecThreads / 64)];
  double d[2];
  reduce_slab<2, kVecThreads>(slab_in, nblocks_in, stride_in, d, smem);
  const double gamma = (d[0] == 0.0 && d[1] == 0.0) ? 0.0 : d[0] / d[1];
  double v[2] = {0.0, 0.0};
  const int64_t n2 = n >> 1;
  auto rows = [&](auto tag) {
    constexpr bool NT = decltype(tag)::value;
    for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kVecThreads) {
      double2 rv = vec_ld<NT>(reinterpret_cast<const double2 *>(r) + i), xv = vec_ld<NT>(reinterpret_cast<const double2 *>(x) + i);
      const double2 sv = vec_ld<NT>(reinterpret_cast<const double2 *>(s) + i);
      xv.x += gamma * rv.x; xv.y += gamma * rv.y;
      rv.x -= gamma * sv.x; rv.y -= gamma * sv.y;
      vec_st<NT>(reinterpret_cast<double2 *>(x) + i, xv);
      vec_st<NT>(reinterpret_cast<double2 *>(r) + i, rv);
      const double2 zv = vec_ld<NT>(reinterpret_cast<const double2 *>(r0) + i);
      v[0] += zv.x * rv.x + zv.y * rv.y;
      v[1] += rv.x * rv.x + rv.y * rv.y;
    }
  };
  if (nt) rows(std::true_type{}); else rows(std::false_type{});
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double ri = r[n - 1];
    x[n - 1] += gamma * ri;
    ri -= gamma * s[n - 1];
    r[n - 1] = ri;
    v[0] += r0[n - 1] * ri;
    v[1] += ri * ri;
  }
  block_sum<2, kVecThreads>(v, smem);
  if (threadIdx.x == 0) {
    slab_out[blockIdx.x] = v[0];
    slab_out[kSlabStride + blockIdx.x] = v[1];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) st->gamma = gamma;
}

// ||r|| of the recursive residual; decides whether the true residual has to be formed
static __global__ __launch_bounds__(kVecThreads) void bicg_norm_kernel(const double *slab, int nblocks, BicgState *st, double *norms) {
  if (st->done != 0) {   // one block: after a stop the true-residual kernels of the iterations still queued return at once
    if (threadIdx.x == 0) st->skip_true = 1;
    return;
  }
  __shared__ double smem[kVecThreads / 64];
  double v[1];
  reduce_slab<1, kVecThreads>(slab, nblocks, kSlabStride, v, smem);
  if (threadIdx.x == 0) {
    const double rn = sqrt(v[0]);
    st->r_norm = rn;
    if (norms) norms[st->iter] = rn;
    st->skip_true = (rn <= st->eps) ? 0 : 1;
  }
}

static __global__ void copy_unless_kernel(double *__restrict__ dst, const double *__restrict__ src, int64_t n, const int *__restrict__ skip) {
  if (*skip != 0) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// slab_in: <r0,r>, <r,r> of the residual as it stands now (replaced by b - A x if it was a candidate).
// Convergence on the true residual; beta = (<r0,r>/res)(alpha/gamma); p = beta (p - gamma q) + r
static __global__ __launch_bounds__(kVecThreads) void bicg_step3_kernel(int64_t n, const double *__restrict__ r, const double *__restrict__ q,
                                                                        double *__restrict__ p, const double *__restrict__ slab_in,
                                                                        int nblocks_in, int stride_in, BicgState *st, int nt, int parity) {
  if (block_flag_raised(&st->done)) return;
  __shared__ double smem[2 * (kVecThreads / 64)];
  double d[2];
  reduce_slab<2, kVecThreads>(slab_in, nblocks_in, stride_in, d, smem);
  const bool leader = (blockIdx.x == 0 && threadIdx.x == 0);
  // skip_true, alpha, gamma and resv[parity] were written by earlier launches and are not written here
  const bool cand = (st->skip_true == 0);
  const double res_old = st->resv[parity], alpha = st->alpha, gamma = st->gamma, eps = st->eps;
  if (cand) {
    const double rn = sqrt(d[1]);
    if (leader) st->r_norm = rn;
    if (rn <= eps) {
      if (leader) { st->converged = 1; st->done = 1; }
      return;
    }
  }
  if (!(fabs(res_old) >= kEpsMac)) {
    if (leader) { st->breakdown = 2; st->done = 2; }
    return;
  }
  double beta = 1.0 / res_old;
  const double res = d[0];
  beta *= res;
  beta *= alpha;
  if (!(fabs(gamma) >= kEpsMac)) {
    if (leader) { st->resv[parity ^ 1] = res; st->breakdown = 3; st->done = 2; }
    return;
  }
  beta /= gamma;
  const int64_t n2 = n >> 1;
  auto rows = [&](auto tag) {
    constexpr bool NT = decltype(tag)::value;
    for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kVecThreads) {
      double2 pv = vec_ld<NT>(reinterpret_cast<const double2 *>(p) + i);
      const double2 qv = vec_ld<NT>(reinterpret_cast<const double2 *>(q) + i), rv = vec_ld<NT>(reinterpret_cast<const double2 *>(r) + i);
      double t0 = pv.x + (-gamma) * qv.x, t1 = pv.y + (-gamma) * qv.y;
      t0 = beta * t0; t1 = beta * t1;
      pv.x = t0 + rv.x; pv.y = t1 + rv.y;
      vec_st<NT>(reinterpret_cast<double2 *>(p) + i, pv);
    }
  };
  if (nt) rows(std::true_type{}); else rows(std::false_type{});
  if ((n & 1) && leader) {
    double t = p[n - 1] + (-gamma) * q[n - 1];
    t = beta * t;
    p[n - 1] = t + r[n - 1];
  }
  if (leader) st->resv[parity ^ 1] = res;
}

}  // namespace mk

struct mk_bicgstab {
  int max_iter = 1000;   // solvers.c:83
  double tol = 1e-6;     // solvers.c:84
  int print_level = 0, logging = 0, check_every = 16;
  int64_t n = 0;
  double *r0 = nullptr, *r = nullptr, *p = nullptr, *q = nullptr, *s = nullptr;
  double *slab1 = nullptr, *slab2 = nullptr, *slab3 = nullptr;
  mk::BicgState *state = nullptr;
  double *norms = nullptr;
  int norms_cap = 0;
  long long iterations = 0;
  double rel_res_norm = 0.0;
  int converged = 0, breakdown = 0;
};

using namespace mk;

static void bicg_release(mk_bicgstab *S) {
  dev_free(S->r0); dev_free(S->r); dev_free(S->p); dev_free(S->q); dev_free(S->s);
  dev_free(S->slab1); dev_free(S->slab2); dev_free(S->slab3); dev_free(S->state); dev_free(S->norms);
  S->r0 = S->r = S->p = S->q = S->s = S->slab1 = S->slab2 = S->slab3 = S->norms = nullptr;
  S->state = nullptr;
  S->n = 0;
}

extern "C" {

int mk_bicgstab_create(mk_bicgstab **solver) {
  MK_REQUIRE_INIT();
  if (!solver) return fail(MK_ERR_INVALID, "mk_bicgstab_create: null argument");
  *solver = new mk_bicgstab();
  return MK_OK;
}

#define MK_BICG_SETTER(name, field, type, cond)                                       \
  int name(mk_bicgstab *solver, type v) {                                              \
    if (!solver) return fail(MK_ERR_INVALID, #name ": null solver");                   \
    if (!(cond)) return fail(MK_ERR_INVALID, #name ": value out of range");            \
    solver->field = v;                                                                 \
    return MK_OK;                                                                      \
  }
MK_BICG_SETTER(mk_bicgstab_set_max_iter, max_iter, int, v >= 0)
MK_BICG_SETTER(mk_bicgstab_set_tol, tol, double, v >= 0.0)
MK_BICG_SETTER(mk_bicgstab_set_print_level, print_level, int, v >= 0)
MK_BICG_SETTER(mk_bicgstab_set_logging, logging, int, v >= 0)
MK_BICG_SETTER(mk_bicgstab_set_check_every, check_every, int, v >= 1)
#undef MK_BICG_SETTER

int mk_bicgstab_setup(mk_bicgstab *S, const mk_matrix *A, const mk_vector *b, const mk_vector *x) {
  MK_REQUIRE_INIT();
  if (!S || !A || !b || !x) return fail(MK_ERR_INVALID, "mk_bicgstab_setup: null argument");
  if (A->nrows != A->ncols) return fail(MK_ERR_INVALID, "mk_bicgstab_setup: matrix is %lld x %lld, must be square", (long long)A->nrows, (long long)A->ncols);
  if (b->n != A->nrows || x->n != A->nrows) return fail(MK_ERR_INVALID, "mk_bicgstab_setup: vector length does not match the matrix");
  bicg_release(S);
  S->n = A->nrows;
  MK_TRY(dev_alloc(&S->r0, (size_t)S->n));
  MK_TRY(dev_alloc(&S->r, (size_t)S->n));
  MK_TRY(dev_alloc(&S->p, (size_t)S->n));
  MK_TRY(dev_alloc(&S->q, (size_t)S->n));
  MK_TRY(dev_alloc(&S->s, (size_t)S->n));
  MK_TRY(dev_alloc(&S->slab1, (size_t)2 * kSlabStride));
  MK_TRY(dev_alloc(&S->slab2, (size_t)2 * kSlabStride));
  MK_TRY(dev_alloc(&S->slab3, (size_t)2 * kSlabStride));
  MK_TRY(dev_alloc(&S->state, 1));
  if (S->logging > 0 || S->print_level > 0) {
    S->norms_cap = S->max_iter + 1;
    MK_TRY(dev_alloc(&S->norms, (size_t)S->norms_cap));
  }
  MK_TRY(analyse_matrix(const_cast<mk_matrix *>(A)));
  MK_HIP(hipStreamSynchronize(ctx().stream));
  return MK_OK;
}

int mk_bicgstab_solve(mk_bicgstab *S, const mk_matrix *A, const mk_vector *b, mk_vector *x) {
  MK_REQUIRE_INIT();
  if (!S || !A || !b || !x) return fail(MK_ERR_INVALID, "mk_bicgstab_solve: null argument");
  if (S->n != A->nrows || !S->state) return fail(MK_ERR_INVALID, "mk_bicgstab_solve: call mk_bicgstab_setup with this matrix first");
  if (b->n != S->n || x->n != S->n) return fail(MK_ERR_INVALID, "mk_bicgstab_solve: vector length does not match the matrix");
  if (S->norms && S->norms_cap < S->max_iter + 1) return fail(MK_ERR_INVALID, "mk_bicgstab_solve: max_iter was raised after setup");
  Context &c = ctx();
  const int64_t n = S->n;
  S->iterations = 0; S->rel_res_norm = 0.0; S->converged = 0; S->breakdown = 0;
  if (n == 0) return MK_OK;
  const int vb = vec_blocks(n);
  const int nt = (int)nt_rows(n);   // nontemporal vector streams once they outgrow the Infinity Cache
  const dim3 grid(vb), block(kVecThreads);
  BicgState *hst = static_cast<BicgState *>(c.pinned);

  hipLaunchKernelGGL(copy_kernel, grid, dim3(256), 0, c.stream, S->r0, b->d, n);
  MK_TRY(launch_spmv(A, -1.0, x->d, 1.0, S->r0));
  hipLaunchKernelGGL(bicg_init_vectors_kernel, grid, block, 0, c.stream, n, b->d, S->r0, S->r, S->p, S->slab1);
  hipLaunchKernelGGL(bicg_init_scalars_kernel, dim3(1), block, 0, c.stream, S->slab1, vb, kSlabStride, S->tol, S->state, S->norms);

  bool stop = false;
  for (int it = 0; it < S->max_iter && !stop; it++) {
    int *done = &S->state->done, *skip = &S->state->skip_true;
    MK_TRY(launch_spmv(A, 1.0, S->p, 0.0, S->q, done));
    hipLaunchKernelGGL(dot2_kernel, grid, block, 0, c.stream, n, S->r0, S->q, nullptr, nullptr, S->slab1, done, nt);
    hipLaunchKernelGGL(bicg_step1_kernel, grid, block, 0, c.stream, n, S->p, S->q, x->d, S->r, S->slab1, vb, S->state, nt, it & 1);
    MK_TRY(launch_spmv(A, 1.0, S->r, 0.0, S->s, done));
    hipLaunchKernelGGL(dot2_kernel, grid, block, 0, c.stream, n, S->r, S->s, S->s, S->s, S->slab2, done, nt);
    hipLaunchKernelGGL(bicg_step2_kernel, grid, block, 0, c.stream, n, S->s, S->r0, x->d, S->r, S->slab2, vb, kSlabStride, S->slab3, S->state, nt);
    hipLaunchKernelGGL(bicg_norm_kernel, dim3(1), block, 0, c.stream, S->slab3 + kSlabStride, vb, S->state, S->norms);
    // only when the recursive residual passed: r = b - A x (skip_true is 1 otherwise, and after a stop)
    hipLaunchKernelGGL(copy_unless_kernel, grid, dim3(256), 0, c.stream, S->r, b->d, n, skip);
    MK_TRY(launch_spmv(A, -1.0, x->d, 1.0, S->r, skip));
    hipLaunchKernelGGL(dot2_kernel, grid, block, 0, c.stream, n, S->r0, S->r, S->r, S->r, S->slab3, skip, nt);   // only behind a true residual
    hipLaunchKernelGGL(bicg_step3_kernel, grid, block, 0, c.stream, n, S->r, S->q, S->p, S->slab3, vb, kSlabStride, S->state, nt, it & 1);
    if ((it + 1) % S->check_every == 0) {
      MK_HIP(hipMemcpyAsync(&hst[0], S->state, sizeof(BicgState), hipMemcpyDeviceToHost, c.stream));
      MK_HIP(hipStreamSynchronize(c.stream));
      if (hst[0].done != 0) stop = true;
    }
  }
  MK_HIP(hipGetLastError());
  MK_HIP(hipMemcpyAsync(&hst[0], S->state, sizeof(BicgState), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  S->iterations = hst[0].iter;
  S->converged = hst[0].converged;
  S->breakdown = hst[0].breakdown;
  S->rel_res_norm = hst[0].b_norm > 0.0 ? hst[0].r_norm / hst[0].b_norm : hst[0].r_norm;
  if (S->print_level > 1)
    printf("BiCGSTAB: %lld iterations, relative residual %e%s\n", S->iterations, S->rel_res_norm, S->converged ? "" : " (not converged)");
  return MK_OK;
}

int mk_bicgstab_get_num_iterations(const mk_bicgstab *S, long long *iterations) {
  if (!S || !iterations) return fail(MK_ERR_INVALID, "mk_bicgstab_get_num_iterations: null argument");
  *iterations = S->iterations;
  return MK_OK;
}

int mk_bicgstab_get_final_relative_residual_norm(const mk_bicgstab *S, double *norm) {
  if (!S || !norm) return fail(MK_ERR_INVALID, "mk_bicgstab_get_final_relative_residual_norm: null argument");
  *norm = S->rel_res_norm;
  return MK_OK;
}

int mk_bicgstab_get_converged(const mk_bicgstab *S, int *converged, int *breakdown) {
  if (!S) return fail(MK_ERR_INVALID, "mk_bicgstab_get_converged: null solver");
  if (converged) *converged = S->converged;
  if (breakdown) *breakdown = S->breakdown;
  return MK_OK;
}

int mk_bicgstab_get_residual_history(const mk_bicgstab *S, double *norms, int capacity) {
  MK_REQUIRE_INIT();
  if (!S || !norms || capacity < 0) return fail(MK_ERR_INVALID, "mk_bicgstab_get_residual_history: bad argument");
  if (!S->norms) return fail(MK_ERR_INVALID, "mk_bicgstab_get_residual_history: logging was off at setup");
  int count = (int)std::min<long long>(S->iterations + 1, capacity);
  if (S->breakdown == 1) count = (int)std::min<long long>(S->iterations, capacity);   // no norm for the aborted step
  if (count > 0) {
    MK_HIP(hipMemcpyAsync(norms, S->norms, sizeof(double) * count, hipMemcpyDeviceToHost, ctx().stream));
    MK_HIP(hipStreamSynchronize(ctx().stream));
  }
  return count;
}

// ---------------------------------------------------------------------------- row blocks
// HYPRE_ParCSRBiCGSTABSetup + Solve over the row blocks (solvers.c:517-564 with num_procs > 1).  The loop of
// mk_bicgstab_solve with every product done by dist_spmv (halo of the operand beside the diag-block product) and every
// reduction folded to one value per component (finalize_slab_kernel), all-reduced, and read by the next kernel's prologue
// as a one-block slab: five all-reduces per iteration, as many as hypre's own inner products.  The scalars live in one
// device record per rank; they derive from all-reduced values only, so every rank takes the same branches.
int mk_dist_bicgstab_solve(int R, mk_bicgstab **solvers, mk_dist_matrix **A, mk_vector **b, mk_vector **x) {
  MK_REQUIRE_INIT();
  MK_TRY(dist_check_ranks("mk_dist_bicgstab_solve", R, A));
  if (!solvers || !b || !x) return fail(MK_ERR_INVALID, "mk_dist_bicgstab_solve: null argument");
  Context &c = ctx();
  mk_bicgstab *S0 = solvers[0];
  for (int r = 0; r < R; r++) {
    if (!solvers[r] || !b[r] || !x[r]) return fail(MK_ERR_INVALID, "mk_dist_bicgstab_solve: null argument");
    if (b[r]->n != A[r]->n_local || x[r]->n != A[r]->n_local || x[r]->cap < A[r]->n_local + A[r]->n_ghost)
      return fail(MK_ERR_INVALID, "mk_dist_bicgstab_solve: vectors must come from mk_dist_vector_create of the same matrix");
    if (solvers[r]->max_iter != S0->max_iter || solvers[r]->tol != S0->tol || solvers[r]->check_every != S0->check_every)
      return fail(MK_ERR_INVALID, "mk_dist_bicgstab_solve: ranks disagree on solver settings");
  }
  const bool want_norms = S0->logging > 0 || S0->print_level > 0;
  struct Work {
    double *r0 = nullptr, *r = nullptr, *p = nullptr, *q = nullptr, *s = nullptr, *slab1 = nullptr, *slab2 = nullptr, *slab3 = nullptr;
    double *buf = nullptr, *norms = nullptr;
    BicgState *state = nullptr;
    void release() {
      dev_free(r0); dev_free(r); dev_free(p); dev_free(q); dev_free(s); dev_free(slab1); dev_free(slab2); dev_free(slab3);
      dev_free(buf); dev_free(norms); dev_free(state);
    }
  };
  std::vector<Work> W(R);
  std::vector<double *> pv(R), rv(R), qv(R), sv(R), xv(R), bufs(R);
  std::vector<const int *> done(R), skip(R);
  std::vector<int> vb(R), nt(R);
  int rc = MK_OK;
  auto cleanup = [&]() { (void)hipDeviceSynchronize(); for (auto &w : W) w.release(); };
#define MK_D(expr) do { rc = (expr); if (rc != MK_OK) { cleanup(); return rc; } } while (0)
#define MK_DH(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(MK_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); } } while (0)
  for (int r = 0; r < R; r++) {
    const size_t nl = (size_t)A[r]->n_local, ne = (size_t)(A[r]->n_local + A[r]->n_ghost);
    MK_D(dev_alloc(&W[r].r0, nl));
    MK_D(dev_alloc(&W[r].r, ne));    // operands of a product carry their ghosts behind the owned rows
    MK_D(dev_alloc(&W[r].p, ne));
    MK_D(dev_alloc(&W[r].q, nl));
    MK_D(dev_alloc(&W[r].s, nl));
    MK_D(dev_alloc(&W[r].slab1, (size_t)2 * kSlabStride));
    MK_D(dev_alloc(&W[r].slab2, (size_t)2 * kSlabStride));
    MK_D(dev_alloc(&W[r].slab3, (size_t)2 * kSlabStride));
    MK_D(dev_alloc(&W[r].buf, 8));
    MK_D(dev_alloc(&W[r].state, 1));
    if (want_norms) MK_D(dev_alloc(&W[r].norms, (size_t)S0->max_iter + 1));
    MK_D(analyse_matrix(&A[r]->diag));
    pv[r] = W[r].p; rv[r] = W[r].r; qv[r] = W[r].q; sv[r] = W[r].s; xv[r] = x[r]->d; bufs[r] = W[r].buf;
    done[r] = &W[r].state->done; skip[r] = &W[r].state->skip_true;
    vb[r] = vec_blocks(A[r]->n_local);
    nt[r] = (int)nt_rows(A[r]->n_local);
    solvers[r]->iterations = 0; solvers[r]->rel_res_norm = 0.0; solvers[r]->converged = 0; solvers[r]->breakdown = 0;
  }
  const dim3 block(kVecThreads);
  // fold the two components of a slab and all-reduce them into buf[0..1] of every rank
  auto reduce2 = [&](auto slab_of) -> int {
    for (int r = 0; r < R; r++)
      hipLaunchKernelGGL((finalize_slab_kernel<2>), dim3(1), block, 0, c.stream, slab_of(r), vb[r], W[r].buf);
    return dist_all_reduce(R, A, bufs.data(), 2);
  };
  BicgState *hst = static_cast<BicgState *>(c.pinned);

  // r0 = b - A x;  r = p = r0;  <b,b>, <r0,r0>
  {
    std::vector<double *> r0v(R);
    for (int r = 0; r < R; r++) {
      r0v[r] = W[r].r0;
      if (A[r]->n_local) hipLaunchKernelGGL(copy_kernel, dim3(vb[r]), dim3(256), 0, c.stream, W[r].r0, b[r]->d, A[r]->n_local);
    }
    MK_D(dist_spmv(R, A, xv.data(), r0v.data(), -1.0, 1.0, nullptr));
  }
  for (int r = 0; r < R; r++)
    hipLaunchKernelGGL(bicg_init_vectors_kernel, dim3(vb[r]), block, 0, c.stream, A[r]->n_local, b[r]->d, W[r].r0, W[r].r, W[r].p, W[r].slab1);
  MK_D(reduce2([&](int r) { return W[r].slab1; }));
  for (int r = 0; r < R; r++)
    hipLaunchKernelGGL(bicg_init_scalars_kernel, dim3(1), block, 0, c.stream, W[r].buf, 1, 1, S0->tol, W[r].state, W[r].norms);

  bool stop = false;
  for (int it = 0; it < S0->max_iter && !stop; it++) {
    MK_D(dist_spmv(R, A, pv.data(), qv.data(), 1.0, 0.0, done.data()));                  // q = A p
    for (int r = 0; r < R; r++)
      hipLaunchKernelGGL(dot2_kernel, dim3(vb[r]), block, 0, c.stream, A[r]->n_local, W[r].r0, W[r].q, (const double *)nullptr, (const double *)nullptr,
                         W[r].slab1, done[r], nt[r]);
    MK_D(reduce2([&](int r) { return W[r].slab1; }));                                    // <r0,q>
    for (int r = 0; r < R; r++)
      hipLaunchKernelGGL(bicg_step1_kernel, dim3(vb[r]), block, 0, c.stream, A[r]->n_local, W[r].p, W[r].q, x[r]->d, W[r].r, W[r].buf, 1, W[r].state,
                         nt[r], it & 1);
    MK_D(dist_spmv(R, A, rv.data(), sv.data(), 1.0, 0.0, done.data()));                  // s = A r
    for (int r = 0; r < R; r++)
      hipLaunchKernelGGL(dot2_kernel, dim3(vb[r]), block, 0, c.stream, A[r]->n_local, W[r].r, W[r].s, W[r].s, W[r].s, W[r].slab2, done[r], nt[r]);
    MK_D(reduce2([&](int r) { return W[r].slab2; }));                                    // <r,s>, <s,s>
    for (int r = 0; r < R; r++)
      hipLaunchKernelGGL(bicg_step2_kernel, dim3(vb[r]), block, 0, c.stream, A[r]->n_local, W[r].s, W[r].r0, x[r]->d, W[r].r, W[r].buf, 1, 1,
                         W[r].slab3, W[r].state, nt[r]);
    MK_D(reduce2([&](int r) { return W[r].slab3; }));                                    // <r0,r>, <r,r> of the recursive residual
    for (int r = 0; r < R; r++)
      hipLaunchKernelGGL(bicg_norm_kernel, dim3(1), block, 0, c.stream, W[r].buf + 1, 1, W[r].state, W[r].norms);
    // only when the recursive residual passed: r = b - A x and its two dots (slab3 keeps step 2's partials otherwise,
    // and the fold + all-reduce below gives the same two values again: the host cannot know which it is without a wait)
    for (int r = 0; r < R; r++)
      if (A[r]->n_local) hipLaunchKernelGGL(copy_unless_kernel, dim3(vb[r]), dim3(256), 0, c.stream, W[r].r, b[r]->d, A[r]->n_local, skip[r]);
    MK_D(dist_spmv(R, A, xv.data(), rv.data(), -1.0, 1.0, skip.data()));
    for (int r = 0; r < R; r++)
      hipLaunchKernelGGL(dot2_kernel, dim3(vb[r]), block, 0, c.stream, A[r]->n_local, W[r].r0, W[r].r, W[r].r, W[r].r, W[r].slab3, skip[r], nt[r]);
    MK_D(reduce2([&](int r) { return W[r].slab3; }));
    for (int r = 0; r < R; r++)
      hipLaunchKernelGGL(bicg_step3_kernel, dim3(vb[r]), block, 0, c.stream, A[r]->n_local, W[r].r, W[r].q, W[r].p, W[r].buf, 1, 1, W[r].state, nt[r],
                         it & 1);
    if ((it + 1) % S0->check_every == 0) {
      MK_DH(hipMemcpyAsync(&hst[0], W[0].state, sizeof(BicgState), hipMemcpyDeviceToHost, c.stream));
      MK_DH(hipStreamSynchronize(c.stream));
      if (hst[0].done != 0) stop = true;   // the same on every rank
    }
  }
  MK_DH(hipGetLastError());
  MK_DH(hipMemcpyAsync(&hst[0], W[0].state, sizeof(BicgState), hipMemcpyDeviceToHost, c.stream));
  MK_DH(hipDeviceSynchronize());
  for (int r = 0; r < R; r++) {
    mk_bicgstab *S = solvers[r];
    S->iterations = hst[0].iter;
    S->converged = hst[0].converged;
    S->breakdown = hst[0].breakdown;
    S->rel_res_norm = hst[0].b_norm > 0.0 ? hst[0].r_norm / hst[0].b_norm : hst[0].r_norm;
    if (want_norms) {   // the history goes to the solver object: mk_bicgstab_get_residual_history reads it there
      dev_free(S->norms);
      S->norms = W[r].norms;
      S->norms_cap = S0->max_iter + 1;
      W[r].norms = nullptr;
    }
  }
  if (S0->print_level > 1 && (!A[0]->comm || A[0]->rank == 0))
    printf("BiCGSTAB: %lld iterations, relative residual %e%s\n", S0->iterations, S0->rel_res_norm, S0->converged ? "" : " (not converged)");
  cleanup();
#undef MK_D
#undef MK_DH
  return MK_OK;
}

int mk_bicgstab_destroy(mk_bicgstab *S) {
  if (!S) return MK_OK;
  if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
  bicg_release(S);
  delete S;
  return MK_OK;
}

}  // extern "C"

// =============================================================================== CGNR
// Stands in for HYPRE_ParCSRCGNRSetup/Solve as the reference drives them for solver id 40
// (solvers.c:864-906: MaxIter, Tol, Logging, identity preconditioner): CG on A^T A x = A^T b following the
// published hypre_CGNRSolve loop (oracle/oracle.c orc_cgnr lists the steps).  PARITY UNPINNED.
// A^T is built once at setup by the device COO -> CSR conversion and analysed like any other matrix (the
// transpose of a stencil is a stencil, so it gets the same kernel).  Per iteration: q = A p, t = A^T r, four
// vector kernels, and the two conditional kernels that recompute r = b - A x when <r,r> passes the test.
// Algorithmic bytes per iteration: 2 SpMV + 112 n.
namespace mk {

struct CgnrState {
  double gamma, gamma_old, i_prod, bi_prod, eps;
  long long iter;
  int done;        // 0 running, 1 converged
  int converged;
  int skip_true;
  int pad;
};

// partials of <b,b>, <r,r>
static __global__ __launch_bounds__(kVecThreads) void cgnr_init_dots_kernel(int64_t n, const double *__restrict__ b, const double *__restrict__ r,
                                                                            double *__restrict__ slab) {
  double v[2] = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kVecThreads) {
    v[0] += b[i] * b[i];
    v[1] += r[i] * r[i];
  }
  __shared__ double smem[2 * (kVecThreads / 64)];
  block_sum<2, kVecThreads>(v, smem);
  if (threadIdx.x == 0) {
    slab[blockIdx.x] = v[0];
    slab[kSlabStride + blockIdx.x] = v[1];
  }
}

// p = t, partial of <t,t>
static __global__ __launch_bounds__(kVecThreads) void cgnr_init_p_kernel(int64_t n, const double *__restrict__ t, double *__restrict__ p,
                                                                         double *__restrict__ slab) {
  double v[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kVecThreads) {
    const double ti = t[i];
    p[i] = ti;
    v[0] += ti * ti;
  }
  __shared__ double smem[kVecThreads / 64];
  block_sum<1, kVecThreads>(v, smem);
  if (threadIdx.x == 0) slab[blockIdx.x] = v[0];
}

static __global__ __launch_bounds__(kVecThreads) void cgnr_init_scalars_kernel(const double *slab_bb_rr, const double *slab_tt, int nblocks, int stride,
                                                                               double tol, CgnrState *st, double *norms) {
  __shared__ double smem[2 * (kVecThreads / 64)];
  double v[2], g[1];
  reduce_slab<2, kVecThreads>(slab_bb_rr, nblocks, stride, v, smem);
  reduce_slab<1, kVecThreads>(slab_tt, nblocks, kSlabStride, g, smem);
  if (threadIdx.x == 0) {
    CgnrState s;
    memset(&s, 0, sizeof(s));
    s.bi_prod = v[0];
    s.i_prod = v[1];
    s.eps = tol * tol * v[0];
    s.gamma = g[0];
    s.skip_true = 1;
    *st = s;
    if (norms) norms[0] = sqrt(v[1] / v[0]);
  }
}

// alpha = gamma / <q,q>;  x += alpha p;  r -= alpha q;  partial of <r,r>
static __global__ __launch_bounds__(kVecThreads) void cgnr_step1_kernel(int64_t n, const double *__restrict__ p, const double *__restrict__ q,
                                                                        double *__restrict__ x, double *__restrict__ r,
                                                                        const double *__restrict__ slab_in, int nblocks_in,
                                                                        double *__restrict__ slab_out, CgnrState *st, int nt) {
  if (st->done != 0) return;
  __shared__ double smem[kVecThreads / 64];
  double d[1];
  reduce_slab<1, kVecThreads>(slab_in, nblocks_in, kSlabStride, d, smem);
  const double alpha = st->gamma / d[0];
  double v[1] = {0.0};
  const int64_t n2 = n >> 1;
  auto rows = [&](auto tag) {
    constexpr bool NT = decltype(tag)::value;
    for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kVecThreads) {
      const double2 pv = vec_ld<NT>(reinterpret_cast<const double2 *>(p) + i), qv = vec_ld<NT>(reinterpret_cast<const double2 *>(q) + i);
      double2 xv = vec_ld<NT>(reinterpret_cast<const double2 *>(x) + i), rv = vec_ld<NT>(reinterpret_cast<const double2 *>(r) + i);
      xv.x += alpha * pv.x; xv.y += alpha * pv.y;
      rv.x -= alpha * qv.x; rv.y -= alpha * qv.y;
      vec_st<NT>(reinterpret_cast<double2 *>(x) + i, xv);
      vec_st<NT>(reinterpret_cast<double2 *>(r) + i, rv);
      v[0] += rv.x * rv.x + rv.y * rv.y;
    }
  };
  if (nt) rows(std::true_type{}); else rows(std::false_type{});
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    x[n - 1] += alpha * p[n - 1];
    double ri = r[n - 1];
    ri -= alpha * q[n - 1];
    r[n - 1] = ri;
    v[0] += ri * ri;
  }
  block_sum<1, kVecThreads>(v, smem);
  if (threadIdx.x == 0) slab_out[blockIdx.x] = v[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) st->iter += 1;
}

// slab_rr: <r,r> of the recursive residual; slab_tt: <t,t>.  Logs the norm, raises the candidate flag
static __global__ __launch_bounds__(kVecThreads) void cgnr_norm_kernel(const double *slab_rr, const double *slab_tt, int nblocks, CgnrState *st,
                                                                       double *norms) {
  if (st->done != 0) {   // one block: see bicg_norm_kernel
    if (threadIdx.x == 0) st->skip_true = 1;
    return;
  }
  __shared__ double smem[kVecThreads / 64];
  double v[1], g[1];
  reduce_slab<1, kVecThreads>(slab_rr, nblocks, kSlabStride, v, smem);
  reduce_slab<1, kVecThreads>(slab_tt, nblocks, kSlabStride, g, smem);
  if (threadIdx.x == 0) {
    st->i_prod = v[0];
    st->gamma_old = st->gamma;
    st->gamma = g[0];
    if (norms) norms[st->iter] = sqrt(v[0] / st->bi_prod);
    st->skip_true = (v[0] < st->eps) ? 0 : 1;
  }
}

// slab_in: <r,r> of r as it stands now (b - A x if it was a candidate).  beta = gamma/gamma_old; p = beta p + t
static __global__ __launch_bounds__(kVecThreads) void cgnr_step2_kernel(int64_t n, const double *__restrict__ t, double *__restrict__ p,
                                                                        const double *__restrict__ slab_in, int nblocks_in, CgnrState *st, int nt) {
  if (block_flag_raised(&st->done)) return;
  __shared__ double smem[kVecThreads / 64];
  const bool leader = (blockIdx.x == 0 && threadIdx.x == 0);
  const bool cand = (st->skip_true == 0);
  const double gamma = st->gamma, gamma_old = st->gamma_old, eps = st->eps;
  if (cand) {
    double d[1];
    reduce_slab<1, kVecThreads>(slab_in, nblocks_in, kSlabStride, d, smem);
    if (leader) st->i_prod = d[0];
    if (d[0] < eps) {
      if (leader) { st->converged = 1; st->done = 1; }   // skip_true is read at entry by every block: not written here
      return;
    }
  }
  const double beta = gamma / gamma_old;
  const int64_t n2 = n >> 1;   // hypre scales p, then adds t: two roundings
  auto rows = [&](auto tag) {
    constexpr bool NT = decltype(tag)::value;
    for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kVecThreads) {
      double2 pv = vec_ld<NT>(reinterpret_cast<const double2 *>(p) + i);
      const double2 tv = vec_ld<NT>(reinterpret_cast<const double2 *>(t) + i);
      pv.x = __dadd_rn(__dmul_rn(beta, pv.x), tv.x);
      pv.y = __dadd_rn(__dmul_rn(beta, pv.y), tv.y);
      vec_st<NT>(reinterpret_cast<double2 *>(p) + i, pv);
    }
  };
  if (nt) rows(std::true_type{}); else rows(std::false_type{});
  if ((n & 1) && leader) p[n - 1] = __dadd_rn(__dmul_rn(beta, p[n - 1]), t[n - 1]);
}

}  // namespace mk

struct mk_cgnr {
  int max_iter = 1000;   // solvers.c:83
  double tol = 1e-6;     // solvers.c:84
  int logging = 0, check_every = 16;
  int64_t n = 0;
  mk_matrix *AT = nullptr;
  double *r = nullptr, *p = nullptr, *q = nullptr, *t = nullptr;
  double *slab1 = nullptr, *slab2 = nullptr, *slab3 = nullptr;
  mk::CgnrState *state = nullptr;
  double *norms = nullptr;
  int norms_cap = 0;
  long long iterations = 0;
  double rel_res_norm = 0.0;
  int converged = 0;
};

static void cgnr_release(mk_cgnr *S) {
  if (S->AT) { mk_matrix_destroy(S->AT); S->AT = nullptr; }
  dev_free(S->r); dev_free(S->p); dev_free(S->q); dev_free(S->t);
  dev_free(S->slab1); dev_free(S->slab2); dev_free(S->slab3); dev_free(S->state); dev_free(S->norms);
  S->r = S->p = S->q = S->t = S->slab1 = S->slab2 = S->slab3 = S->norms = nullptr;
  S->state = nullptr;
  S->n = 0;
}

extern "C" {

int mk_cgnr_create(mk_cgnr **solver) {
  MK_REQUIRE_INIT();
  if (!solver) return fail(MK_ERR_INVALID, "mk_cgnr_create: null argument");
  *solver = new mk_cgnr();
  return MK_OK;
}

#define MK_CGNR_SETTER(name, field, type, cond)                                       \
  int name(mk_cgnr *solver, type v) {                                                  \
    if (!solver) return fail(MK_ERR_INVALID, #name ": null solver");                   \
    if (!(cond)) return fail(MK_ERR_INVALID, #name ": value out of range");            \
    solver->field = v;                                                                 \
    return MK_OK;                                                                      \
  }
MK_CGNR_SETTER(mk_cgnr_set_max_iter, max_iter, int, v >= 0)
MK_CGNR_SETTER(mk_cgnr_set_tol, tol, double, v >= 0.0)
MK_CGNR_SETTER(mk_cgnr_set_logging, logging, int, v >= 0)
MK_CGNR_SETTER(mk_cgnr_set_check_every, check_every, int, v >= 1)
#undef MK_CGNR_SETTER

int mk_cgnr_setup(mk_cgnr *S, const mk_matrix *A, const mk_vector *b, const mk_vector *x) {
  MK_REQUIRE_INIT();
  if (!S || !A || !b || !x) return fail(MK_ERR_INVALID, "mk_cgnr_setup: null argument");
  if (A->nrows != A->ncols) return fail(MK_ERR_INVALID, "mk_cgnr_setup: matrix is %lld x %lld, must be square", (long long)A->nrows, (long long)A->ncols);
  if (b->n != A->nrows || x->n != A->nrows) return fail(MK_ERR_INVALID, "mk_cgnr_setup: vector length does not match the matrix");
  cgnr_release(S);
  S->n = A->nrows;
  MK_TRY(build_transpose(A, &S->AT));
  MK_TRY(dev_alloc(&S->r, (size_t)S->n));
  MK_TRY(dev_alloc(&S->p, (size_t)S->n));
  MK_TRY(dev_alloc(&S->q, (size_t)S->n));
  MK_TRY(dev_alloc(&S->t, (size_t)S->n));
  MK_TRY(dev_alloc(&S->slab1, (size_t)2 * kSlabStride));
  MK_TRY(dev_alloc(&S->slab2, (size_t)2 * kSlabStride));
  MK_TRY(dev_alloc(&S->slab3, (size_t)2 * kSlabStride));
  MK_TRY(dev_alloc(&S->state, 1));
  if (S->logging > 0) {
    S->norms_cap = S->max_iter + 1;
    MK_TRY(dev_alloc(&S->norms, (size_t)S->norms_cap));
  }
  MK_TRY(analyse_matrix(const_cast<mk_matrix *>(A)));
  MK_TRY(analyse_matrix(S->AT));
  MK_HIP(hipStreamSynchronize(ctx().stream));
  return MK_OK;
}

int mk_cgnr_solve(mk_cgnr *S, const mk_matrix *A, const mk_vector *b, mk_vector *x) {
  MK_REQUIRE_INIT();
  if (!S || !A || !b || !x) return fail(MK_ERR_INVALID, "mk_cgnr_solve: null argument");
  if (S->n != A->nrows || !S->state || !S->AT) return fail(MK_ERR_INVALID, "mk_cgnr_solve: call mk_cgnr_setup with this matrix first");
  if (b->n != S->n || x->n != S->n) return fail(MK_ERR_INVALID, "mk_cgnr_solve: vector length does not match the matrix");
  if (S->norms && S->norms_cap < S->max_iter + 1) return fail(MK_ERR_INVALID, "mk_cgnr_solve: max_iter was raised after setup");
  Context &c = ctx();
  const int64_t n = S->n;
  S->iterations = 0; S->rel_res_norm = 0.0; S->converged = 0;
  if (n == 0) return MK_OK;
  const int vb = vec_blocks(n);
  const int nt = (int)nt_rows(n);
  const dim3 grid(vb), block(kVecThreads);
  CgnrState *hst = static_cast<CgnrState *>(c.pinned);

  // r = b - A x;  t = A^T r;  p = t;  gamma = <t,t>
  hipLaunchKernelGGL(copy_kernel, grid, dim3(256), 0, c.stream, S->r, b->d, n);
  MK_TRY(launch_spmv(A, -1.0, x->d, 1.0, S->r));
  hipLaunchKernelGGL(cgnr_init_dots_kernel, grid, block, 0, c.stream, n, b->d, S->r, S->slab1);
  MK_TRY(launch_spmv(S->AT, 1.0, S->r, 0.0, S->t));
  hipLaunchKernelGGL(cgnr_init_p_kernel, grid, block, 0, c.stream, n, S->t, S->p, S->slab2);
  hipLaunchKernelGGL(cgnr_init_scalars_kernel, dim3(1), block, 0, c.stream, S->slab1, S->slab2, vb, kSlabStride, S->tol, S->state, S->norms);
  MK_HIP(hipMemcpyAsync(&hst[0], S->state, sizeof(CgnrState), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  if (hst[0].bi_prod == 0.0) {   // hypre: zero right-hand side -> x = b, no iterations
    hipLaunchKernelGGL(copy_kernel, grid, dim3(256), 0, c.stream, x->d, b->d, n);
    MK_HIP(hipStreamSynchronize(c.stream));
    return MK_OK;
  }

  bool stop = false;
  for (int it = 0; it < S->max_iter && !stop; it++) {
    int *done = &S->state->done, *skip = &S->state->skip_true;
    MK_TRY(launch_spmv(A, 1.0, S->p, 0.0, S->q, done));
    hipLaunchKernelGGL(dot2_kernel, grid, block, 0, c.stream, n, S->q, S->q, nullptr, nullptr, S->slab1, done, nt);
    hipLaunchKernelGGL(cgnr_step1_kernel, grid, block, 0, c.stream, n, S->p, S->q, x->d, S->r, S->slab1, vb, S->slab2, S->state, nt);
    MK_TRY(launch_spmv(S->AT, 1.0, S->r, 0.0, S->t, done));
    hipLaunchKernelGGL(dot2_kernel, grid, block, 0, c.stream, n, S->t, S->t, nullptr, nullptr, S->slab3, done, nt);
    hipLaunchKernelGGL(cgnr_norm_kernel, dim3(1), block, 0, c.stream, S->slab2, S->slab3, vb, S->state, S->norms);
    // only when <r,r> passed: r = b - A x, and its norm
    hipLaunchKernelGGL(copy_unless_kernel, grid, dim3(256), 0, c.stream, S->r, b->d, n, skip);
    MK_TRY(launch_spmv(A, -1.0, x->d, 1.0, S->r, skip));
    hipLaunchKernelGGL(dot2_kernel, grid, block, 0, c.stream, n, S->r, S->r, nullptr, nullptr, S->slab1, skip, nt);
    hipLaunchKernelGGL(cgnr_step2_kernel, grid, block, 0, c.stream, n, S->t, S->p, S->slab1, vb, S->state, nt);
    if ((it + 1) % S->check_every == 0) {
      MK_HIP(hipMemcpyAsync(&hst[0], S->state, sizeof(CgnrState), hipMemcpyDeviceToHost, c.stream));
      MK_HIP(hipStreamSynchronize(c.stream));
      if (hst[0].done != 0) stop = true;
    }
  }
  MK_HIP(hipGetLastError());
  MK_HIP(hipMemcpyAsync(&hst[0], S->state, sizeof(CgnrState), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  S->iterations = hst[0].iter;
  S->converged = hst[0].converged;
  S->rel_res_norm = hst[0].bi_prod > 0.0 ? sqrt(hst[0].i_prod / hst[0].bi_prod) : 0.0;
  return MK_OK;
}

int mk_cgnr_get_num_iterations(const mk_cgnr *S, long long *iterations) {
  if (!S || !iterations) return fail(MK_ERR_INVALID, "mk_cgnr_get_num_iterations: null argument");
  *iterations = S->iterations;
  return MK_OK;
}

int mk_cgnr_get_final_relative_residual_norm(const mk_cgnr *S, double *norm) {
  if (!S || !norm) return fail(MK_ERR_INVALID, "mk_cgnr_get_final_relative_residual_norm: null argument");
  *norm = S->rel_res_norm;
  return MK_OK;
}

int mk_cgnr_get_converged(const mk_cgnr *S, int *converged) {
  if (!S || !converged) return fail(MK_ERR_INVALID, "mk_cgnr_get_converged: null argument");
  *converged = S->converged;
  return MK_OK;
}

int mk_cgnr_get_residual_history(const mk_cgnr *S, double *norms, int capacity) {
  MK_REQUIRE_INIT();
  if (!S || !norms || capacity < 0) return fail(MK_ERR_INVALID, "mk_cgnr_get_residual_history: bad argument");
  if (!S->norms) return fail(MK_ERR_INVALID, "mk_cgnr_get_residual_history: logging was off at setup");
  const int count = (int)std::min<long long>(S->iterations + 1, capacity);
  if (count > 0) {
    MK_HIP(hipMemcpyAsync(norms, S->norms, sizeof(double) * count, hipMemcpyDeviceToHost, ctx().stream));
    MK_HIP(hipStreamSynchronize(ctx().stream));
  }
  return count;
}

// ---------------------------------------------------------------------------- row blocks
// HYPRE_ParCSRCGNRSetup + Solve over the row blocks (solvers.c:864-906 with num_procs > 1): the loop of mk_cgnr_solve with
// q = A p behind a halo exchange, t = A^T r through the reverse exchange (dist_spmv_transpose: every rank's rows contribute
// to rows of its neighbours, the contributions travel back along the plan) and four all-reduces per iteration.
int mk_dist_cgnr_solve(int R, mk_cgnr **solvers, mk_dist_matrix **A, mk_vector **b, mk_vector **x) {
  MK_REQUIRE_INIT();
  MK_TRY(dist_check_ranks("mk_dist_cgnr_solve", R, A));
  if (!solvers || !b || !x) return fail(MK_ERR_INVALID, "mk_dist_cgnr_solve: null argument");
  Context &c = ctx();
  mk_cgnr *S0 = solvers[0];
  for (int r = 0; r < R; r++) {
    if (!solvers[r] || !b[r] || !x[r]) return fail(MK_ERR_INVALID, "mk_dist_cgnr_solve: null argument");
    if (b[r]->n != A[r]->n_local || x[r]->n != A[r]->n_local || x[r]->cap < A[r]->n_local + A[r]->n_ghost)
      return fail(MK_ERR_INVALID, "mk_dist_cgnr_solve: vectors must come from mk_dist_vector_create of the same matrix");
    if (solvers[r]->max_iter != S0->max_iter || solvers[r]->tol != S0->tol || solvers[r]->check_every != S0->check_every)
      return fail(MK_ERR_INVALID, "mk_dist_cgnr_solve: ranks disagree on solver settings");
  }
  const bool want_norms = S0->logging > 0;
  struct Work {
    double *r = nullptr, *p = nullptr, *q = nullptr, *t = nullptr, *slab1 = nullptr, *slab2 = nullptr, *slab3 = nullptr, *buf = nullptr, *norms = nullptr;
    CgnrState *state = nullptr;
    void release() {
      dev_free(r); dev_free(p); dev_free(q); dev_free(t); dev_free(slab1); dev_free(slab2); dev_free(slab3); dev_free(buf); dev_free(norms);
      dev_free(state);
    }
  };
  std::vector<Work> W(R);
  std::vector<double *> pv(R), rv(R), qv(R), tv(R), xv(R), bufs(R);
  std::vector<const int *> done(R), skip(R);
  std::vector<int> vb(R), nt(R);
  std::vector<int64_t> nl(R);
  int rc = MK_OK;
  auto cleanup = [&]() { (void)hipDeviceSynchronize(); for (auto &w : W) w.release(); };
#define MK_D(expr) do { rc = (expr); if (rc != MK_OK) { cleanup(); return rc; } } while (0)
#define MK_DH(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(MK_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); } } while (0)
  for (int r = 0; r < R; r++) {
    nl[r] = A[r]->n_local;
    const size_t n1 = (size_t)std::max<int64_t>(nl[r], 1), ne = (size_t)std::max<int64_t>(nl[r] + A[r]->n_ghost, 1);
    MK_D(dev_alloc(&W[r].r, n1));
    MK_D(dev_alloc(&W[r].p, ne));    // operand of q = A p: ghosts behind the owned rows
    MK_D(dev_alloc(&W[r].q, n1));
    MK_D(dev_alloc(&W[r].t, n1));
    MK_D(dev_alloc(&W[r].slab1, (size_t)2 * kSlabStride));
    MK_D(dev_alloc(&W[r].slab2, (size_t)2 * kSlabStride));
    MK_D(dev_alloc(&W[r].slab3, (size_t)2 * kSlabStride));
    MK_D(dev_alloc(&W[r].buf, 8));
    MK_D(dev_alloc(&W[r].state, 1));
    if (want_norms) MK_D(dev_alloc(&W[r].norms, (size_t)S0->max_iter + 1));
    MK_D(analyse_matrix(&A[r]->diag));
    pv[r] = W[r].p; rv[r] = W[r].r; qv[r] = W[r].q; tv[r] = W[r].t; xv[r] = x[r]->d; bufs[r] = W[r].buf;
    done[r] = &W[r].state->done; skip[r] = &W[r].state->skip_true;
    vb[r] = vec_blocks(nl[r]);
    nt[r] = (int)nt_rows(nl[r]);
    solvers[r]->iterations = 0; solvers[r]->rel_res_norm = 0.0; solvers[r]->converged = 0;
  }
  const dim3 block(kVecThreads);
  auto each = [&](auto f) { for (int r = 0; r < R; r++) f(r); };
  auto fold1 = [&](int r, const double *slab, int at) {
    hipLaunchKernelGGL((finalize_slab_kernel<1>), dim3(1), block, 0, c.stream, slab, vb[r], W[r].buf + at);
  };
  CgnrState *hst = static_cast<CgnrState *>(c.pinned);

  // r = b - A x;  t = A^T r;  p = t;  <b,b>, <r,r>, gamma = <t,t>
  each([&](int r) { if (nl[r]) hipLaunchKernelGGL(copy_kernel, dim3(vb[r]), dim3(256), 0, c.stream, W[r].r, b[r]->d, nl[r]); });
  MK_D(dist_spmv(R, A, xv.data(), rv.data(), -1.0, 1.0, nullptr));
  each([&](int r) { hipLaunchKernelGGL(cgnr_init_dots_kernel, dim3(vb[r]), block, 0, c.stream, nl[r], b[r]->d, W[r].r, W[r].slab1); });
  MK_D(dist_spmv_transpose(R, A, rv.data(), tv.data(), 1.0, 0.0, nullptr));
  each([&](int r) {
    hipLaunchKernelGGL(cgnr_init_p_kernel, dim3(vb[r]), block, 0, c.stream, nl[r], W[r].t, W[r].p, W[r].slab2);
    hipLaunchKernelGGL((finalize_slab_kernel<2>), dim3(1), block, 0, c.stream, W[r].slab1, vb[r], W[r].buf);
    fold1(r, W[r].slab2, 2);
  });
  MK_D(dist_all_reduce(R, A, bufs.data(), 3));
  each([&](int r) {
    hipLaunchKernelGGL(cgnr_init_scalars_kernel, dim3(1), block, 0, c.stream, W[r].buf, W[r].buf + 2, 1, 1, S0->tol, W[r].state, W[r].norms);
  });
  MK_DH(hipMemcpyAsync(&hst[0], W[0].state, sizeof(CgnrState), hipMemcpyDeviceToHost, c.stream));
  MK_DH(hipStreamSynchronize(c.stream));
  const bool zero_rhs = (hst[0].bi_prod == 0.0);
  if (zero_rhs)   // hypre: zero right-hand side -> x = b, no iterations
    each([&](int r) { if (nl[r]) hipLaunchKernelGGL(copy_kernel, dim3(vb[r]), dim3(256), 0, c.stream, x[r]->d, b[r]->d, nl[r]); });

  bool stop = zero_rhs;
  for (int it = 0; it < S0->max_iter && !stop; it++) {
    MK_D(dist_spmv(R, A, pv.data(), qv.data(), 1.0, 0.0, done.data()));                      // q = A p
    each([&](int r) {
      hipLaunchKernelGGL(dot2_kernel, dim3(vb[r]), block, 0, c.stream, nl[r], W[r].q, W[r].q, (const double *)nullptr, (const double *)nullptr, W[r].slab1,
                         done[r], nt[r]);
      fold1(r, W[r].slab1, 0);
    });
    MK_D(dist_all_reduce(R, A, bufs.data(), 1));                                             // <q,q>
    each([&](int r) {
      hipLaunchKernelGGL(cgnr_step1_kernel, dim3(vb[r]), block, 0, c.stream, nl[r], W[r].p, W[r].q, x[r]->d, W[r].r, W[r].buf, 1, W[r].slab2, W[r].state,
                         nt[r]);
    });
    MK_D(dist_spmv_transpose(R, A, rv.data(), tv.data(), 1.0, 0.0, done.data()));            // t = A^T r
    each([&](int r) {
      hipLaunchKernelGGL(dot2_kernel, dim3(vb[r]), block, 0, c.stream, nl[r], W[r].t, W[r].t, (const double *)nullptr, (const double *)nullptr, W[r].slab3,
                         done[r], nt[r]);
      fold1(r, W[r].slab2, 0);
      fold1(r, W[r].slab3, 1);
    });
    MK_D(dist_all_reduce(R, A, bufs.data(), 2));                                             // <r,r>, <t,t>
    each([&](int r) { hipLaunchKernelGGL(cgnr_norm_kernel, dim3(1), block, 0, c.stream, W[r].buf, W[r].buf + 1, 1, W[r].state, W[r].norms); });
    // only when <r,r> passed: r = b - A x and its norm (the fold + all-reduce run either way: the host cannot know without a
    // wait; step 2 reads the value only behind a candidate)
    each([&](int r) { if (nl[r]) hipLaunchKernelGGL(copy_unless_kernel, dim3(vb[r]), dim3(256), 0, c.stream, W[r].r, b[r]->d, nl[r], skip[r]); });
    MK_D(dist_spmv(R, A, xv.data(), rv.data(), -1.0, 1.0, skip.data()));
    each([&](int r) {
      hipLaunchKernelGGL(dot2_kernel, dim3(vb[r]), block, 0, c.stream, nl[r], W[r].r, W[r].r, (const double *)nullptr, (const double *)nullptr, W[r].slab1,
                         skip[r], nt[r]);
      fold1(r, W[r].slab1, 0);
    });
    MK_D(dist_all_reduce(R, A, bufs.data(), 1));
    each([&](int r) { hipLaunchKernelGGL(cgnr_step2_kernel, dim3(vb[r]), block, 0, c.stream, nl[r], W[r].t, W[r].p, W[r].buf, 1, W[r].state, nt[r]); });
    if ((it + 1) % S0->check_every == 0) {
      MK_DH(hipMemcpyAsync(&hst[0], W[0].state, sizeof(CgnrState), hipMemcpyDeviceToHost, c.stream));
      MK_DH(hipStreamSynchronize(c.stream));
      if (hst[0].done != 0) stop = true;
    }
  }
  MK_DH(hipGetLastError());
  MK_DH(hipMemcpyAsync(&hst[0], W[0].state, sizeof(CgnrState), hipMemcpyDeviceToHost, c.stream));
  MK_DH(hipDeviceSynchronize());
  for (int r = 0; r < R; r++) {
    mk_cgnr *S = solvers[r];
    S->iterations = zero_rhs ? 0 : hst[0].iter;
    S->converged = zero_rhs ? 0 : hst[0].converged;
    S->rel_res_norm = hst[0].bi_prod > 0.0 ? sqrt(hst[0].i_prod / hst[0].bi_prod) : 0.0;
    if (want_norms) {
      dev_free(S->norms);
      S->norms = W[r].norms;
      S->norms_cap = S0->max_iter + 1;
      W[r].norms = nullptr;
    }
  }
  cleanup();
#undef MK_D
#undef MK_DH
  return MK_OK;
}

int mk_cgnr_destroy(mk_cgnr *S) {
  if (!S) return MK_OK;
  if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
  cgnr_release(S);
  delete S;
  return MK_OK;
}

}  // extern "C"
