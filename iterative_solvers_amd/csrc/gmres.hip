// gmres.hip -- restarted GMRES on one GPU (solver id 12).
//
// Stands in for HYPRE_ParCSRGMRESSetup/Solve as the reference drives them for solver id 12
// (solvers.c:1183-1245: MaxIter, KDim = restart = 30, Tol, PrintLevel, Logging, identity preconditioner).
// hypre_GMRESSolve is not in /root/reference; the loop follows its published algorithm step by step
// (oracle/oracle.c orc_gmres): Arnoldi with modified Gram-Schmidt, Givens rotations applied to the
// Hessenberg column as it arrives, |rs[i]| as residual estimate, x += sum_j y_j p_j at the end of a
// cycle, the true residual b - A x as the final judge, and the next cycle's residual vector rebuilt
// from the basis with the rotations (no extra product).  PARITY UNPINNED.
//
// Device side: the Gram-Schmidt chain of iteration i is i+1 fused kernels -- kernel j subtracts the
// projection onto p[j-1] (coefficient reduced from the previous kernel's partials in its prologue,
// fixed order, no atomics) and forms the partials of <p[j], p[i]> in the same pass: 32 B per row and
// basis vector instead of 40.  The Hessenberg column (i+1 doubles) then goes to the host, which does
// the rotations exactly as hypre does and decides; one stream synchronisation per iteration, small
// against the 16 (i+1) n bytes the chain moves.  Basis: (k_dim + 1) vectors in one allocation.
// Algorithmic bytes of iteration i of a cycle: SpMV + (32 i + 40) n.
#include <cmath>
#include <vector>

#include "dist_internal.h"
#include "krylov_kernels.h"

namespace mk {

constexpr int kGmresMaxK = 62;

struct GmresCoef { double v[kGmresMaxK + 2]; };

static __global__ __launch_bounds__(kVecThreads) void gmres_dot2_kernel(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                                        double *__restrict__ slab) {
  double v[2] = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kVecThreads) {
    v[0] += a[i] * a[i];
    v[1] += b[i] * b[i];
  }
  __shared__ double smem[2 * (kVecThreads / 64)];
  block_sum<2, kVecThreads>(v, smem);
  if (threadIdx.x == 0) {
    slab[blockIdx.x] = v[0];
    slab[kSlabStride + blockIdx.x] = v[1];
  }
}

// out[0..1] = sqrt of the two reduced components
static __global__ __launch_bounds__(kVecThreads) void gmres_norm2_kernel(const double *slab, int nblocks, int stride, double *out) {
  __shared__ double smem[2 * (kVecThreads / 64)];
  double v[2];
  reduce_slab<2, kVecThreads>(slab, nblocks, stride, v, smem);
  if (threadIdx.x == 0) { out[0] = sqrt(v[0]); out[1] = sqrt(v[1]); }
}

// One link of the modified Gram-Schmidt chain for the new vector pi:
//   pprev != null: h = sum(slab_in) (= <pprev, pi> formed by the previous link); hcol[jprev] = h; pi -= h pprev
//   then partials of <pj, pi> (pj != null) or of <pi, pi> (last link) into slab_out
static __global__ __launch_bounds__(kVecThreads) void gmres_mgs_kernel(int64_t n, double *__restrict__ pi, const double *__restrict__ pprev,
                                                                       const double *__restrict__ pj, const double *__restrict__ slab_in,
                                                                       int nblocks_in, double *__restrict__ slab_out,
                                                                       double *__restrict__ hcol, int jprev, int nt) {
  __shared__ double smem[kVecThreads / 64];
  double h = 0.0;
  if (pprev) {
    double d[1];
    reduce_slab<1, kVecThreads>(slab_in, nblocks_in, kSlabStride, d, smem);
    h = d[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) hcol[jprev] = h;
  }
  const double mh = -h;
  double v[1] = {0.0};
  const int64_t n2 = n >> 1;
  auto rows = [&](auto tag) {
    constexpr bool NT = decltype(tag)::value;
    for (int64_t k = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; k < n2; k += (int64_t)gridDim.x * kVecThreads) {
      double2 w = vec_ld<NT>(reinterpret_cast<const double2 *>(pi) + k);
      if (pprev) {
        const double2 pp = vec_ld<NT>(reinterpret_cast<const double2 *>(pprev) + k);
        w.x += mh * pp.x;   // hypre: axpy(-h, p[j], p[i])
        w.y += mh * pp.y;
        vec_st<NT>(reinterpret_cast<double2 *>(pi) + k, w);
      }
      if (pj) {
        const double2 pjv = vec_ld<NT>(reinterpret_cast<const double2 *>(pj) + k);
        v[0] += pjv.x * w.x + pjv.y * w.y;
      } else {
        v[0] += w.x * w.x + w.y * w.y;
      }
    }
  };
  if (nt) rows(std::true_type{}); else rows(std::false_type{});
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t k = n - 1;
    double w = pi[k];
    if (pprev) {
      w += mh * pprev[k];
      pi[k] = w;
    }
    v[0] += (pj ? pj[k] : w) * w;
  }
  block_sum<1, kVecThreads>(v, smem);
  if (threadIdx.x == 0) slab_out[blockIdx.x] = v[0];
}

// hcol[i] = ||pi|| from the last link's partials
static __global__ __launch_bounds__(kVecThreads) void gmres_hnorm_kernel(const double *slab, int nblocks, double *hcol, int i) {
  __shared__ double smem[kVecThreads / 64];
  double v[1];
  reduce_slab<1, kVecThreads>(slab, nblocks, kSlabStride, v, smem);
  if (threadIdx.x == 0) hcol[i] = sqrt(v[0]);
}

static __global__ void gmres_scale_kernel(int64_t n, double *__restrict__ p, double t) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) p[k] *= t;
}

// x += rs[i-1] p[i-1] + rs[i-2] p[i-2] + ... + rs[0] p[0], summed in hypre's order
static __global__ void gmres_update_x_kernel(int64_t n, const double *__restrict__ basis, int64_t ld, int i, GmresCoef rs,
                                             double *__restrict__ x) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    double w = rs.v[i - 1] * basis[(int64_t)(i - 1) * ld + k];
    for (int j = i - 2; j >= 0; j--) w += rs.v[j] * basis[(int64_t)j * ld + k];
    x[k] += w;
  }
}

// p[0] <- residual vector of the next cycle: rs[0] p[0] + (rs[i] p[i] + rs[i-1] p[i-1] + ... + rs[1] p[1]) in hypre's
// order of operations (its in-place axpy(rs-1, p, p) scalings included)
static __global__ void gmres_residual_kernel(int64_t n, double *__restrict__ basis, int64_t ld, int i, GmresCoef rs) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const double pik = basis[(int64_t)i * ld + k];
    double acc = pik + (rs.v[i] - 1.0) * pik;
    for (int j = i - 1; j > 0; j--) acc += rs.v[j] * basis[(int64_t)j * ld + k];
    const double p0 = basis[k];
    double q = p0 + (rs.v[0] - 1.0) * p0;
    q += acc;
    basis[k] = q;
  }
}

}  // namespace mk

struct mk_gmres {
  int max_iter = 1000;   // solvers.c:83
  double tol = 1e-6;     // solvers.c:84
  int k_dim = 30;        // solvers.c:460
  int print_level = 0, logging = 0;
  int64_t n = 0, ld = 0;
  int k_alloc = 0;
  double *basis = nullptr, *w = nullptr, *slab1 = nullptr, *slab2 = nullptr, *hcol = nullptr;
  std::vector<double> norms;
  long long iterations = 0;
  double rel_res_norm = 0.0;
  int converged = 0;
};

using namespace mk;

static void gmres_release(mk_gmres *S) {
  dev_free(S->basis); dev_free(S->w); dev_free(S->slab1); dev_free(S->slab2); dev_free(S->hcol);
  S->basis = S->w = S->slab1 = S->slab2 = S->hcol = nullptr;
  S->n = 0;
}

extern "C" {

int mk_gmres_create(mk_gmres **solver) {
  MK_REQUIRE_INIT();
  if (!solver) return fail(MK_ERR_INVALID, "mk_gmres_create: null argument");
  *solver = new mk_gmres();
  return MK_OK;
}

#define MK_GMRES_SETTER(name, field, type, cond)                                      \
  int name(mk_gmres *solver, type v) {                                                 \
    if (!solver) return fail(MK_ERR_INVALID, #name ": null solver");                   \
    if (!(cond)) return fail(MK_ERR_INVALID, #name ": value out of range");            \
    solver->field = v;                                                                 \
    return MK_OK;                                                                      \
  }
MK_GMRES_SETTER(mk_gmres_set_max_iter, max_iter, int, v >= 0)
MK_GMRES_SETTER(mk_gmres_set_k_dim, k_dim, int, v >= 1 && v <= kGmresMaxK)
MK_GMRES_SETTER(mk_gmres_set_tol, tol, double, v >= 0.0)
MK_GMRES_SETTER(mk_gmres_set_print_level, print_level, int, v >= 0)
MK_GMRES_SETTER(mk_gmres_set_logging, logging, int, v >= 0)
#undef MK_GMRES_SETTER

int mk_gmres_setup(mk_gmres *S, const mk_matrix *A, const mk_vector *b, const mk_vector *x) {
  MK_REQUIRE_INIT();
  if (!S || !A || !b || !x) return fail(MK_ERR_INVALID, "mk_gmres_setup: null argument");
  if (A->nrows != A->ncols) return fail(MK_ERR_INVALID, "mk_gmres_setup: matrix is %lld x %lld, must be square", (long long)A->nrows, (long long)A->ncols);
  if (b->n != A->nrows || x->n != A->nrows) return fail(MK_ERR_INVALID, "mk_gmres_setup: vector length does not match the matrix");
  gmres_release(S);
  S->n = A->nrows;
  S->ld = (S->n + 31) & ~(int64_t)31;   // every basis vector starts 256-byte aligned
  S->k_alloc = S->k_dim;
  MK_TRY(dev_alloc(&S->basis, (size_t)S->ld * (size_t)(S->k_dim + 1)));
  MK_TRY(dev_alloc(&S->w, (size_t)S->n));
  MK_TRY(dev_alloc(&S->slab1, (size_t)2 * kSlabStride));
  MK_TRY(dev_alloc(&S->slab2, (size_t)2 * kSlabStride));
  MK_TRY(dev_alloc(&S->hcol, (size_t)kGmresMaxK + 2));
  MK_TRY(analyse_matrix(const_cast<mk_matrix *>(A)));
  MK_HIP(hipStreamSynchronize(ctx().stream));
  return MK_OK;
}

int mk_gmres_solve(mk_gmres *S, const mk_matrix *A, const mk_vector *b, mk_vector *x) {
  MK_REQUIRE_INIT();
  if (!S || !A || !b || !x) return fail(MK_ERR_INVALID, "mk_gmres_solve: null argument");
  if (S->n != A->nrows || !S->basis) return fail(MK_ERR_INVALID, "mk_gmres_solve: call mk_gmres_setup with this matrix first");
  if (b->n != S->n || x->n != S->n) return fail(MK_ERR_INVALID, "mk_gmres_solve: vector length does not match the matrix");
  if (S->k_dim != S->k_alloc) return fail(MK_ERR_INVALID, "mk_gmres_solve: k_dim was changed after setup");
  Context &c = ctx();
  const int64_t n = S->n, ld = S->ld;
  const int k_dim = S->k_dim;
  S->iterations = 0; S->rel_res_norm = 0.0; S->converged = 0;
  S->norms.clear();
  if (n == 0) return MK_OK;
  const int vb = vec_blocks(n);
  const int nt = (int)nt_rows(n);   // the basis vectors stream past nontemporally once they outgrow the Infinity Cache
  const dim3 grid(vb), block(kVecThreads);
  double *hpin = static_cast<double *>(c.pinned);   // k_dim + 2 doubles
  auto P = [&](int j) { return S->basis + (int64_t)j * ld; };
  const double epsmac = 1.e-16;
  std::vector<double> rs(k_dim + 1, 0.0), cs(k_dim, 0.0), sn(k_dim, 0.0), hh((size_t)(k_dim + 1) * k_dim, 0.0);
  auto HH = [&](int a, int bb) -> double & { return hh[(size_t)a * k_dim + bb]; };

  // ||r|| of r = b - A x into `dst`, returned on the host together with ||b|| when asked for
  auto residual_into = [&](double *dst, double *r_norm, double *b_norm) -> int {
    hipLaunchKernelGGL(copy_kernel, grid, dim3(256), 0, c.stream, dst, b->d, n);
    MK_TRY(launch_spmv(A, -1.0, x->d, 1.0, dst));
    hipLaunchKernelGGL(gmres_dot2_kernel, grid, block, 0, c.stream, n, dst, b->d, S->slab1);
    hipLaunchKernelGGL(gmres_norm2_kernel, dim3(1), block, 0, c.stream, S->slab1, vb, kSlabStride, S->hcol);
    MK_HIP(hipMemcpyAsync(hpin, S->hcol, 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    MK_HIP(hipStreamSynchronize(c.stream));
    *r_norm = hpin[0];
    if (b_norm) *b_norm = hpin[1];
    return MK_OK;
  };

  double r_norm = 0.0, b_norm = 0.0;
  MK_TRY(residual_into(P(0), &r_norm, &b_norm));
  if (S->logging > 0 || S->print_level > 0) S->norms.push_back(r_norm);
  const double den_norm = b_norm > 0.0 ? b_norm : r_norm;
  const double epsilon = S->tol * den_norm;
  long long iter = 0;
  while (iter < S->max_iter) {
    rs[0] = r_norm;
    if (r_norm == 0.0) break;
    if (r_norm <= epsilon) {   // converged on entry to a cycle: the true residual decides
      MK_TRY(residual_into(S->w, &r_norm, nullptr));
      if (r_norm <= epsilon) { S->converged = 1; break; }
    }
    hipLaunchKernelGGL(gmres_scale_kernel, grid, dim3(256), 0, c.stream, n, P(0), 1.0 / r_norm);
    int i = 0;
    while (i < k_dim && iter < S->max_iter) {
      i++; iter++;
      MK_TRY(launch_spmv(A, 1.0, P(i - 1), 0.0, P(i)));
      // modified Gram-Schmidt chain: link j subtracts the projection onto p[j-1] and starts <p[j], p[i]>
      double *sl_in = S->slab1, *sl_out = S->slab2;
      for (int j = 0; j <= i; j++) {
        std::swap(sl_in, sl_out);
        hipLaunchKernelGGL(gmres_mgs_kernel, grid, block, 0, c.stream, n, P(i), j > 0 ? P(j - 1) : (const double *)nullptr,
                           j < i ? P(j) : (const double *)nullptr, sl_in, vb, sl_out, S->hcol, j - 1, nt);
      }
      hipLaunchKernelGGL(gmres_hnorm_kernel, dim3(1), block, 0, c.stream, sl_out, vb, S->hcol, i);
      MK_HIP(hipMemcpyAsync(hpin, S->hcol, (size_t)(i + 1) * sizeof(double), hipMemcpyDeviceToHost, c.stream));
      MK_HIP(hipStreamSynchronize(c.stream));
      for (int j = 0; j <= i; j++) HH(j, i - 1) = hpin[j];
      double t = HH(i, i - 1);
      if (t != 0.0) hipLaunchKernelGGL(gmres_scale_kernel, grid, dim3(256), 0, c.stream, n, P(i), 1.0 / t);
      for (int j = 1; j < i; j++) {
        t = HH(j - 1, i - 1);
        HH(j - 1, i - 1) = sn[j - 1] * HH(j, i - 1) + cs[j - 1] * t;
        HH(j, i - 1) = -sn[j - 1] * t + cs[j - 1] * HH(j, i - 1);
      }
      t = HH(i, i - 1) * HH(i, i - 1);
      t += HH(i - 1, i - 1) * HH(i - 1, i - 1);
      double gamma = sqrt(t);
      if (gamma == 0.0) gamma = epsmac;
      cs[i - 1] = HH(i - 1, i - 1) / gamma;
      sn[i - 1] = HH(i, i - 1) / gamma;
      rs[i] = -HH(i, i - 1) * rs[i - 1];
      rs[i] /= gamma;
      rs[i - 1] = cs[i - 1] * rs[i - 1];
      HH(i - 1, i - 1) = sn[i - 1] * HH(i, i - 1) + cs[i - 1] * HH(i - 1, i - 1);
      r_norm = fabs(rs[i]);
      if (S->logging > 0 || S->print_level > 0) S->norms.push_back(r_norm);
      if (r_norm <= epsilon) break;
    }
    // triangular solve on the host, x += sum_j y_j p_j on the device
    rs[i - 1] = rs[i - 1] / HH(i - 1, i - 1);
    for (int k = i - 2; k >= 0; k--) {
      double t = 0.0;
      for (int j = k + 1; j < i; j++) t -= HH(k, j) * rs[j];
      t += rs[k];
      rs[k] = t / HH(k, k);
    }
    GmresCoef coef;
    for (int j = 0; j <= i; j++) coef.v[j] = rs[j];
    hipLaunchKernelGGL(gmres_update_x_kernel, grid, dim3(256), 0, c.stream, n, S->basis, ld, i, coef, x->d);
    if (r_norm <= epsilon) {
      MK_TRY(residual_into(S->w, &r_norm, nullptr));
      if (r_norm <= epsilon) { S->converged = 1; break; }
      hipLaunchKernelGGL(copy_kernel, grid, dim3(256), 0, c.stream, P(0), S->w, n);   // false convergence: restart from b - A x
      continue;
    }
    for (int j = i; j > 0; j--) {
      rs[j - 1] = -sn[j - 1] * rs[j];
      rs[j] = cs[j - 1] * rs[j];
    }
    for (int j = 0; j <= i; j++) coef.v[j] = rs[j];
    hipLaunchKernelGGL(gmres_residual_kernel, grid, dim3(256), 0, c.stream, n, S->basis, ld, i, coef);
  }
  MK_HIP(hipGetLastError());
  MK_HIP(hipStreamSynchronize(c.stream));
  S->iterations = iter;
  S->rel_res_norm = b_norm > 0.0 ? r_norm / b_norm : r_norm;
  if (S->print_level > 1)
    printf("GMRES(%d): %lld iterations, relative residual %e%s\n", k_dim, S->iterations, S->rel_res_norm, S->converged ? "" : " (not converged)");
  return MK_OK;
}

int mk_gmres_get_num_iterations(const mk_gmres *S, long long *iterations) {
  if (!S || !iterations) return fail(MK_ERR_INVALID, "mk_gmres_get_num_iterations: null argument");
  *iterations = S->iterations;
  return MK_OK;
}

int mk_gmres_get_final_relative_residual_norm(const mk_gmres *S, double *norm) {
  if (!S || !norm) return fail(MK_ERR_INVALID, "mk_gmres_get_final_relative_residual_norm: null argument");
  *norm = S->rel_res_norm;
  return MK_OK;
}

int mk_gmres_get_converged(const mk_gmres *S, int *converged) {
  if (!S || !converged) return fail(MK_ERR_INVALID, "mk_gmres_get_converged: null argument");
  *converged = S->converged;
  return MK_OK;
}

int mk_gmres_get_residual_history(const mk_gmres *S, double *norms, int capacity) {
  if (!S || !norms || capacity < 0) return fail(MK_ERR_INVALID, "mk_gmres_get_residual_history: bad argument");
  const int count = (int)std::min<size_t>(S->norms.size(), (size_t)capacity);
  for (int k = 0; k < count; k++) norms[k] = S->norms[k];
  return count;
}

// ---------------------------------------------------------------------------- row blocks
// HYPRE_ParCSRGMRESSetup + Solve over the row blocks (solvers.c:1183-1245 with num_procs > 1).  The loop of mk_gmres_solve
// with every product done by dist_spmv and every inner product of the modified Gram-Schmidt chain all-reduced before the
// next link reads it (hypre_ParKrylovInnerProd is an MPI_Allreduce: i + 1 of them in iteration i of a cycle, the same
// count here).  The Hessenberg column is the same on every rank (all-reduced values only), so every rank's host takes the
// same decisions from its own copy; in a local world rank 0's copy drives all of them.
int mk_dist_gmres_solve(int R, mk_gmres **solvers, mk_dist_matrix **A, mk_vector **b, mk_vector **x) {
  MK_REQUIRE_INIT();
  MK_TRY(dist_check_ranks("mk_dist_gmres_solve", R, A));
  if (!solvers || !b || !x) return fail(MK_ERR_INVALID, "mk_dist_gmres_solve: null argument");
  Context &c = ctx();
  mk_gmres *S0 = solvers[0];
  for (int r = 0; r < R; r++) {
    if (!solvers[r] || !b[r] || !x[r]) return fail(MK_ERR_INVALID, "mk_dist_gmres_solve: null argument");
    if (b[r]->n != A[r]->n_local || x[r]->n != A[r]->n_local || x[r]->cap < A[r]->n_local + A[r]->n_ghost)
      return fail(MK_ERR_INVALID, "mk_dist_gmres_solve: vectors must come from mk_dist_vector_create of the same matrix");
    if (solvers[r]->max_iter != S0->max_iter || solvers[r]->tol != S0->tol || solvers[r]->k_dim != S0->k_dim)
      return fail(MK_ERR_INVALID, "mk_dist_gmres_solve: ranks disagree on solver settings");
  }
  const int k_dim = S0->k_dim;
  const bool want_norms = S0->logging > 0 || S0->print_level > 0;
  struct Work {
    double *basis = nullptr, *w = nullptr, *slab1 = nullptr, *slab2 = nullptr, *hcol = nullptr, *buf = nullptr;
    int64_t ld = 0;
    void release() { dev_free(basis); dev_free(w); dev_free(slab1); dev_free(slab2); dev_free(hcol); dev_free(buf); }
  };
  std::vector<Work> W(R);
  std::vector<double *> xv(R), bufs(R), va(R), vo(R);
  std::vector<int> vb(R), nt(R);
  std::vector<int64_t> nl(R);
  int rc = MK_OK;
  auto cleanup = [&]() { (void)hipDeviceSynchronize(); for (auto &w : W) w.release(); };
#define MK_D(expr) do { rc = (expr); if (rc != MK_OK) { cleanup(); return rc; } } while (0)
#define MK_DH(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(MK_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); } } while (0)
  for (int r = 0; r < R; r++) {
    nl[r] = A[r]->n_local;
    W[r].ld = (nl[r] + A[r]->n_ghost + 31) & ~(int64_t)31;   // every basis vector is the operand of a product: ghosts behind the owned rows
    W[r].ld = std::max<int64_t>(W[r].ld, 32);
    MK_D(dev_alloc(&W[r].basis, (size_t)W[r].ld * (size_t)(k_dim + 1)));
    MK_D(dev_alloc(&W[r].w, (size_t)std::max<int64_t>(nl[r], 1)));
    MK_D(dev_alloc(&W[r].slab1, (size_t)2 * kSlabStride));
    MK_D(dev_alloc(&W[r].slab2, (size_t)2 * kSlabStride));
    MK_D(dev_alloc(&W[r].hcol, (size_t)kGmresMaxK + 2));
    MK_D(dev_alloc(&W[r].buf, 8));
    MK_D(analyse_matrix(&A[r]->diag));
    xv[r] = x[r]->d; bufs[r] = W[r].buf;
    vb[r] = vec_blocks(nl[r]);
    nt[r] = (int)nt_rows(nl[r]);
    solvers[r]->iterations = 0; solvers[r]->rel_res_norm = 0.0; solvers[r]->converged = 0;
    solvers[r]->norms.clear();
  }
  const dim3 block(kVecThreads);
  double *hpin = static_cast<double *>(c.pinned);
  auto P = [&](int r, int j) { return W[r].basis + (int64_t)j * W[r].ld; };
  const double epsmac = 1.e-16;
  std::vector<double> rs(k_dim + 1, 0.0), cs(k_dim, 0.0), sn(k_dim, 0.0), hh((size_t)(k_dim + 1) * k_dim, 0.0), norms;
  auto HH = [&](int a, int bb) -> double & { return hh[(size_t)a * k_dim + bb]; };
  auto each = [&](auto f) { for (int r = 0; r < R; r++) f(r); };

  // dst = b - A x on every rank (dst: basis vector 0, or w), ||dst|| and ||b|| all-reduced and brought to the host
  auto residual_into = [&](bool into_w, double *r_norm, double *b_norm) -> int {
    each([&](int r) {
      vo[r] = into_w ? W[r].w : P(r, 0);
      if (nl[r]) hipLaunchKernelGGL(copy_kernel, dim3(vb[r]), dim3(256), 0, c.stream, vo[r], b[r]->d, nl[r]);
    });
    MK_TRY(dist_spmv(R, A, xv.data(), vo.data(), -1.0, 1.0, nullptr));
    each([&](int r) {
      hipLaunchKernelGGL(gmres_dot2_kernel, dim3(vb[r]), block, 0, c.stream, nl[r], vo[r], b[r]->d, W[r].slab1);
      hipLaunchKernelGGL((finalize_slab_kernel<2>), dim3(1), block, 0, c.stream, W[r].slab1, vb[r], W[r].buf);
    });
    MK_TRY(dist_all_reduce(R, A, bufs.data(), 2));
    each([&](int r) { hipLaunchKernelGGL(gmres_norm2_kernel, dim3(1), block, 0, c.stream, W[r].buf, 1, 1, W[r].hcol); });
    MK_HIP(hipMemcpyAsync(hpin, W[0].hcol, 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    MK_HIP(hipStreamSynchronize(c.stream));
    *r_norm = hpin[0];
    if (b_norm) *b_norm = hpin[1];
    return MK_OK;
  };

  double r_norm = 0.0, b_norm = 0.0;
  MK_D(residual_into(false, &r_norm, &b_norm));
  if (want_norms) norms.push_back(r_norm);
  const double den_norm = b_norm > 0.0 ? b_norm : r_norm;
  const double epsilon = S0->tol * den_norm;
  long long iter = 0;
  int converged = 0;
  while (iter < S0->max_iter) {
    rs[0] = r_norm;
    if (r_norm == 0.0) break;
    if (r_norm <= epsilon) {
      MK_D(residual_into(true, &r_norm, nullptr));
      if (r_norm <= epsilon) { converged = 1; break; }
    }
    each([&](int r) { if (nl[r]) hipLaunchKernelGGL(gmres_scale_kernel, dim3(vb[r]), dim3(256), 0, c.stream, nl[r], P(r, 0), 1.0 / r_norm); });
    int i = 0;
    while (i < k_dim && iter < S0->max_iter) {
      i++; iter++;
      each([&](int r) { va[r] = P(r, i - 1); vo[r] = P(r, i); });
      MK_D(dist_spmv(R, A, va.data(), vo.data(), 1.0, 0.0, nullptr));
      // the chain: link j reads the all-reduced <p[j-1], p[i]> from buf, its own partials go to slab1, folded and all-reduced
      for (int j = 0; j <= i; j++) {
        each([&](int r) {
          hipLaunchKernelGGL(gmres_mgs_kernel, dim3(vb[r]), block, 0, c.stream, nl[r], P(r, i), j > 0 ? P(r, j - 1) : (const double *)nullptr,
                             j < i ? P(r, j) : (const double *)nullptr, W[r].buf, 1, W[r].slab1, W[r].hcol, j - 1, nt[r]);
          hipLaunchKernelGGL((finalize_slab_kernel<1>), dim3(1), block, 0, c.stream, W[r].slab1, vb[r], W[r].buf);
        });
        MK_D(dist_all_reduce(R, A, bufs.data(), 1));
      }
      each([&](int r) { hipLaunchKernelGGL(gmres_hnorm_kernel, dim3(1), block, 0, c.stream, W[r].buf, 1, W[r].hcol, i); });
      MK_DH(hipMemcpyAsync(hpin, W[0].hcol, (size_t)(i + 1) * sizeof(double), hipMemcpyDeviceToHost, c.stream));
      MK_DH(hipStreamSynchronize(c.stream));
      for (int j = 0; j <= i; j++) HH(j, i - 1) = hpin[j];
      double t = HH(i, i - 1);
      if (t != 0.0) each([&](int r) { if (nl[r]) hipLaunchKernelGGL(gmres_scale_kernel, dim3(vb[r]), dim3(256), 0, c.stream, nl[r], P(r, i), 1.0 / t); });
      for (int j = 1; j < i; j++) {
        t = HH(j - 1, i - 1);
        HH(j - 1, i - 1) = sn[j - 1] * HH(j, i - 1) + cs[j - 1] * t;
        HH(j, i - 1) = -sn[j - 1] * t + cs[j - 1] * HH(j, i - 1);
      }
      t = HH(i, i - 1) * HH(i, i - 1);
      t += HH(i - 1, i - 1) * HH(i - 1, i - 1);
      double gamma = sqrt(t);
      if (gamma == 0.0) gamma = epsmac;
      cs[i - 1] = HH(i - 1, i - 1) / gamma;
      sn[i - 1] = HH(i, i - 1) / gamma;
      rs[i] = -HH(i, i - 1) * rs[i - 1];
      rs[i] /= gamma;
      rs[i - 1] = cs[i - 1] * rs[i - 1];
      HH(i - 1, i - 1) = sn[i - 1] * HH(i, i - 1) + cs[i - 1] * HH(i - 1, i - 1);
      r_norm = fabs(rs[i]);
      if (want_norms) norms.push_back(r_norm);
      if (r_norm <= epsilon) break;
    }
    rs[i - 1] = rs[i - 1] / HH(i - 1, i - 1);
    for (int k = i - 2; k >= 0; k--) {
      double t = 0.0;
      for (int j = k + 1; j < i; j++) t -= HH(k, j) * rs[j];
      t += rs[k];
      rs[k] = t / HH(k, k);
    }
    GmresCoef coef;
    for (int j = 0; j <= i; j++) coef.v[j] = rs[j];
    each([&](int r) { if (nl[r]) hipLaunchKernelGGL(gmres_update_x_kernel, dim3(vb[r]), dim3(256), 0, c.stream, nl[r], W[r].basis, W[r].ld, i, coef, x[r]->d); });
    if (r_norm <= epsilon) {
      MK_D(residual_into(true, &r_norm, nullptr));
      if (r_norm <= epsilon) { converged = 1; break; }
      each([&](int r) { if (nl[r]) hipLaunchKernelGGL(copy_kernel, dim3(vb[r]), dim3(256), 0, c.stream, P(r, 0), W[r].w, nl[r]); });
      continue;
    }
    for (int j = i; j > 0; j--) {
      rs[j - 1] = -sn[j - 1] * rs[j];
      rs[j] = cs[j - 1] * rs[j];
    }
    for (int j = 0; j <= i; j++) coef.v[j] = rs[j];
    each([&](int r) { if (nl[r]) hipLaunchKernelGGL(gmres_residual_kernel, dim3(vb[r]), dim3(256), 0, c.stream, nl[r], W[r].basis, W[r].ld, i, coef); });
  }
  MK_DH(hipGetLastError());
  MK_DH(hipDeviceSynchronize());
  for (int r = 0; r < R; r++) {
    solvers[r]->iterations = iter;
    solvers[r]->converged = converged;
    solvers[r]->rel_res_norm = b_norm > 0.0 ? r_norm / b_norm : r_norm;
    solvers[r]->norms = norms;
  }
  if (S0->print_level > 1 && (!A[0]->comm || A[0]->rank == 0))
    printf("GMRES(%d): %lld iterations, relative residual %e%s\n", k_dim, S0->iterations, S0->rel_res_norm, S0->converged ? "" : " (not converged)");
  cleanup();
#undef MK_D
#undef MK_DH
  return MK_OK;
}

int mk_gmres_destroy(mk_gmres *S) {
  if (!S) return MK_OK;
  if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
  gmres_release(S);
  delete S;
  return MK_OK;
}

}  // extern "C"
