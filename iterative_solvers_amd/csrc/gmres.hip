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
static __global__ __launch_bounds__(kVecThreads) void gmres_norm2_kernel(const double *slab, int nblocks, double *out) {
  __shared__ double smem[2 * (kVecThreads / 64)];
  double v[2];
  reduce_slab<2, kVecThreads>(slab, nblocks, kSlabStride, v, smem);
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
    hipLaunchKernelGGL(gmres_norm2_kernel, dim3(1), block, 0, c.stream, S->slab1, vb, S->hcol);
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

int mk_gmres_destroy(mk_gmres *S) {
  if (!S) return MK_OK;
  if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
  gmres_release(S);
  delete S;
  return MK_OK;
}

}  // extern "C"
