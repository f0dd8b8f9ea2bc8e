// krylov_kernels.h -- device kernels of the fused CG iteration, shared by the single-GPU
// solver (krylov.hip) and the row-block distributed solver (dist.hip).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdio>

#include "mk_internal.h"

#include <type_traits>
#include <vector>
namespace mk {

constexpr int kVecThreads = 256;

// p = z = C r; partials of <b,b> (or <Cb,b>), <r,z>, <r,r>
static __global__ __launch_bounds__(kVecThreads) void pcg_init_vectors_kernel(int64_t n, const double *__restrict__ b,
                                                                       const double *__restrict__ r,
                                                                       const double *__restrict__ dinv,
                                                                       double *__restrict__ p, int two_norm,
                                                                       double *__restrict__ slab) {
  double v[3] = {0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kVecThreads) {
    const double bi = b[i], ri = r[i];
    const double di = dinv ? dinv[i] : 1.0;
    const double zi = dinv ? ri * di : ri;
    p[i] = zi;
    v[0] += two_norm ? bi * bi : (bi * di) * bi;
    v[1] += ri * zi;
    v[2] += ri * ri;
  }
  __shared__ double smem[3 * (kVecThreads / 64)];
  block_sum<3, kVecThreads>(v, smem);
  if (threadIdx.x == 0)
    for (int k = 0; k < 3; k++) slab[k * kSlabStride + blockIdx.x] = v[k];
}

static __global__ __launch_bounds__(kVecThreads) void pcg_init_scalars_kernel(const double *slab, int nblocks, int stride_in, double tol,
                                                                       int two_norm, PcgState *st, double *norms) {
  __shared__ double smem[3 * (kVecThreads / 64)];
  double v[3];
  reduce_slab<3, kVecThreads>(slab, nblocks, stride_in, v, smem);
  if (threadIdx.x == 0) {
    PcgState s;
    memset(&s, 0, sizeof(s));
    s.bi_prod = v[0];
    s.gamma = v[1];
    s.i_prod = two_norm ? v[2] : v[1];
    s.eps = tol * tol;
    s.done = (v[0] > 0.0) ? 0 : 1;  // zero (or NaN) right-hand side: nothing to iterate, the host decides which it was
    *st = s;
    if (norms) norms[0] = (v[0] > 0.0) ? sqrt(s.i_prod / v[0]) : 0.0;
  }
}

// K2 / K3 are pure streams: nothing is read twice in a launch.  With nt != 0 (vectors larger than the Infinity Cache,
// nt_rows()) every access is nontemporal: 512^3 on the stream-kernel path 224.6 -> 231.2 it/s, 384^3 +8 %; below
// ~10 M rows the default policy is 7-13 % faster (the next kernel finds the vectors in the cache).
typedef double mk_d2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 vec_ld(const double2 *p) {
  if (!NT) return *p;
  const mk_d2v v = __builtin_nontemporal_load(reinterpret_cast<const mk_d2v *>(p));
  return make_double2(v.x, v.y);
}
template <bool NT>
__device__ __forceinline__ void vec_st(double2 *p, double2 v) {
  if (!NT) { *p = v; return; }
  mk_d2v w;
  w.x = v.x; w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<mk_d2v *>(p));
}
// which double2 a thread takes: layout 0 = grid-stride (consecutive blocks touch consecutive 4 KB pieces), layout 1 =
// every block streams one contiguous range.  `nt` arguments of the kernels carry NT in bit 0 and the layout above it.
template <typename F>
__device__ __forceinline__ void vec_rows(int64_t n2, int layout, F f) {
  if (layout == 0) {
    for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kVecThreads) f(i);
  } else {
    const int64_t per = ((n2 + gridDim.x - 1) / gridDim.x + kVecThreads - 1) / kVecThreads * kVecThreads;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = min(lo + per, n2);
    if (layout == 1) {
      for (int64_t i = lo + threadIdx.x; i < hi; i += kVecThreads) f(i);
    } else {   // two pieces per trip: the loads of the second are issued before the stores of the first
      int64_t i = lo + threadIdx.x;
      for (; i + kVecThreads < hi; i += 2 * kVecThreads) { f(i); f(i + kVecThreads); }
      if (i < hi) f(i);
    }
  }
}
// Measured, 512^3 (134 M rows): K2 1.32 ms grid-stride, 1.18 ms contiguous ranges, 1.17 ms with two pieces per trip
// (4.9 -> 5.5 TB/s); K3 0.65 -> 0.60 ms.  At 256^3 (16.7 M rows) grid-stride is ahead by 3 %: the switch sits at 50 M rows.
static inline int vec_flags(int64_t n) {
  static const int env_layout = getenv("MK_VEC_LAYOUT") ? atoi(getenv("MK_VEC_LAYOUT")) : -1;
  const int layout = env_layout >= 0 ? env_layout : (n > 50 * 1000 * 1000 ? 2 : 0);
  return (int)nt_rows(n) | (layout << 1);
}
#define MK_VEC_LD(ptr) vec_ld<NT>(ptr)
#define MK_VEC_ST(ptr, v) vec_st<NT>((ptr), (v))

// K2
template <bool PRECOND>
__global__ __launch_bounds__(kVecThreads) void pcg_update_xr_kernel(int64_t n, const double *__restrict__ p,
                                                                    const double *__restrict__ s, const double *__restrict__ dinv,
                                                                    double *__restrict__ x, double *__restrict__ r,
                                                                    const double *__restrict__ slab_in, int nblocks_in, int stride_in,
                                                                    double *__restrict__ slab_out, PcgState *st, int nt) {
  if (block_flag_raised(&st->done)) return;
  __shared__ double smem[3 * (kVecThreads / 64)];
  double d[2];
  reduce_slab<2, kVecThreads>(slab_in, nblocks_in, stride_in, d, smem);
  const double sdotp = d[0], pp = d[1];
  const double gamma = st->gamma;
  const bool leader = (blockIdx.x == 0 && threadIdx.x == 0);
  if (sdotp == 0.0) {
    if (leader) { st->iter += 1; st->breakdown = 1; st->done = 2; }
    return;
  }
  const double alpha = gamma / sdotp;
  if (!(alpha > DBL_MIN)) {
    if (leader) { st->iter += 1; st->breakdown = 2; st->done = 2; }
    return;
  }
  double v[3] = {0.0, 0.0, 0.0};
  const int64_t n2 = n >> 1;
  const double2 *p2 = reinterpret_cast<const double2 *>(p);
  const double2 *s2 = reinterpret_cast<const double2 *>(s);
  const double2 *d2 = reinterpret_cast<const double2 *>(dinv);
  double2 *x2 = reinterpret_cast<double2 *>(x);
  double2 *r2 = reinterpret_cast<double2 *>(r);
  auto rows = [&](auto tag) {
  constexpr bool NT = decltype(tag)::value;
  vec_rows(n2, nt >> 1, [&](int64_t i) {
    const double2 pv = MK_VEC_LD(p2 + i), sv = MK_VEC_LD(s2 + i);
    double2 xv = MK_VEC_LD(x2 + i), rv = MK_VEC_LD(r2 + i);
    xv.x += alpha * pv.x; xv.y += alpha * pv.y;
    rv.x -= alpha * sv.x; rv.y -= alpha * sv.y;
    MK_VEC_ST(x2 + i, xv);
    MK_VEC_ST(r2 + i, rv);
    double zx = rv.x, zy = rv.y;
    if (PRECOND) { const double2 dv = MK_VEC_LD(d2 + i); zx *= dv.x; zy *= dv.y; }
    v[0] += rv.x * zx + rv.y * zy;
    v[1] += rv.x * rv.x + rv.y * rv.y;
    v[2] += xv.x * xv.x + xv.y * xv.y;
  });
  };
  if (nt & 1) rows(std::true_type{}); else rows(std::false_type{});
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    const double xi = x[i] + alpha * p[i], ri = r[i] - alpha * s[i];
    x[i] = xi; r[i] = ri;
    const double zi = PRECOND ? ri * dinv[i] : ri;
    v[0] += ri * zi; v[1] += ri * ri; v[2] += xi * xi;
  }
  block_sum<3, kVecThreads>(v, smem);
  if (threadIdx.x == 0)
    for (int k = 0; k < 3; k++) slab_out[k * kSlabStride + blockIdx.x] = v[k];
  if (leader) { st->alpha = alpha; st->gamma_old = gamma; st->pp = pp; st->sdotp = sdotp; st->iter += 1; }
}

// K2 of the CG loop with the deferred solution update (vsweep matrices): r -= alpha s only.  x += alpha p is left pending
// (state record: x_pending, pbuf) for the window fill of the next product, which has p in registers anyway; <x,x> after
// that update follows from the partials the product kernel left: <x,x> + 2 alpha <x,p> + alpha^2 <p,p>.  slab_in: 4 components
// (<s,p>, <p,p>, <x,x>, <x,p>); slab_out as K2's (<r,z>, <r,r>, <x,x> in block 0's slot).
template <bool PRECOND>
__global__ __launch_bounds__(kVecThreads) void pcg_update_r_kernel(int64_t n, const double *__restrict__ s, const double *__restrict__ dinv,
                                                                   double *__restrict__ r, const double *__restrict__ slab_in, int nblocks_in,
                                                                   int stride_in, double *__restrict__ slab_out, PcgState *st, int pbuf, int nt) {
  if (block_flag_raised(&st->done)) return;
  __shared__ double smem[4 * (kVecThreads / 64)];
  double d[4];
  reduce_slab<4, kVecThreads>(slab_in, nblocks_in, stride_in, d, smem);
  const double sdotp = d[0], pp = d[1];
  const double gamma = st->gamma;
  const bool leader = (blockIdx.x == 0 && threadIdx.x == 0);
  if (sdotp == 0.0) {
    if (leader) { st->iter += 1; st->breakdown = 1; st->done = 2; st->x_pending = 0; }
    return;
  }
  const double alpha = gamma / sdotp;
  if (!(alpha > DBL_MIN)) {
    if (leader) { st->iter += 1; st->breakdown = 2; st->done = 2; st->x_pending = 0; }
    return;
  }
  double v[3] = {0.0, 0.0, 0.0};
  const int64_t n2 = n >> 1;
  const double2 *s2 = reinterpret_cast<const double2 *>(s);
  const double2 *d2 = reinterpret_cast<const double2 *>(dinv);
  double2 *r2 = reinterpret_cast<double2 *>(r);
  auto rows = [&](auto tag) {
  constexpr bool NT = decltype(tag)::value;
  vec_rows(n2, nt >> 1, [&](int64_t i) {
    const double2 sv = MK_VEC_LD(s2 + i);
    double2 rv = MK_VEC_LD(r2 + i);
    rv.x -= alpha * sv.x; rv.y -= alpha * sv.y;
    MK_VEC_ST(r2 + i, rv);
    double zx = rv.x, zy = rv.y;
    if (PRECOND) { const double2 dv = MK_VEC_LD(d2 + i); zx *= dv.x; zy *= dv.y; }
    v[0] += rv.x * zx + rv.y * zy;
    v[1] += rv.x * rv.x + rv.y * rv.y;
  });
  };
  if (nt & 1) rows(std::true_type{}); else rows(std::false_type{});
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    const double ri = r[i] - alpha * s[i];
    r[i] = ri;
    const double zi = PRECOND ? ri * dinv[i] : ri;
    v[0] += ri * zi; v[1] += ri * ri;
  }
  block_sum<3, kVecThreads>(v, smem);
  if (threadIdx.x == 0) {
    slab_out[0 * kSlabStride + blockIdx.x] = v[0];
    slab_out[1 * kSlabStride + blockIdx.x] = v[1];
    // <x,x> once the pending update is in: one number, carried in block 0's slot (the others add zero)
    slab_out[2 * kSlabStride + blockIdx.x] = (blockIdx.x == 0) ? d[2] + 2.0 * alpha * d[3] + alpha * alpha * pp : 0.0;
  }
  if (leader) { st->alpha = alpha; st->gamma_old = gamma; st->pp = pp; st->sdotp = sdotp; st->iter += 1; st->x_pending = 1; st->pbuf = pbuf; }
}

// the pending solution update of the last completed step (see above), applied after the loop; p0 / p1: the two p buffers
static __global__ __launch_bounds__(kVecThreads) void pcg_apply_pending_x_kernel(int64_t n, double *__restrict__ x, const double *__restrict__ p0,
                                                                                 const double *__restrict__ p1, const PcgState *st) {
  if (st->x_pending == 0) return;
  const double alpha = st->alpha;
  const double *p = st->pbuf ? p1 : p0;
  for (int64_t i = (int64_t)blockIdx.x * kVecThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kVecThreads) x[i] += alpha * p[i];
}
static __global__ void pcg_clear_pending_kernel(PcgState *st) { st->x_pending = 0; }

// K3
template <bool PRECOND>
__global__ __launch_bounds__(kVecThreads) void pcg_update_p_kernel(int64_t n, const double *__restrict__ r,
                                                                   const double *__restrict__ dinv, double *__restrict__ p,
                                                                   const double *__restrict__ slab_in, int nblocks_in, int stride_in,
                                                                   int two_norm, int rel_change, PcgState *st,
                                                                   double *__restrict__ norms, int nt) {
  if (block_flag_raised(&st->done)) return;
  __shared__ double smem[3 * (kVecThreads / 64)];
  double d[3];
  reduce_slab<3, kVecThreads>(slab_in, nblocks_in, stride_in, d, smem);
  const double gamma_new = d[0];
  const double i_prod = two_norm ? d[1] : d[0];
  const double xx = d[2];
  const double bi_prod = st->bi_prod, eps = st->eps, alpha = st->alpha, pp = st->pp, gamma_old = st->gamma_old;
  const bool leader = (blockIdx.x == 0 && threadIdx.x == 0);
  bool conv = (i_prod / bi_prod < eps);
  if (conv && rel_change && i_prod > 0.0) {
    const double ratio = alpha * alpha * pp / xx;
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
  const double beta = gamma_new / gamma_old;
  const int64_t n2 = n >> 1;
  const double2 *r2 = reinterpret_cast<const double2 *>(r);
  const double2 *d2 = reinterpret_cast<const double2 *>(dinv);
  double2 *p2 = reinterpret_cast<double2 *>(p);
  auto rows = [&](auto tag) {
  constexpr bool NT = decltype(tag)::value;
  vec_rows(n2, nt >> 1, [&](int64_t i) {
    double2 zv = MK_VEC_LD(r2 + i);
    if (PRECOND) { const double2 dv = MK_VEC_LD(d2 + i); zv.x *= dv.x; zv.y *= dv.y; }
    double2 pv = MK_VEC_LD(p2 + i);
    pv.x = beta * pv.x + zv.x;
    pv.y = beta * pv.y + zv.y;
    MK_VEC_ST(p2 + i, pv);
  });
  };
  if (nt & 1) rows(std::true_type{}); else rows(std::false_type{});
  if ((n & 1) && leader) {
    const int64_t i = n - 1;
    const double zi = PRECOND ? r[i] * dinv[i] : r[i];
    p[i] = beta * p[i] + zi;
  }
}

// hypre_PCGSolve's per-iteration table at print level 2 and above (the reference driver's default, solvers.c:742 with
// prnt_info = 2), written from the residual history after the solve: rel[i] = ||r_i|| / ||b|| in the norm the solve uses
// (two_norm: l2, else the C-norm), bnorm = ||b|| in that norm.  Layout and formats as in the published hypre pcg.c.
static inline void print_pcg_table(int two_norm, const double *rel, long long iterations, double bnorm) {
  printf("\n\n");
  if (two_norm) printf("Iters       ||r||_2     conv.rate  ||r||_2/||b||_2\n");
  else printf("Iters       ||r||_C     conv.rate  ||r||_C/||b||_C\n");
  printf("-----    ------------    ---------  ------------ \n");
  for (long long i = 1; i <= iterations; i++) {
    const double ni = rel[i] * bnorm, np = rel[i - 1] * bnorm;
    printf("% 5d    %e    %f    %e\n", (int)i, ni, ni / np, rel[i]);
  }
  printf("\n\n");
}

static __global__ void copy_kernel(double *__restrict__ dst, const double *__restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}


// sum the per-block partials of NP components into out[0..NP) (one block; feeds an all-reduce)
template <int NP>
static __global__ __launch_bounds__(kVecThreads) void finalize_slab_kernel(const double *slab, int nblocks, double *out) {
  __shared__ double smem[NP * (kVecThreads / 64)];
  double v[NP];
  reduce_slab<NP, kVecThreads>(slab, nblocks, kSlabStride, v, smem);
  if (threadIdx.x == 0)
    for (int k = 0; k < NP; k++) out[k] = v[k];
}

}  // namespace mk

namespace mk {

static int vec_blocks(int64_t n) {
  static const int env_blocks = getenv("MK_VEC_BLOCKS") ? atoi(getenv("MK_VEC_BLOCKS")) : 0;
  if (env_blocks > 0) return std::min(env_blocks, kSlabStride);
  // Every block of the NEXT kernel re-reads all partials of this one (prologue reduce), so the
  // grid grows with the vector: at least 8 double2 per thread, one block per CU at the small end
  // (1M rows: 256 blocks; with 2048 blocks the prologues moved 4x the bytes of the update itself).
  int64_t want = (n / 2 + kVecThreads * 8 - 1) / (kVecThreads * 8);
  int64_t cap = (int64_t)ctx().num_cus * 8;
  if (cap > kSlabStride) cap = kSlabStride;
  want = std::max<int64_t>(want, std::min<int64_t>(ctx().num_cus, (n / 2 + kVecThreads - 1) / kVecThreads));
  return (int)std::max<int64_t>(1, std::min(want, cap));
}

}  // namespace mk

struct mk_pcg {
  int max_iter = 1000;      // solvers.c:83
  double tol = 1e-6;        // solvers.c:84
  int two_norm = 0;         // HYPRE defaults; the driver sets both to 1 (solvers.c:735,737)
  int rel_change = 0;
  int print_level = 0;
  int logging = 0;
  int precond = MK_PRECOND_NONE;
  int check_every = 16;
  int single_reduction = 0;        // row-block solve: ask for the single-reduction step (mk_pcg_set_single_reduction)
  int single_reduction_used = 0;   // ... and whether the last row-block solve took it
  // work space (setup)
  int64_t n = 0;
  double *r = nullptr, *p = nullptr, *s = nullptr, *p2 = nullptr;
  double *slab1 = nullptr, *slab2 = nullptr;
  mk::PcgState *state = nullptr;
  double *norms = nullptr;
  int norms_cap = 0;
  hipEvent_t ev[2] = {nullptr, nullptr};
  // kernel timing (mk_pcg_set_kernel_timing): one HIP event in front of every kernel of iterations 2..kTimedIters+1
  // of a solve and one behind the last; slot k = k-th kernel of a step
  static constexpr int kTimedIters = 32, kTimedSlots = 3;
  int timing = 0;
  std::vector<hipEvent_t> tev;
  double kt_ms[kTimedSlots] = {0.0, 0.0, 0.0};
  long long kt_launches[kTimedSlots] = {0, 0, 0};
  int kt_slots = 0;
  // results
  long long iterations = 0;
  double rel_res_norm = 0.0;
  int converged = 0, breakdown = 0;
};

// ---- plain Jacobi iteration (extension id 70): kernels shared with dist.hip
namespace mk {

struct JacobiState {
  double rr, bb, eps;
  long long iter;
  int done;
  int converged;
};

template <int G>
static __global__ __launch_bounds__(256) void jacobi_step_kernel(int64_t nrows, const int *__restrict__ rowptr,
                                                           const int *__restrict__ col, const double *__restrict__ val,
                                                           const double *__restrict__ b, const double *__restrict__ xold,
                                                           double *__restrict__ xnew, double *__restrict__ slab,
                                                           const JacobiState *st, const double *__restrict__ extra) {
  if (st->done != 0) return;
  constexpr int ROWS = 256 / G;
  const int g = threadIdx.x % G, rl = threadIdx.x / G;
  double acc = 0.0;
  for (int64_t base = (int64_t)blockIdx.x * ROWS; base < nrows; base += (int64_t)gridDim.x * ROWS) {
    const int64_t row = base + rl;
    double sum = 0.0, dv = 0.0;
    int dj = -1;
    if (row < nrows) {
      const int bgn = rowptr[row], end = rowptr[row + 1];
      for (int j = bgn + g; j < end; j += G) {
        const int cj = col[j];
        const double vj = val[j];
        sum += vj * xold[cj];
        if (cj == (int)row) { dj = j; dv = vj; }   // the last stored diagonal entry wins, as in a serial sweep
      }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
      sum += __shfl_down(sum, off, G);
      const int oj = __shfl_down(dj, off, G);
      const double ov = __shfl_down(dv, off, G);
      if (oj > dj) { dj = oj; dv = ov; }
    }
    if (g == 0 && row < nrows) {
      const double res = extra ? (b[row] - sum) - extra[row] : b[row] - sum;   // extra = off-rank part of A x
      acc += res * res;
      xnew[row] = xold[row] + res / dv;
    }
  }
  __shared__ double smem[256 / 64];
  double v[1] = {acc};
  block_sum<1, 256>(v, smem);
  if (threadIdx.x == 0) slab[blockIdx.x] = v[0];
}

static __global__ __launch_bounds__(256) void jacobi_check_kernel(const double *slab, int nblocks, int stride, long long k, JacobiState *st) {
  if (st->done != 0) return;
  __shared__ double smem[256 / 64];
  double v[1];
  reduce_slab<1, 256>(slab, nblocks, stride, v, smem);
  if (threadIdx.x == 0) {
    st->rr = v[0];
    st->iter = k;
    if (v[0] / st->bb < st->eps) { st->converged = 1; st->done = 1; }
  }
}

static __global__ __launch_bounds__(256) void dot_self_kernel(int64_t n, const double *__restrict__ a, double *__restrict__ slab) {
  double v[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v[0] += a[i] * a[i];
  __shared__ double smem[256 / 64];
  block_sum<1, 256>(v, smem);
  if (threadIdx.x == 0) slab[blockIdx.x] = v[0];
}

static __global__ __launch_bounds__(256) void jacobi_init_kernel(const double *slab, int nblocks, int stride, double tol, JacobiState *st) {
  __shared__ double smem[256 / 64];
  double v[1];
  reduce_slab<1, 256>(slab, nblocks, stride, v, smem);
  if (threadIdx.x == 0) {
    JacobiState s;
    memset(&s, 0, sizeof(s));
    s.bb = v[0];
    s.eps = tol * tol;
    s.done = (v[0] > 0.0) ? 0 : 1;
    *st = s;
  }
}

}  // namespace mk


struct mk_jacobi {
  int max_iter = 1000;
  double tol = 1e-6;
  int check_every = 10;
  int64_t n = 0;
  double *xalt = nullptr, *slab = nullptr;
  mk::JacobiState *state = nullptr;
  long long iterations = 0;
  double rel_res_norm = 0.0;
  int converged = 0;
};

