// krylov.hip -- fused CG (HYPRE-PCG semantics) and plain Jacobi on one GPU.
//
// Stands in for HYPRE_ParCSRPCGSetup/Solve as the reference drives them for solver id 50
// (solvers.c:729-775: MaxIter, Tol, TwoNorm=1, RelChange=1, Logging=1, identity
// preconditioner).  hypre_PCGSolve itself is not in /root/reference; the loop follows its
// published algorithm (see oracle/oracle.c orc_pcg for the step list).  Extension: diagonal
// scaling as preconditioner (HYPRE_ParCSRDiagScale semantics), solver id 51.
//
// One iteration = three kernels and two grid-level reductions, nothing returns to the host:
//   K1  s = A p          + per-block partials of <s,p>, <p,p>                (spmv.hip)
//   K2  alpha = gamma/<s,p>;  x += alpha p;  r -= alpha s;  z = C r (not stored)
//                         + partials of <r,z>, <r,r>, <x,x>
//   K3  beta = gamma'/gamma;  convergence test;  p = C r + beta p
// Each reduction is finished by the NEXT kernel's prologue (every block sums the slab of
// partials in a fixed order), so there is no in-launch inter-workgroup hand-off and the
// result is bitwise reproducible.  Scalars live in a device-side state record; the host
// polls its stop flag every `check_every` iterations, and kernels launched after the flag
// is raised return at once, so the reported iteration count is exact.
// Algorithmic bytes per iteration: 12*nnz + 92*n (identity) / 12*nnz + 108*n (Jacobi).

#include "krylov_kernels.h"


namespace mk {

static void pcg_release(mk_pcg *S) {
  dev_free(S->r); dev_free(S->p); dev_free(S->s); dev_free(S->slab1); dev_free(S->slab2);
  dev_free(S->state); dev_free(S->norms); dev_free(S->p2);
  S->r = S->p = S->s = S->slab1 = S->slab2 = S->norms = S->p2 = nullptr;
  S->state = nullptr;
  for (int k = 0; k < 2; k++)
    if (S->ev[k]) { (void)hipEventDestroy(S->ev[k]); S->ev[k] = nullptr; }
  for (hipEvent_t e : S->tev) (void)hipEventDestroy(e);
  S->tev.clear();
  S->n = 0;
}

}  // namespace mk

using namespace mk;

extern "C" {

int mk_pcg_create(mk_pcg **solver) {
  MK_REQUIRE_INIT();
  if (!solver) return fail(MK_ERR_INVALID, "mk_pcg_create: null argument");
  *solver = new mk_pcg();
  return MK_OK;
}

#define MK_PCG_SETTER(name, field, type, cond)                                        \
  int name(mk_pcg *solver, type v) {                                                   \
    if (!solver) return fail(MK_ERR_INVALID, #name ": null solver");                   \
    if (!(cond)) return fail(MK_ERR_INVALID, #name ": value out of range");            \
    solver->field = v;                                                                 \
    return MK_OK;                                                                      \
  }
MK_PCG_SETTER(mk_pcg_set_max_iter, max_iter, int, v >= 0)
MK_PCG_SETTER(mk_pcg_set_tol, tol, double, v >= 0.0)
MK_PCG_SETTER(mk_pcg_set_two_norm, two_norm, int, v == 0 || v == 1)
MK_PCG_SETTER(mk_pcg_set_rel_change, rel_change, int, v == 0 || v == 1)
MK_PCG_SETTER(mk_pcg_set_print_level, print_level, int, v >= 0)
MK_PCG_SETTER(mk_pcg_set_logging, logging, int, v >= 0)
MK_PCG_SETTER(mk_pcg_set_precond, precond, int, v == MK_PRECOND_NONE || v == MK_PRECOND_JACOBI)
MK_PCG_SETTER(mk_pcg_set_check_every, check_every, int, v >= 1)
MK_PCG_SETTER(mk_pcg_set_kernel_timing, timing, int, v == 0 || v == 1)
MK_PCG_SETTER(mk_pcg_set_single_reduction, single_reduction, int, v == 0 || v == 1)
#undef MK_PCG_SETTER

int mk_pcg_get_single_reduction(const mk_pcg *solver, int *used) {
  if (!solver || !used) return fail(MK_ERR_INVALID, "mk_pcg_get_single_reduction: null argument");
  *used = solver->single_reduction_used;
  return MK_OK;
}

int mk_pcg_setup(mk_pcg *S, const mk_matrix *A, const mk_vector *b, const mk_vector *x) {
  MK_REQUIRE_INIT();
  if (!S || !A || !b || !x) return fail(MK_ERR_INVALID, "mk_pcg_setup: null argument");
  if (A->nrows != A->ncols) return fail(MK_ERR_INVALID, "mk_pcg_setup: matrix is %lld x %lld, must be square", (long long)A->nrows, (long long)A->ncols);
  if (b->n != A->nrows || x->n != A->nrows) return fail(MK_ERR_INVALID, "mk_pcg_setup: vector length does not match the matrix");
  pcg_release(S);
  S->n = A->nrows;
  MK_TRY(dev_alloc(&S->r, (size_t)S->n));
  MK_TRY(dev_alloc(&S->p, (size_t)S->n));
  MK_TRY(dev_alloc(&S->s, (size_t)S->n));
  MK_TRY(dev_alloc(&S->slab1, (size_t)kMaxParts * kSlabStride));
  MK_TRY(dev_alloc(&S->slab2, (size_t)kMaxParts * kSlabStride));
  MK_TRY(dev_alloc(&S->state, 1));
  if (S->logging > 0 || S->print_level > 0) {
    S->norms_cap = S->max_iter + 1;
    MK_TRY(dev_alloc(&S->norms, (size_t)S->norms_cap));
  }
  for (int k = 0; k < 2; k++) MK_HIP(hipEventCreateWithFlags(&S->ev[k], hipEventDisableTiming));
  MK_TRY(analyse_matrix(const_cast<mk_matrix *>(A)));
  if (S->precond == MK_PRECOND_JACOBI) MK_TRY(ensure_dinv(const_cast<mk_matrix *>(A)));
  if (A->kernel == MK_SPMV_VSWEEP) MK_TRY(dev_alloc(&S->p2, (size_t)S->n));   // K3 folded into the product writes p' to the other buffer
  MK_HIP(hipStreamSynchronize(ctx().stream));
  return MK_OK;
}

int mk_pcg_solve(mk_pcg *S, const mk_matrix *A, const mk_vector *b, mk_vector *x) {
  MK_REQUIRE_INIT();
  if (!S || !A || !b || !x) return fail(MK_ERR_INVALID, "mk_pcg_solve: null argument");
  if (S->n != A->nrows || !S->state) return fail(MK_ERR_INVALID, "mk_pcg_solve: call mk_pcg_setup with this matrix first");
  if (b->n != S->n || x->n != S->n) return fail(MK_ERR_INVALID, "mk_pcg_solve: vector length does not match the matrix");
  if (S->norms && S->norms_cap < S->max_iter + 1) return fail(MK_ERR_INVALID, "mk_pcg_solve: max_iter was raised after setup");
  if (S->precond == MK_PRECOND_JACOBI && !A->dinv) return fail(MK_ERR_INVALID, "mk_pcg_solve: preconditioner changed after setup");
  Context &c = ctx();
  const int64_t n = S->n;
  const double *dinv = (S->precond == MK_PRECOND_JACOBI) ? A->dinv : nullptr;
  PcgState *hst = static_cast<PcgState *>(c.pinned);  // two slots
  S->iterations = 0; S->rel_res_norm = 0.0; S->converged = 0; S->breakdown = 0;
  if (n == 0) return MK_OK;
  const int vb = vec_blocks(n);

  // r = b - A x;  p = C r;  gamma = <r,p>;  bi_prod = <b,b>
  hipLaunchKernelGGL(copy_kernel, dim3(vb), dim3(256), 0, c.stream, S->r, b->d, n);
  MK_TRY(launch_spmv(A, -1.0, x->d, 1.0, S->r));
  hipLaunchKernelGGL(pcg_init_vectors_kernel, dim3(vb), dim3(kVecThreads), 0, c.stream, n, b->d, S->r, dinv, S->p, S->two_norm, S->slab2);
  hipLaunchKernelGGL(pcg_init_scalars_kernel, dim3(1), dim3(kVecThreads), 0, c.stream, S->slab2, vb, kSlabStride, S->tol, S->two_norm, S->state, S->norms);
  MK_HIP(hipMemcpyAsync(&hst[0], S->state, sizeof(PcgState), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  if (!std::isfinite(hst[0].bi_prod) || hst[0].bi_prod < 0.0 || !std::isfinite(hst[0].i_prod)) {
    // hypre_PCGSolve's ieee_check: NaN or Inf in b, x0 or A is an error, not a zero right-hand side
    S->breakdown = 4;
    return fail(MK_ERR_INVALID, "mk_pcg_solve: <b,b> = %g, <r,r> = %g: NaN or Inf in the right-hand side, initial guess or matrix", hst[0].bi_prod, hst[0].i_prod);
  }
  if (hst[0].bi_prod == 0.0) {
    // hypre_PCGSolve: zero right-hand side -> x = b, zero iterations, norm 0
    hipLaunchKernelGGL(copy_kernel, dim3(vb), dim3(256), 0, c.stream, x->d, b->d, n);
    MK_HIP(hipStreamSynchronize(c.stream));
    return MK_OK;
  }

  int pending = -1;  // slot with a state copy in flight
  int nchk = 0;
  bool stop = false;
  // Matrices served by the sweep kernel take the fused step: the sweep runs twice (dots of A p; then
  // r -= alpha A p with A p recomputed instead of written and re-read, x += alpha p in the same pass since p is
  // in the LDS window anyway) and K3 is unchanged: 76 B per row and iteration instead of 90 (s is never stored,
  // p is read twice instead of three times).  Same recurrences and stopping rule, other summation order of the
  // inner products.
  static const int no_fused = getenv("MK_NO_FUSED_CG") ? atoi(getenv("MK_NO_FUSED_CG")) : 0;
  const bool fused = (A->kernel == MK_SPMV_SWEEP) && !no_fused;
  // ... and K3 rides in the window fill of the first sweep of the NEXT step (p' = z + beta p computed while the
  // window is loaded, the block's own rows written to the other p buffer; S->s is free for that): the 24 B/row of
  // K3 become 16 B/row more in a sweep that was bound by instruction issue, not bytes.  With the Jacobi
  // preconditioner this needs 1/diag to be one number (every row keeps the single diagonal entry of the pattern).
  double *pcur = S->p, *palt = S->s;
  double zscale = 1.0;
  static const int no_pfuse = getenv("MK_NO_FUSED_K3") ? atoi(getenv("MK_NO_FUSED_K3")) : 0;
  const bool fuse_p = fused && !no_pfuse && (!dinv || sweep_uniform_diag(A, &zscale));
  // Variable-coefficient grids (vsweep kernel): K1 and K2 stay (the values cannot be streamed twice), K3 rides in K1's
  // window fill from the second iteration on: p' = z + beta p goes to the other p buffer while the window is loaded.
  // 12 nnz + 86 n bytes of the CSR model's 12 nnz + 92 n, and one launch less.
  // (rows of more than 8 entries -- one row per lane, SweepArgs.rpl == 1 -- keep K1 + K2 + K3: vsweep.hip launch_vsweep_m)
  const bool vfuse = (A->kernel == MK_SPMV_VSWEEP) && !no_fused && !no_pfuse && S->p2 != nullptr && A->sweep && A->sweep->rpl != 1;
  if (vfuse) palt = S->p2;
  // ... and the solution update rides there as well, one kernel late (x += alpha p while the fill has p in registers):
  // K2 then streams s and r only.  12 nnz + 78 n bytes.  The update of the last completed step is applied after the loop.
  static const int no_defx = getenv("MK_NO_DEFERRED_X") ? atoi(getenv("MK_NO_DEFERRED_X")) : 0;
  // (not with the Jacobi preconditioner: its 1/diag is a third window stream already and the kernel runs out of registers:
  // 256^3 0.511 -> 0.535 ms per iteration; without a preconditioner 512^3 3.58 -> 3.32 ms, 256^3 0.436 -> 0.434)
  const bool defx = vfuse && !no_defx && !dinv;
  // Constant-coefficient sweeps: the solution update is 16 of the step's 68 B per row; x is updated every SECOND step with
  // both steps' terms (sweep MODE 7 / 8): 60 B per row on average.  MK_NO_HALVED_X=1 keeps x += alpha p in every step.
  static const int no_halfx = getenv("MK_NO_HALVED_X") ? atoi(getenv("MK_NO_HALVED_X")) : 0;
  const bool halfx = fuse_p && !no_halfx && S->max_iter > 1;
  int nb2 = 0;
  // kernel timing: an event in front of every kernel of iterations 1 .. kTimedIters (the first iteration runs another
  // first kernel on the fused path) and one behind the last of them
  const int nslots = fused ? (fuse_p ? 2 : 3) : (vfuse ? 2 : 3);
  const int timed_iters = S->timing ? std::min(mk_pcg::kTimedIters, S->max_iter - 1) : 0;
  if (timed_iters > 0 && S->tev.empty()) {
    S->tev.resize((size_t)mk_pcg::kTimedIters * mk_pcg::kTimedSlots + 1, nullptr);
    for (auto &e : S->tev) MK_HIP(hipEventCreate(&e));
  }
  int tev_used = 0;
  auto stamp = [&](int it) { if (it >= 1 && it <= timed_iters) (void)hipEventRecord(S->tev[tev_used++], c.stream); };
  for (int it = 0; it < S->max_iter && !stop; it++) {
    int nb1 = 0;
    if (it == timed_iters + 1 && tev_used > 0 && tev_used == timed_iters * nslots) (void)hipEventRecord(S->tev[tev_used++], c.stream);
    stamp(it);
    if (fused) {
      if (fuse_p && it > 0) {
        MK_TRY(launch_sweep_cg_pdots(A, S->r, pcur, palt, zscale, S->slab2, nb2, kSlabStride, S->slab1, S->state, S->two_norm,
                                     S->rel_change, S->norms, &nb1));
        std::swap(pcur, palt);
      } else {
        MK_TRY(launch_sweep_cg_dots(A, pcur, S->slab1, S->state, &nb1));
      }
      stamp(it);
      MK_TRY(launch_sweep_cg_resid_x(A, pcur, S->r, x->d, dinv, S->slab1, nb1, kSlabStride, S->slab2, S->state, &nb2,
                                     halfx ? 1 + (it & 1) : 0, it & 1, pcur == S->s ? 1 : 0));
      if (!fuse_p) {
        stamp(it);
        if (dinv)
          hipLaunchKernelGGL((pcg_update_p_kernel<true>), dim3(vb), dim3(kVecThreads), 0, c.stream, n, S->r, dinv, pcur, S->slab2, nb2, kSlabStride, S->two_norm, S->rel_change, S->state, S->norms, vec_flags(n));
        else
          hipLaunchKernelGGL((pcg_update_p_kernel<false>), dim3(vb), dim3(kVecThreads), 0, c.stream, n, S->r, dinv, pcur, S->slab2, nb2, kSlabStride, S->two_norm, S->rel_change, S->state, S->norms, vec_flags(n));
      }
    } else if (vfuse) {
      if (it > 0) {
        MK_TRY(launch_vsweep_cg_pdots(A, S->r, pcur, palt, dinv, S->s, S->slab2, vb, kSlabStride, S->slab1, S->state, S->two_norm,
                                      S->rel_change, S->norms, &nb1, defx ? x->d : nullptr));
        std::swap(pcur, palt);
      } else {
        MK_TRY(launch_spmv_dot(A, pcur, S->s, S->slab1, &S->state->done, &nb1));
      }
      stamp(it);
      if (defx && it > 0) {   // (the first step's product ran without a fill: its K2 updates x itself)
        const int pb = (pcur == S->p2) ? 1 : 0;
        if (dinv)
          hipLaunchKernelGGL((pcg_update_r_kernel<true>), dim3(vb), dim3(kVecThreads), 0, c.stream, n, S->s, dinv, S->r, S->slab1, nb1, kSlabStride, S->slab2, S->state, pb, vec_flags(n));
        else
          hipLaunchKernelGGL((pcg_update_r_kernel<false>), dim3(vb), dim3(kVecThreads), 0, c.stream, n, S->s, dinv, S->r, S->slab1, nb1, kSlabStride, S->slab2, S->state, pb, vec_flags(n));
      } else if (dinv)
        hipLaunchKernelGGL((pcg_update_xr_kernel<true>), dim3(vb), dim3(kVecThreads), 0, c.stream, n, pcur, S->s, dinv, x->d, S->r, S->slab1, nb1, kSlabStride, S->slab2, S->state, vec_flags(n));
      else
        hipLaunchKernelGGL((pcg_update_xr_kernel<false>), dim3(vb), dim3(kVecThreads), 0, c.stream, n, pcur, S->s, dinv, x->d, S->r, S->slab1, nb1, kSlabStride, S->slab2, S->state, vec_flags(n));
      nb2 = vb;
    } else {
    MK_TRY(launch_spmv_dot(A, S->p, S->s, S->slab1, &S->state->done, &nb1));
    stamp(it);
    if (dinv) {
      hipLaunchKernelGGL((pcg_update_xr_kernel<true>), dim3(vb), dim3(kVecThreads), 0, c.stream, n, S->p, S->s, dinv, x->d, S->r, S->slab1, nb1, kSlabStride, S->slab2, S->state, vec_flags(n));
      stamp(it);
      hipLaunchKernelGGL((pcg_update_p_kernel<true>), dim3(vb), dim3(kVecThreads), 0, c.stream, n, S->r, dinv, S->p, S->slab2, vb, kSlabStride, S->two_norm, S->rel_change, S->state, S->norms, vec_flags(n));
    } else {
      hipLaunchKernelGGL((pcg_update_xr_kernel<false>), dim3(vb), dim3(kVecThreads), 0, c.stream, n, S->p, S->s, dinv, x->d, S->r, S->slab1, nb1, kSlabStride, S->slab2, S->state, vec_flags(n));
      stamp(it);
      hipLaunchKernelGGL((pcg_update_p_kernel<false>), dim3(vb), dim3(kVecThreads), 0, c.stream, n, S->r, dinv, S->p, S->slab2, vb, kSlabStride, S->two_norm, S->rel_change, S->state, S->norms, vec_flags(n));
    }
    }
    if ((it + 1) % S->check_every == 0) {
      const int slot = nchk & 1;
      // look at the previous checkpoint first: the GPU stays one batch ahead of the host
      if (pending >= 0) {
        MK_HIP(hipEventSynchronize(S->ev[pending]));
        if (hst[pending].done != 0) stop = true;
      }
      if (!stop) {
        MK_HIP(hipMemcpyAsync(&hst[slot], S->state, sizeof(PcgState), hipMemcpyDeviceToHost, c.stream));
        MK_HIP(hipEventRecord(S->ev[slot], c.stream));
        pending = slot;
        nchk++;
      }
    }
  }
  if (tev_used > 0 && tev_used == timed_iters * nslots) (void)hipEventRecord(S->tev[tev_used++], c.stream);   // solve ended with the timed window
  if (defx && S->max_iter > 1) {   // the solution update of the last completed step, if it is still pending
    hipLaunchKernelGGL(pcg_apply_pending_x_kernel, dim3(vb), dim3(kVecThreads), 0, c.stream, n, x->d, S->p, S->p2, S->state);
    hipLaunchKernelGGL(pcg_clear_pending_kernel, dim3(1), dim3(1), 0, c.stream, S->state);
  }
  if (halfx) {                     // ... likewise after a solve that ended on an even step of the halved update (p buffers: S->p, S->s)
    hipLaunchKernelGGL(pcg_apply_pending_x_kernel, dim3(vb), dim3(kVecThreads), 0, c.stream, n, x->d, S->p, S->s, S->state);
    hipLaunchKernelGGL(pcg_clear_pending_kernel, dim3(1), dim3(1), 0, c.stream, S->state);
  }
  if ((fuse_p || vfuse) && S->max_iter > 0)   // the test that the next step's first kernel would have made (K3's prologue; n = 0: no p update)
    hipLaunchKernelGGL((pcg_update_p_kernel<false>), dim3(1), dim3(kVecThreads), 0, c.stream, (int64_t)0, S->r, (const double *)nullptr, pcur, S->slab2, nb2, kSlabStride, S->two_norm, S->rel_change, S->state, S->norms, 0);
  MK_HIP(hipGetLastError());
  MK_HIP(hipMemcpyAsync(&hst[0], S->state, sizeof(PcgState), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  S->iterations = hst[0].iter;
  S->converged = hst[0].converged;
  S->breakdown = hst[0].breakdown;
  S->rel_res_norm = sqrt(hst[0].i_prod / hst[0].bi_prod);
  if (S->timing) {
    S->kt_slots = nslots;
    for (int k = 0; k < mk_pcg::kTimedSlots; k++) { S->kt_ms[k] = 0.0; S->kt_launches[k] = 0; }
    // only iterations that ran to their end: an early stop leaves later events unrecorded or around no-op kernels
    const long long ran = std::min<long long>(hst[0].iter - 1, timed_iters);
    if (tev_used == timed_iters * nslots + 1)
      for (long long i = 0; i < ran; i++)
        for (int k = 0; k < nslots; k++) {
          float ms = 0.f;
          if (hipEventElapsedTime(&ms, S->tev[i * nslots + k], S->tev[i * nslots + k + 1]) == hipSuccess) { S->kt_ms[k] += ms; S->kt_launches[k]++; }
        }
  }
  if (S->print_level > 1 && S->norms && S->iterations >= 0 && S->iterations < S->norms_cap) {
    std::vector<double> h((size_t)S->iterations + 1);
    MK_HIP(hipMemcpyAsync(h.data(), S->norms, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c.stream));
    MK_HIP(hipStreamSynchronize(c.stream));
    print_pcg_table(S->two_norm, h.data(), S->iterations, sqrt(hst[0].bi_prod));
  }
  return MK_OK;
}

int mk_pcg_get_num_iterations(const mk_pcg *S, long long *iterations) {
  if (!S || !iterations) return fail(MK_ERR_INVALID, "mk_pcg_get_num_iterations: null argument");
  *iterations = S->iterations;
  return MK_OK;
}

int mk_pcg_get_final_relative_residual_norm(const mk_pcg *S, double *norm) {
  if (!S || !norm) return fail(MK_ERR_INVALID, "mk_pcg_get_final_relative_residual_norm: null argument");
  *norm = S->rel_res_norm;
  return MK_OK;
}

int mk_pcg_get_kernel_timing(const mk_pcg *S, int *nslots, double *avg_ms, long long *launches, int capacity) {
  if (!S || !nslots || !avg_ms || !launches || capacity < mk_pcg::kTimedSlots) return fail(MK_ERR_INVALID, "mk_pcg_get_kernel_timing: bad argument (capacity >= 3)");
  if (!S->timing) return fail(MK_ERR_INVALID, "mk_pcg_get_kernel_timing: kernel timing was not switched on");
  *nslots = S->kt_slots;
  for (int k = 0; k < mk_pcg::kTimedSlots; k++) {
    launches[k] = S->kt_launches[k];
    avg_ms[k] = S->kt_launches[k] ? S->kt_ms[k] / (double)S->kt_launches[k] : 0.0;
  }
  return MK_OK;
}

int mk_pcg_get_converged(const mk_pcg *S, int *converged, int *breakdown) {
  if (!S) return fail(MK_ERR_INVALID, "mk_pcg_get_converged: null solver");
  if (converged) *converged = S->converged;
  if (breakdown) *breakdown = S->breakdown;
  return MK_OK;
}

int mk_pcg_get_residual_history(const mk_pcg *S, double *norms, int capacity) {
  MK_REQUIRE_INIT();
  if (!S || !norms || capacity < 0) return fail(MK_ERR_INVALID, "mk_pcg_get_residual_history: bad argument");
  if (!S->norms) return fail(MK_ERR_INVALID, "mk_pcg_get_residual_history: logging was off at setup");
  int count = (int)std::min<long long>(S->iterations + 1, capacity);
  if (S->breakdown == 1 || S->breakdown == 2) count = (int)std::min<long long>(S->iterations, capacity);  // no norm for the aborted step
  if (count > 0) {
    MK_HIP(hipMemcpyAsync(norms, S->norms, sizeof(double) * count, hipMemcpyDeviceToHost, ctx().stream));
    MK_HIP(hipStreamSynchronize(ctx().stream));
  }
  return count;
}

int mk_pcg_destroy(mk_pcg *S) {
  if (!S) return MK_OK;
  if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
  pcg_release(S);
  delete S;
  return MK_OK;
}

}  // extern "C"

// =========================================================================== Jacobi
// Extension, solver id 70 (the reference has no Jacobi iteration: SURVEY.md 8a row A11).
//   x_{k+1} = x_k + D^-1 (b - A x_k), D taken from the row itself while it streams by;
//   the same pass yields ||b - A x_k||^2, tested every check_every iterations.
// Algorithmic bytes per iteration: 12*nnz + 28*n.
extern "C" {

int mk_jacobi_create(mk_jacobi **solver) {
  MK_REQUIRE_INIT();
  if (!solver) return fail(MK_ERR_INVALID, "mk_jacobi_create: null argument");
  *solver = new mk_jacobi();
  return MK_OK;
}
int mk_jacobi_set_max_iter(mk_jacobi *S, int v) {
  if (!S || v < 0) return fail(MK_ERR_INVALID, "mk_jacobi_set_max_iter: bad argument");
  S->max_iter = v; return MK_OK;
}
int mk_jacobi_set_tol(mk_jacobi *S, double v) {
  if (!S || !(v >= 0.0)) return fail(MK_ERR_INVALID, "mk_jacobi_set_tol: bad argument");
  S->tol = v; return MK_OK;
}
int mk_jacobi_set_check_every(mk_jacobi *S, int v) {
  if (!S || v < 1) return fail(MK_ERR_INVALID, "mk_jacobi_set_check_every: bad argument");
  S->check_every = v; return MK_OK;
}

int mk_jacobi_setup(mk_jacobi *S, const mk_matrix *A, const mk_vector *b, const mk_vector *x) {
  MK_REQUIRE_INIT();
  if (!S || !A || !b || !x) return fail(MK_ERR_INVALID, "mk_jacobi_setup: null argument");
  if (A->nrows != A->ncols) return fail(MK_ERR_INVALID, "mk_jacobi_setup: matrix must be square");
  if (b->n != A->nrows || x->n != A->nrows) return fail(MK_ERR_INVALID, "mk_jacobi_setup: vector length does not match the matrix");
  dev_free(S->xalt); dev_free(S->slab); dev_free(S->state);
  S->xalt = nullptr; S->slab = nullptr; S->state = nullptr;
  S->n = A->nrows;
  MK_TRY(dev_alloc(&S->xalt, (size_t)S->n));
  MK_TRY(dev_alloc(&S->slab, (size_t)kSlabStride));
  MK_TRY(dev_alloc(&S->state, 1));
  MK_TRY(analyse_matrix(const_cast<mk_matrix *>(A)));
  return MK_OK;
}

int mk_jacobi_solve(mk_jacobi *S, const mk_matrix *A, const mk_vector *b, mk_vector *x) {
  MK_REQUIRE_INIT();
  if (!S || !A || !b || !x) return fail(MK_ERR_INVALID, "mk_jacobi_solve: null argument");
  if (S->n != A->nrows || !S->state) return fail(MK_ERR_INVALID, "mk_jacobi_solve: call mk_jacobi_setup with this matrix first");
  if (b->n != S->n || x->n != S->n) return fail(MK_ERR_INVALID, "mk_jacobi_solve: vector length does not match the matrix");
  Context &c = ctx();
  const int64_t n = S->n;
  S->iterations = 0; S->rel_res_norm = 0.0; S->converged = 0;
  if (n == 0) return MK_OK;
  JacobiState *hst = static_cast<JacobiState *>(c.pinned);
  const int vb = vec_blocks(n);
  hipLaunchKernelGGL(dot_self_kernel, dim3(vb), dim3(256), 0, c.stream, n, b->d, S->slab);
  hipLaunchKernelGGL(jacobi_init_kernel, dim3(1), dim3(256), 0, c.stream, S->slab, vb, kSlabStride, S->tol, S->state);
  MK_HIP(hipMemcpyAsync(hst, S->state, sizeof(JacobiState), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  if (!std::isfinite(hst->bb) || hst->bb < 0.0)
    return fail(MK_ERR_INVALID, "mk_jacobi_solve: <b,b> = %g: NaN or Inf in the right-hand side", hst->bb);
  if (hst->bb == 0.0) {
    hipLaunchKernelGGL(copy_kernel, dim3(vb), dim3(256), 0, c.stream, x->d, b->d, n);
    MK_HIP(hipStreamSynchronize(c.stream));
    return MK_OK;
  }
  const int G = A->lanes;
  const int64_t rows_per_block = 256 / G;
  int blocks = (int)std::min<int64_t>((n + rows_per_block - 1) / rows_per_block, std::min<int64_t>((int64_t)c.num_cus * 8, kSlabStride));
  double *cur = x->d, *nxt = S->xalt;
  long long k = 0;
  int checks = 0;
  for (;;) {
    int sweep_blocks = 0;
    int src = launch_jacobi_sweep(A, b->d, nullptr, cur, nxt, S->slab, &S->state->done, &sweep_blocks);
    if (src == MK_OK) {
      blocks = sweep_blocks;
    } else if (src == MK_ERR_UNSUPPORTED) {
#define MK_JSTEP(GG) case GG: hipLaunchKernelGGL((jacobi_step_kernel<GG>), dim3(blocks), dim3(256), 0, c.stream, n, A->rowptr, A->col, A->val, b->d, cur, nxt, S->slab, S->state, (const double *)nullptr); break;
    switch (G) { MK_JSTEP(1) MK_JSTEP(2) MK_JSTEP(4) MK_JSTEP(8) MK_JSTEP(16) MK_JSTEP(32) MK_JSTEP(64)
      default: return fail(MK_ERR_INVALID, "jacobi: bad lanes per row %d", G); }
#undef MK_JSTEP
    } else {
      return src;
    }
    const bool last = (k >= S->max_iter);
    if (k % S->check_every == 0 || last) {
      hipLaunchKernelGGL(jacobi_check_kernel, dim3(1), dim3(256), 0, c.stream, S->slab, blocks, kSlabStride, k, S->state);
      checks++;
      // read the flag back every 4th check (and at the end); later steps are no-ops once it is set
      if (last || checks % 4 == 0) {
        MK_HIP(hipMemcpyAsync(hst, S->state, sizeof(JacobiState), hipMemcpyDeviceToHost, c.stream));
        MK_HIP(hipStreamSynchronize(c.stream));
        if (hst->done != 0 || last) break;
      }
    }
    std::swap(cur, nxt);
    k++;
  }
  MK_HIP(hipGetLastError());
  // x_k sits in the buffer that was `cur` when iteration hst->iter ran: parity of iter
  S->iterations = hst->iter;
  S->converged = hst->converged;
  S->rel_res_norm = sqrt(hst->rr / hst->bb);
  if (hst->iter & 1) {  // odd iterate lives in xalt
    hipLaunchKernelGGL(copy_kernel, dim3(vb), dim3(256), 0, c.stream, x->d, S->xalt, n);
    MK_HIP(hipStreamSynchronize(c.stream));
  }
  return MK_OK;
}

int mk_jacobi_get_num_iterations(const mk_jacobi *S, long long *iterations) {
  if (!S || !iterations) return fail(MK_ERR_INVALID, "mk_jacobi_get_num_iterations: null argument");
  *iterations = S->iterations; return MK_OK;
}
int mk_jacobi_get_final_relative_residual_norm(const mk_jacobi *S, double *norm) {
  if (!S || !norm) return fail(MK_ERR_INVALID, "mk_jacobi_get_final_relative_residual_norm: null argument");
  *norm = S->rel_res_norm; return MK_OK;
}
int mk_jacobi_get_converged(const mk_jacobi *S, int *converged) {
  if (!S || !converged) return fail(MK_ERR_INVALID, "mk_jacobi_get_converged: null argument");
  *converged = S->converged; return MK_OK;
}
int mk_jacobi_destroy(mk_jacobi *S) {
  if (!S) return MK_OK;
  if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
  dev_free(S->xalt); dev_free(S->slab); dev_free(S->state);
  delete S;
  return MK_OK;
}

}  // extern "C"
