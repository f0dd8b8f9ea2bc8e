/*
 * solvers.c -- host driver with the reference's command line and output lines
 * (src/solvers.c:98-152 flags, :762-772 result lines), calling libmi355_krylov.so where the
 * reference calls HYPRE.  One process drives one MI355X.
 *
 *   -solver 6   BiCGSTAB, no preconditioner                                (solvers.c:517-564)
 *   -solver 12  GMRES(30), no preconditioner                               (solvers.c:1183-1245)
 *   -solver 40  CGNR, no preconditioner                                    (solvers.c:864-906)
 *   -solver 50  PCG, no preconditioner, two-norm + rel_change stopping   (solvers.c:724-776)
 *   -solver 51  PCG with diagonal scaling      (extension)
 *   -solver 70  plain Jacobi iteration          (extension)
 * The other ids of the reference (AMG, ILU, ParaSails, GMRES, ...; solvers.c:465-1548) are HYPRE
 * preconditioner stacks outside this library's scope and are refused by name.
 * -p <file> [-id <gpu>] samples the board power while the solver runs (cpower.h, solvers.c:537-546).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sparse_matrix.h"
#include "power.h"

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

#define CK(call)                                                                  \
  do {                                                                            \
    int rc_ = (call);                                                             \
    if (rc_ != 0) {                                                               \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mk_last_error_string()); \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

static void usage(const char *argv0) {
  printf("\nUsage: %s [<options>]\n\n", argv0);
  printf("  -mtx <matrix.mtx>      : matrix market path\n");
  printf("  -b <vector.mtx>        : RHS matrix market format path (default: all ones)\n");
  printf("  -x <vector.mtx>        : initial guess (default: zeros)\n");
  printf("  -maxit <n>             : Max number of iterations (default 1000)\n");
  printf("  -tol <t>               : tolerance (default 1e-6)\n");
  printf("  -log <n>               : printing level out of the solver\n");
  printf("  -precond_log <n>       : accepted, unused (no HYPRE preconditioners here)\n");
  printf("  -niter <n>             : solve problem n times (later solves start from the previous x)\n");
  printf("  -dev <id>              : GPU to use (default 0)\n");
  printf("  -solver <ID>           : solver ID\n");
  printf("                           6  - BiCGSTAB\n");
  printf("                           12 - GMRES(30)\n");
  printf("                           40 - CGNR\n");
  printf("                           50 - PCG\n");
  printf("                           51 - PCG + diagonal scaling\n");
  printf("                           70 - Jacobi iteration\n");
  printf("  -p <file>              : append the board power in watts to <file>, 10 samples a second, while the solver runs\n");
  printf("  -id <device>           : GPU whose power is sampled (default: the one in use)\n");
  printf("  -cache                 : keep <matrix.mtx>.mkcsr (binary CSR) and load it instead of the text when present\n");
  printf("  -print_system          : write the assembled CSR back to IJ.out.A as text\n\n");
}

int main(int argc, char *argv[]) {
  int maxit = 1000, prnt_info = 2, precond_log = 1, niter = 1, solver_id = 0, print_system = 0, dev = 0, comp_id = 0, use_cache = 0;
  double tol = 1e-6;
  const char *mtx = NULL, *bfile = NULL, *xfile = NULL, *pow_file = NULL;
  (void)precond_log;
  int have_comp_id = 0;
  for (int i = 1; i < argc;) {
    const char *a = argv[i];
    const int has = i + 1 < argc;
    if (!strcmp(a, "-maxit") && has) { maxit = atoi(argv[i + 1]); i += 2; }
    else if (!strcmp(a, "-tol") && has) { tol = atof(argv[i + 1]); i += 2; }
    else if (!strcmp(a, "-log") && has) { prnt_info = atoi(argv[i + 1]); i += 2; }
    else if (!strcmp(a, "-niter") && has) { niter = atoi(argv[i + 1]); i += 2; }
    else if (!strcmp(a, "-padding")) { i += 1; }                       /* parsed and ignored by the reference too (:115-117) */
    else if (!strcmp(a, "-id") && has) { comp_id = atoi(argv[i + 1]); have_comp_id = 1; i += 2; }
    else if (!strcmp(a, "-precond_log") && has) { precond_log = atoi(argv[i + 1]); i += 2; }
    else if (!strcmp(a, "-solver") && has) { solver_id = atoi(argv[i + 1]); i += 2; }
    else if (!strcmp(a, "-mtx") && has) { mtx = argv[i + 1]; i += 2; }
    else if (!strcmp(a, "-p") && has) { pow_file = argv[i + 1]; i += 2; }
    else if (!strcmp(a, "-b") && has) { bfile = argv[i + 1]; i += 2; }
    else if (!strcmp(a, "-x") && has) { xfile = argv[i + 1]; i += 2; }
    else if (!strcmp(a, "-dev") && has) { dev = atoi(argv[i + 1]); i += 2; }
    else if (!strcmp(a, "-print_system")) { print_system = 1; i += 1; }
    else if (!strcmp(a, "-cache")) { use_cache = 1; i += 1; }
    else if (!strcmp(a, "-help")) { usage(argv[0]); return 0; }
    else i += 1;                                                      /* unknown flags are skipped (:149-151) */
  }
  if (!mtx) { usage(argv[0]); return 1; }
  if (solver_id != 6 && solver_id != 12 && solver_id != 40 && solver_id != 50 && solver_id != 51 && solver_id != 70) {
    printf("Invalid solver id specified.\n");
    fprintf(stderr, "solver %d is a HYPRE preconditioner/solver stack outside this library; use 6, 12, 40, 50, 51 or 70\n", solver_id);
    return 1;
  }

  CK(mk_init(dev));                                                   /* HYPRE_Init (:79) */

  mk_matrix *A = NULL;
  unsigned int n = 0;
  char cache_path[4096];
  snprintf(cache_path, sizeof cache_path, "%s.mkcsr", mtx);
  SparseMatrixCSR cached;
  double t0 = now_s();
  if (use_cache && read_csr_cache(cache_path, &cached) == 0) {
    if (cached.nrows != cached.ncolumns) { fprintf(stderr, "matrix is %u x %u: a square system is required\n", cached.nrows, cached.ncolumns); return 1; }
    CK(mk_matrix_from_csr(&cached, &A));
    n = cached.nrows;
    free_csr(&cached);
    fprintf(stderr, "Loaded binary CSR cache %s: %lg msec\n", cache_path, (now_s() - t0) * 1e3);
  } else {
    SparseMatrixCOO coo;
    if (fast_load_from_mtx_file(mtx, &coo) != 0) return 1;              /* :196 */
    double t1 = now_s();
    if (coo.nrows != coo.ncolumns) { fprintf(stderr, "matrix is %u x %u: a square system is required\n", coo.nrows, coo.ncolumns); return 1; }
    /* the reference sorts on the host (:198) and hands slices to HYPRE (:330); here the
     * unsorted COO goes to the device, which sorts, converts and keeps the CSR */
    CK(mk_matrix_from_coo(&coo, &A));                                   /* :289-374 */
    double t2 = now_s();
    fprintf(stderr, "Finish reading and sorting of COO data\n");        /* :212 */
    fprintf(stderr, "Dataset reading time: %lg msec, device assembly: %lg msec\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
    n = coo.nrows;
    free_coo(&coo);                                                     /* :376-378 */
    if (use_cache) {
      SparseMatrixCSR csr;
      CK(mk_matrix_get_csr(A, &csr));
      if (write_csr_cache(cache_path, &csr) != 0) fprintf(stderr, "could not write %s\n", cache_path);
      free_csr(&csr);
    }
  }

  mk_vector *b = NULL, *x = NULL;
  CK(mk_vector_create(n, &b));
  CK(mk_vector_create(n, &x));
  if (bfile) {                                                        /* :413-418 */
    double *h = NULL; unsigned int m = 0;
    if (fast_load_from_array_file(bfile, &h, &m) != 0) return 1;
    if (m < n) { fprintf(stderr, "%s holds %u values, %u needed\n", bfile, m, n); return 1; }
    CK(mk_vector_set(b, h));
    free(h);
  } else {
    CK(mk_vector_fill(b, 1.0));                                       /* :404-412 */
  }
  if (xfile) {                                                        /* :431-436 */
    double *h = NULL; unsigned int m = 0;
    if (fast_load_from_array_file(xfile, &h, &m) != 0) return 1;
    if (m < n) { fprintf(stderr, "%s holds %u values, %u needed\n", xfile, m, n); return 1; }
    CK(mk_vector_set(x, h));
    free(h);
  } else {
    CK(mk_vector_fill(x, 0.0));                                       /* :420-430 */
  }
  if (print_system) {                                                 /* :450-454 */
    SparseMatrixCSR csr;
    CK(mk_matrix_get_csr(A, &csr));
    FILE *f = fopen("IJ.out.A", "w");
    if (f) {
      fprintf(f, "%u %u %u\n", csr.nrows, csr.ncolumns, csr.nnz);
      for (unsigned int r = 0; r < csr.nrows; r++)
        for (int k = csr.row_pointers[r]; k < csr.row_pointers[r + 1]; k++)
          fprintf(f, "%u %d %.17g\n", r, csr.column_indices[k], csr.values[k]);
      fclose(f);
    }
    free_csr(&csr);
  }

  /* power sampling brackets setup + solve as the reference's c_popen / c_pclose pair does (solvers.c:537-546) */
  power_sampler *sampler = NULL;
  if (pow_file) {
    char bus[64];
    CK(mk_get_device_pci_bus_id(have_comp_id ? comp_id : -1, bus, (int)sizeof bus));
    if (power_sampler_start(bus, pow_file, 100, &sampler) != 0) sampler = NULL;
  }

  long long num_iterations = 0;
  double final_res_norm = 0.0, walltime;
  if (solver_id == 50 || solver_id == 51) {
    mk_pcg *solver = NULL;
    CK(mk_pcg_create(&solver));                                       /* :729 */
    CK(mk_pcg_set_max_iter(solver, maxit));                           /* :732 */
    CK(mk_pcg_set_tol(solver, tol));                                  /* :734 */
    CK(mk_pcg_set_two_norm(solver, 1));                               /* :735 */
    CK(mk_pcg_set_rel_change(solver, 1));                             /* :737 */
    CK(mk_pcg_set_print_level(solver, prnt_info));                    /* :742 */
    CK(mk_pcg_set_logging(solver, 1));                                /* :744 */
    if (solver_id == 51) CK(mk_pcg_set_precond(solver, MK_PRECOND_JACOBI));
    CK(mk_pcg_setup(solver, A, b, x));                                /* :747 */
    CK(mk_synchronize());
    double t = -now_s();                                              /* :752 */
    for (int i = 0; i < niter; i++) CK(mk_pcg_solve(solver, A, b, x)); /* :753-754 */
    CK(mk_synchronize());
    t += now_s();
    walltime = t;
    printf("\nPCG Solve time = %f seconds\n\n", walltime);            /* :762 */
    CK(mk_pcg_get_num_iterations(solver, &num_iterations));           /* :765 */
    CK(mk_pcg_get_final_relative_residual_norm(solver, &final_res_norm)); /* :766 */
    CK(mk_pcg_destroy(solver));                                       /* :775 */
  } else if (solver_id == 6) {
    mk_bicgstab *solver = NULL;
    CK(mk_bicgstab_create(&solver));                                  /* :524 */
    CK(mk_bicgstab_set_print_level(solver, prnt_info));               /* :527 */
    CK(mk_bicgstab_set_logging(solver, 1));                           /* :529 */
    CK(mk_bicgstab_set_tol(solver, tol));                             /* :532 */
    CK(mk_bicgstab_set_max_iter(solver, maxit));                      /* :533 */
    CK(mk_bicgstab_setup(solver, A, b, x));                           /* :536 */
    CK(mk_synchronize());
    double t = -now_s();                                              /* :540 */
    for (int i = 0; i < niter; i++) CK(mk_bicgstab_solve(solver, A, b, x)); /* :541-542 */
    CK(mk_synchronize());
    t += now_s();
    walltime = t;
    printf("\nParCSRBiCGSTABC Solve time = %f seconds\n\n", walltime); /* :550, the reference's spelling */
    CK(mk_bicgstab_get_num_iterations(solver, &num_iterations));      /* :553 */
    CK(mk_bicgstab_get_final_relative_residual_norm(solver, &final_res_norm)); /* :554 */
    CK(mk_bicgstab_destroy(solver));                                  /* :563 */
  } else if (solver_id == 12) {
    mk_gmres *solver = NULL;
    double ts = -now_s();                                             /* :1187 */
    CK(mk_gmres_create(&solver));                                     /* :1189 */
    CK(mk_gmres_set_max_iter(solver, maxit));                         /* :1191 */
    CK(mk_gmres_set_k_dim(solver, 30));                               /* :1192, restart = 30 (:460) */
    CK(mk_gmres_set_tol(solver, tol));                                /* :1193 */
    CK(mk_gmres_set_print_level(solver, prnt_info));                  /* :1195 */
    CK(mk_gmres_set_logging(solver, 1));                              /* :1196 */
    CK(mk_gmres_setup(solver, A, b, x));                              /* :1200 */
    CK(mk_synchronize());
    ts += now_s();
    printf("\nGMRES Setup time = %f seconds\n\n", ts);               /* :1206 */
    double t = -now_s();                                              /* :1213 */
    for (int i = 0; i < niter; i++) CK(mk_gmres_solve(solver, A, b, x)); /* :1216-1217 */
    CK(mk_synchronize());
    t += now_s();
    walltime = t;
    printf("\nGMRES Solve time = %f seconds\n\n", walltime);         /* :1227 */
    CK(mk_gmres_get_num_iterations(solver, &num_iterations));         /* :1231 */
    CK(mk_gmres_get_final_relative_residual_norm(solver, &final_res_norm)); /* :1232 */
    CK(mk_gmres_destroy(solver));
  } else if (solver_id == 40) {
    mk_cgnr *solver = NULL;
    CK(mk_cgnr_create(&solver));                                      /* :869 */
    CK(mk_cgnr_set_max_iter(solver, maxit));                          /* :872 */
    CK(mk_cgnr_set_tol(solver, tol));                                 /* :873 */
    CK(mk_cgnr_set_logging(solver, 1));                               /* :874 */
    CK(mk_cgnr_setup(solver, A, b, x));                               /* :877 */
    CK(mk_synchronize());
    double t = -now_s();                                              /* :882 */
    for (int i = 0; i < niter; i++) CK(mk_cgnr_solve(solver, A, b, x)); /* :883-884 */
    CK(mk_synchronize());
    t += now_s();
    walltime = t;
    printf("\nCGNR Solve time = %f seconds\n\n", walltime);          /* :892 */
    CK(mk_cgnr_get_num_iterations(solver, &num_iterations));          /* :895 */
    CK(mk_cgnr_get_final_relative_residual_norm(solver, &final_res_norm)); /* :896 */
    CK(mk_cgnr_destroy(solver));                                      /* :905 */
  } else {
    mk_jacobi *solver = NULL;
    CK(mk_jacobi_create(&solver));
    CK(mk_jacobi_set_max_iter(solver, maxit));
    CK(mk_jacobi_set_tol(solver, tol));
    CK(mk_jacobi_setup(solver, A, b, x));
    CK(mk_synchronize());
    double t = -now_s();
    for (int i = 0; i < niter; i++) CK(mk_jacobi_solve(solver, A, b, x));
    CK(mk_synchronize());
    t += now_s();
    walltime = t;
    printf("\nJacobi Solve time = %f seconds\n\n", walltime);
    CK(mk_jacobi_get_num_iterations(solver, &num_iterations));
    CK(mk_jacobi_get_final_relative_residual_norm(solver, &final_res_norm));
    CK(mk_jacobi_destroy(solver));
  }
  printf("\n");
  printf("Iterations = %lld\n", num_iterations);                      /* :770 */
  printf("Final Relative Residual Norm = %e\n", final_res_norm);      /* :771 */
  printf("\n");
  if (sampler) {
    double mean_w = 0.0, joules = 0.0;
    int ns = 0;
    power_sampler_stop(sampler, &mean_w, &joules, &ns);
    printf("Power samples = %d, mean board power = %.1f W, energy = %.1f J\n", ns, mean_w, joules);
  }
  /* extra lines the reference does not print: rate, for scripts/grep_res.sh-style scraping */
  if (walltime > 0.0 && num_iterations > 0) printf("Iterations per second = %f\n", (double)num_iterations * niter / walltime);

  mk_vector_destroy(b);
  mk_vector_destroy(x);
  mk_matrix_destroy(A);                                               /* :1555 */
  mk_finalize();                                                      /* :1560 */
  return 0;
}
