/*
 * solvers.c -- host driver with the reference's command line and output lines
 * (src/solvers.c:98-152 flags, :762-772 result lines), calling libmi355_krylov.so where the
 * reference calls HYPRE.  One process drives one MI355X; with -np N (or RANK / WORLD_SIZE in the
 * environment, as mpirun / torchrun / srun set them) N such processes solve one system over row blocks
 * (solvers.c:68-70 MPI_Init, :214-231 partition, :270-281 row ranges, :752-763 max-over-ranks time).
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
#include <unistd.h>

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

static int g_single_reduction = 0;   /* -single_reduction: ask mk_dist_pcg_solve for one all-reduce per iteration */

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
  printf("  -single_reduction      : row-block PCG (50, 51 with -np / -dist): one all-reduce per iteration instead of two\n");
  printf("  -p <file>              : append the board power in watts to <file>, 10 samples a second, while the solver runs\n");
  printf("  -id <device>           : GPU whose power is sampled (default: the one in use)\n");
  printf("  -np <N> -rank <r>      : N processes, one GPU each, solve one system over nnz-balanced row blocks\n");
  printf("                           (defaults: WORLD_SIZE / RANK, OMPI_COMM_WORLD_*, PMI_*, SLURM_* from the environment;\n");
  printf("                           -dev defaults to LOCAL_RANK); solvers 50, 51 and 70\n");
  printf("  -id_file <path>        : file on a shared directory through which rank 0 hands the communicator id to the others\n");
  printf("  -dist                  : take the row-block path even with one process\n");
  printf("  -cache                 : keep <matrix.mtx>.mkcsr (binary CSR) and load it instead of the text when present\n");
  printf("  -print_system          : write the system as IJ.out.A.XXXXX, IJ.out.b.XXXXX, IJ.out.x.XXXXX (XXXXX = rank)\n\n");
}

static int env_int(const char *const *names, int fallback) {
  for (int i = 0; names[i]; i++) {
    const char *v = getenv(names[i]);
    if (v && *v) return atoi(v);
  }
  return fallback;
}

/* HYPRE_IJMatrixPrint / HYPRE_IJVectorPrint layout (solvers.c:450-454): one file per rank, first line the
 * index range this rank owns, then "row col value" / "row value" lines with global indices */
static void print_system_files(int rank, long long row_lo, const SparseMatrixCSR *csr, const double *b, const double *x) {
  char name[64];
  snprintf(name, sizeof name, "IJ.out.A.%05d", rank);
  FILE *f = fopen(name, "w");
  if (f) {
    fprintf(f, "%lld %lld %d %u\n", row_lo, row_lo + (long long)csr->nrows - 1, 0, csr->ncolumns - 1);
    for (unsigned int r = 0; r < csr->nrows; r++)
      for (int k = csr->row_pointers[r]; k < csr->row_pointers[r + 1]; k++)
        fprintf(f, "%lld %d %.14e\n", row_lo + r, csr->column_indices[k], csr->values[k]);
    fclose(f);
  }
  const double *vec[2] = {b, x};
  const char *tag[2] = {"b", "x"};
  for (int v = 0; v < 2; v++) {
    snprintf(name, sizeof name, "IJ.out.%s.%05d", tag[v], rank);
    f = fopen(name, "w");
    if (!f) continue;
    fprintf(f, "%lld %lld\n", row_lo, row_lo + (long long)csr->nrows - 1);
    for (unsigned int r = 0; r < csr->nrows; r++) fprintf(f, "%lld %.14e\n", row_lo + r, vec[v][r]);
    fclose(f);
  }
}

/* Rank 0 draws the communicator id and leaves it in `path`; the others wait for THIS launch's file (idfile.c: leftovers of
 * a run that died are recognised by their launch nonce or their age).  With MPI at hand this is one MPI_Bcast of 256
 * bytes; the driver itself needs no MPI.  Rank 0 removes the file on every exit path it controls. */
static const char *g_id_file_to_remove = NULL;
static void remove_id_file(void) {
  if (g_id_file_to_remove) remove(g_id_file_to_remove);
}
static int exchange_comm_id(const char *path, int rank, unsigned char *id256) {
  if (rank == 0) {
    mk_idfile_clear(path);                 /* before anything slow: a leftover must not be there when the others look */
    g_id_file_to_remove = path;
    atexit(remove_id_file);                /* error returns and exit() alike */
    if (mk_comm_get_unique_id(id256) != 0) { fprintf(stderr, "mk_comm_get_unique_id: %s\n", mk_last_error_string()); return 1; }
    if (mk_idfile_publish(path, id256) != 0) { fprintf(stderr, "cannot write the communicator id to %s\n", path); return 1; }
    return 0;
  }
  const char *age = getenv("MK_ID_FILE_MAX_AGE");
  const int rc = mk_idfile_wait(path, id256, 600.0 /* rank 0 may still be starting */, age ? atof(age) : 120.0);
  if (rc == 2) fprintf(stderr, "rank %d: %s holds the id of another launch (stale file?) and no new one appeared\n", rank, path);
  else if (rc != 0) fprintf(stderr, "rank %d: no communicator id appeared in %s\n", rank, path);
  return rc != 0;
}

/* -np N: this process is one rank of N (solvers.c with num_procs > 1).  Every rank reads the file, keeps the rows
 * mk_partition_rows gives it, and the ranks solve together through mk_dist_*. */
static int run_distributed(const char *mtx, const char *bfile, const char *xfile, const char *id_file, int np, int rank,
                           int solver_id, int maxit, double tol, int prnt_info, int niter, int print_system) {
  mk_comm *comm = NULL;
  unsigned char id[256];
  if (np > 1 && !id_file) { fprintf(stderr, "-np %d needs -id_file <path on a directory all ranks see>\n", np); return 1; }
  if (np == 1 && !id_file) {
    CK(mk_comm_get_unique_id(id));
  } else if (exchange_comm_id(id_file, rank, id) != 0) {
    return 1;
  }
  CK(mk_comm_create(id, np, rank, &comm));                              /* MPI_Init, MPI_Comm_rank/size (:68-70) */

  SparseMatrixCOO coo, mine;
  double t0 = now_s();
  if (fast_load_from_mtx_file(mtx, &coo) != 0) return 1;                /* :196, on every rank instead of a broadcast (:263-265) */
  if (coo.nrows != coo.ncolumns) { fprintf(stderr, "matrix is %u x %u: a square system is required\n", coo.nrows, coo.ncolumns); return 1; }
  const long long n = coo.nrows;
  long long *rowptr = (long long *)malloc(((size_t)n + 1) * sizeof(long long));
  long long *row_start = (long long *)malloc(((size_t)np + 1) * sizeof(long long));
  if (!rowptr || !row_start || coo_row_pointers64(&coo, rowptr) != 0) { fprintf(stderr, "%s: row index out of range\n", mtx); return 1; }
  CK(mk_partition_rows(n, (const int64_t *)rowptr, np, (int64_t *)row_start));   /* :214-231 */
  free(rowptr);
  const long long lo = row_start[rank], hi = row_start[rank + 1];                /* ilower / iupper + 1 (:270-281) */
  if (slice_coo_rows(&coo, lo, hi, &mine) != 0) { fprintf(stderr, "rank %d: cannot slice rows %lld..%lld\n", rank, lo, hi); return 1; }
  free_coo(&coo);                                                       /* :376-378 */
  mk_matrix *block = NULL;
  mk_dist_matrix *A = NULL;
  CK(mk_matrix_from_coo(&mine, &block));                                /* device sort + conversion of this rank's rows */
  free_coo(&mine);
  SparseMatrixCSR csr;
  if (print_system) CK(mk_matrix_get_csr(block, &csr));
  CK(mk_dist_matrix_create(block, n, (const int64_t *)row_start, comm, &A));   /* :289-374 */
  if (rank == 0) {
    fprintf(stderr, "Finish reading and sorting of COO data\n");        /* :212 */
    fprintf(stderr, "Dataset reading + device assembly: %lg msec, rank 0 owns rows 0..%lld of %lld\n", (now_s() - t0) * 1e3, hi - 1, n);
  }

  mk_vector *b = NULL, *x = NULL;
  CK(mk_dist_vector_create(A, &b));
  CK(mk_dist_vector_create(A, &x));
  double *hb = NULL, *hx = NULL;
  if (bfile) {                                                          /* :413-418: every rank takes its slice */
    unsigned int m = 0;
    if (load_array_file(bfile, &hb, &m) != 0) return 1;
    if ((long long)m < n) { fprintf(stderr, "%s holds %u values, %lld needed\n", bfile, m, n); return 1; }
    CK(mk_vector_set(b, hb + lo));
  } else {
    CK(mk_vector_fill(b, 1.0));
  }
  if (xfile) {
    unsigned int m = 0;
    if (load_array_file(xfile, &hx, &m) != 0) return 1;
    if ((long long)m < n) { fprintf(stderr, "%s holds %u values, %lld needed\n", xfile, m, n); return 1; }
    CK(mk_vector_set(x, hx + lo));
  } else {
    CK(mk_vector_fill(x, 0.0));
  }
  if (print_system) {
    const long long nl = hi - lo;
    double *pb = (double *)malloc((size_t)(nl > 0 ? nl : 1) * sizeof(double)), *px = (double *)malloc((size_t)(nl > 0 ? nl : 1) * sizeof(double));
    if (pb && px) {
      CK(mk_vector_get(b, pb));
      CK(mk_vector_get(x, px));
      print_system_files(rank, lo, &csr, pb, px);
    }
    free(pb); free(px);
    free_csr(&csr);
  }
  free(hb); free(hx);

  long long num_iterations = 0;
  double final_res_norm = 0.0, walltime = 0.0;
  const char *label = solver_id == 70 ? "Jacobi" : solver_id == 6 ? "ParCSRBiCGSTABC" : solver_id == 12 ? "GMRES" : solver_id == 40 ? "CGNR" : "PCG";
  if (solver_id == 6) {
    mk_bicgstab *solver = NULL;
    CK(mk_bicgstab_create(&solver));                                    /* :524 */
    CK(mk_bicgstab_set_print_level(solver, rank == 0 ? prnt_info : 0)); /* :527 */
    CK(mk_bicgstab_set_logging(solver, 1));                             /* :529 */
    CK(mk_bicgstab_set_tol(solver, tol));                               /* :532 */
    CK(mk_bicgstab_set_max_iter(solver, maxit));                        /* :533 */
    CK(mk_synchronize());
    double t = -now_s();                                                /* :540 */
    for (int i = 0; i < niter; i++) CK(mk_dist_bicgstab_solve(1, &solver, &A, &b, &x));   /* :536-542 */
    CK(mk_synchronize());
    walltime = t + now_s();
    CK(mk_bicgstab_get_num_iterations(solver, &num_iterations));        /* :553 */
    CK(mk_bicgstab_get_final_relative_residual_norm(solver, &final_res_norm)); /* :554 */
    CK(mk_bicgstab_destroy(solver));                                    /* :563 */
  } else if (solver_id == 12) {
    mk_gmres *solver = NULL;
    CK(mk_gmres_create(&solver));                                       /* :1189 */
    CK(mk_gmres_set_max_iter(solver, maxit));                           /* :1191 */
    CK(mk_gmres_set_k_dim(solver, 30));                                 /* :1192 */
    CK(mk_gmres_set_tol(solver, tol));                                  /* :1193 */
    CK(mk_gmres_set_print_level(solver, rank == 0 ? prnt_info : 0));    /* :1195 */
    CK(mk_gmres_set_logging(solver, 1));                                /* :1196 */
    CK(mk_synchronize());
    double t = -now_s();                                                /* :1213 */
    for (int i = 0; i < niter; i++) CK(mk_dist_gmres_solve(1, &solver, &A, &b, &x));      /* :1200-1217 */
    CK(mk_synchronize());
    walltime = t + now_s();
    CK(mk_gmres_get_num_iterations(solver, &num_iterations));           /* :1231 */
    CK(mk_gmres_get_final_relative_residual_norm(solver, &final_res_norm)); /* :1232 */
    CK(mk_gmres_destroy(solver));
  } else if (solver_id == 40) {
    mk_cgnr *solver = NULL;
    CK(mk_cgnr_create(&solver));                                        /* :869 */
    CK(mk_cgnr_set_max_iter(solver, maxit));                            /* :872 */
    CK(mk_cgnr_set_tol(solver, tol));                                   /* :873 */
    CK(mk_cgnr_set_logging(solver, 1));                                 /* :874 */
    CK(mk_synchronize());
    double t = -now_s();                                                /* :882 */
    for (int i = 0; i < niter; i++) CK(mk_dist_cgnr_solve(1, &solver, &A, &b, &x));       /* :877-884 */
    CK(mk_synchronize());
    walltime = t + now_s();
    CK(mk_cgnr_get_num_iterations(solver, &num_iterations));            /* :895 */
    CK(mk_cgnr_get_final_relative_residual_norm(solver, &final_res_norm)); /* :896 */
    CK(mk_cgnr_destroy(solver));                                        /* :905 */
  } else if (solver_id == 70) {
    mk_jacobi *solver = NULL;
    CK(mk_jacobi_create(&solver));
    CK(mk_jacobi_set_max_iter(solver, maxit));
    CK(mk_jacobi_set_tol(solver, tol));
    CK(mk_synchronize());
    double t = -now_s();
    for (int i = 0; i < niter; i++) CK(mk_dist_jacobi_solve(1, &solver, &A, &b, &x));
    CK(mk_synchronize());
    walltime = t + now_s();
    CK(mk_jacobi_get_num_iterations(solver, &num_iterations));
    CK(mk_jacobi_get_final_relative_residual_norm(solver, &final_res_norm));
    CK(mk_jacobi_destroy(solver));
  } else {
    mk_pcg *solver = NULL;
    CK(mk_pcg_create(&solver));                                         /* :729 */
    CK(mk_pcg_set_max_iter(solver, maxit));                             /* :732 */
    CK(mk_pcg_set_tol(solver, tol));                                    /* :734 */
    CK(mk_pcg_set_two_norm(solver, 1));                                 /* :735 */
    CK(mk_pcg_set_rel_change(solver, 1));                               /* :737 */
    CK(mk_pcg_set_print_level(solver, rank == 0 ? prnt_info : 0));      /* :742 */
    CK(mk_pcg_set_logging(solver, 1));                                  /* :744 */
    if (solver_id == 51) CK(mk_pcg_set_precond(solver, MK_PRECOND_JACOBI));
    if (g_single_reduction) CK(mk_pcg_set_single_reduction(solver, 1));
    CK(mk_synchronize());
    double t = -now_s();                                                /* :752 */
    for (int i = 0; i < niter; i++) CK(mk_dist_pcg_solve(1, &solver, &A, &b, &x));   /* :747-754 */
    CK(mk_synchronize());
    walltime = t + now_s();                                             /* :755 */
    if (g_single_reduction && rank == 0) {
      int used = 0;
      CK(mk_pcg_get_single_reduction(solver, &used));
      printf("CG step = %s\n", used ? "single reduction per iteration" : "two reductions per iteration (single-reduction step declined: a rank's diagonal block is not on the sweep kernel)");
    }
    CK(mk_pcg_get_num_iterations(solver, &num_iterations));             /* :765 */
    CK(mk_pcg_get_final_relative_residual_norm(solver, &final_res_norm)); /* :766 */
    CK(mk_pcg_destroy(solver));                                         /* :775 */
  }
  CK(mk_comm_all_reduce_host(comm, &walltime, 1, 1));                   /* MPI_Allreduce(MPI_MAX) :760 */
  if (rank == 0) {
    printf("\n%s Solve time = %f seconds\n\n", label, walltime);      /* :762 */
    printf("\n");
    printf("Iterations = %lld\n", num_iterations);                      /* :770 */
    printf("Final Relative Residual Norm = %e\n", final_res_norm);      /* :771 */
    printf("\n");
    if (walltime > 0.0 && num_iterations > 0) printf("Iterations per second = %f\n", (double)num_iterations * niter / walltime);
    printf("Ranks = %d\n", np);
  }
  free(row_start);
  mk_vector_destroy(b);
  mk_vector_destroy(x);
  mk_dist_matrix_destroy(A);
  mk_comm_destroy(comm);
  if (rank == 0 && id_file) { remove(id_file); g_id_file_to_remove = NULL; }
  mk_finalize();
  return 0;
}

int main(int argc, char *argv[]) {
  int maxit = 1000, prnt_info = 2, precond_log = 1, niter = 1, solver_id = 0, print_system = 0, dev = -1, comp_id = 0, use_cache = 0;
  double tol = 1e-6;
  const char *mtx = NULL, *bfile = NULL, *xfile = NULL, *pow_file = NULL, *id_file = NULL;
  static const char *const size_env[] = {"WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS", NULL};
  static const char *const rank_env[] = {"RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "SLURM_PROCID", NULL};
  static const char *const local_env[] = {"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", NULL};
  int np = env_int(size_env, 1), rank = env_int(rank_env, 0), force_dist = 0;
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
    else if (!strcmp(a, "-np") && has) { np = atoi(argv[i + 1]); i += 2; }
    else if (!strcmp(a, "-rank") && has) { rank = atoi(argv[i + 1]); i += 2; }
    else if (!strcmp(a, "-id_file") && has) { id_file = argv[i + 1]; i += 2; }
    else if (!strcmp(a, "-dist")) { force_dist = 1; i += 1; }
    else if (!strcmp(a, "-single_reduction")) { g_single_reduction = 1; i += 1; }
    else if (!strcmp(a, "-help")) { usage(argv[0]); return 0; }
    else i += 1;                                                      /* unknown flags are skipped (:149-151) */
  }
  if (!mtx) { usage(argv[0]); return 1; }
  if (solver_id != 6 && solver_id != 12 && solver_id != 40 && solver_id != 50 && solver_id != 51 && solver_id != 70) {
    printf("Invalid solver id specified.\n");
    fprintf(stderr, "solver %d is a HYPRE preconditioner/solver stack outside this library; use 6, 12, 40, 50, 51 or 70\n", solver_id);
    return 1;
  }

  if (np < 1 || rank < 0 || rank >= np) { fprintf(stderr, "rank %d of %d processes: out of range\n", rank, np); return 1; }
  if (dev < 0) dev = (np > 1) ? env_int(local_env, rank) : 0;         /* one rank per GPU */

  CK(mk_init(dev));                                                   /* HYPRE_Init (:79) */
  if (np > 1 || force_dist)
    return run_distributed(mtx, bfile, xfile, id_file, np, rank, solver_id, maxit, tol, prnt_info, niter, print_system);

  mk_matrix *A = NULL;
  unsigned int n = 0;
  char cache_path[4096];
  snprintf(cache_path, sizeof cache_path, "%s.mkcsr", mtx);
  SparseMatrixCSR cached;
  double t0 = now_s();
  if (use_cache && read_csr_cache_for(cache_path, &cached, mtx) == 0) {   /* refused when the .mtx changed since */
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
      if (write_csr_cache_for(cache_path, &csr, mtx) != 0) fprintf(stderr, "could not write %s\n", cache_path);
      free_csr(&csr);
    }
  }

  mk_vector *b = NULL, *x = NULL;
  CK(mk_vector_create(n, &b));
  CK(mk_vector_create(n, &x));
  if (bfile) {                                                        /* :413-418 */
    double *h = NULL; unsigned int m = 0;
    if (load_array_file(bfile, &h, &m) != 0) return 1;
    if (m < n) { fprintf(stderr, "%s holds %u values, %u needed\n", bfile, m, n); return 1; }
    CK(mk_vector_set(b, h));
    free(h);
  } else {
    CK(mk_vector_fill(b, 1.0));                                       /* :404-412 */
  }
  if (xfile) {                                                        /* :431-436 */
    double *h = NULL; unsigned int m = 0;
    if (load_array_file(xfile, &h, &m) != 0) return 1;
    if (m < n) { fprintf(stderr, "%s holds %u values, %u needed\n", xfile, m, n); return 1; }
    CK(mk_vector_set(x, h));
    free(h);
  } else {
    CK(mk_vector_fill(x, 0.0));                                       /* :420-430 */
  }
  if (print_system) {                                                 /* :450-454 */
    SparseMatrixCSR csr;
    CK(mk_matrix_get_csr(A, &csr));
    double *pb = (double *)malloc((size_t)(n ? n : 1) * sizeof(double)), *px = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
    if (pb && px) {
      CK(mk_vector_get(b, pb));
      CK(mk_vector_get(x, px));
      print_system_files(0, 0, &csr, pb, px);
    }
    free(pb); free(px);
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
