/*
 * oracle.h -- CPU restatement of the reference's SpMV + Krylov hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The shipped
 * path is the HIP library under iterative_solvers_amd/csrc; nothing there calls into
 * this file and there is no CPU fallback.
 *
 * Parity status
 *   - conversion half (sort_coo_row, convert_coo_to_csr): PINNED.  Checked bit-for-bit
 *     against the reference's own sparse_matrix.c compiled into oracle/_ref/ and against
 *     the fixtures under tests/golden/ that were generated from it.
 *   - SpMV: pinned by the one known-answer input the reference holds
 *     (test/sblas_mv_rd.c:15-32; expected y derived by hand, the file has no outputs).
 *   - PCG / Jacobi: PARITY UNPINNED.  The arithmetic lives in HYPRE (krylov/pcg.c), an
 *     un-vendored, un-versioned dependency (src/Makefile:7) that is absent here.  The
 *     loop below restates HYPRE's published PCG algorithm with the settings the
 *     reference passes at solvers.c:729-744; it is cross-checked against scipy only.
 *
 * All citations are file:line inside /root/reference.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* sparse_matrix.c:471-513 (x86 branch of sort_coo_row): stable counting sort by row. */
void orc_sort_coo_row(uint32_t nnz, uint32_t nrows, const int *rows, const int *cols,
                      const double *vals, int *s_rows, int *s_cols, double *s_vals);

/* sparse_matrix.c:649-700 (convert_coo_to_csr): histogram, exclusive scan, stable scatter
 * with the row pointer as cursor, shift-restore.  No per-row column sort (:714-735). */
void orc_coo_to_csr(uint32_t nnz, uint32_t nrows, const int *rows, const int *cols,
                    const double *vals, int *row_pointers, int *column_indices,
                    double *values);

/* solvers.c:214-231,270-281: nnz-balanced contiguous row blocks, with defects Q1-Q4 of
 * SURVEY.md 3.4 repaired.  row_start has np+1 entries; part i owns rows
 * [row_start[i], row_start[i+1]).  rowptr may be 64-bit (config 4 needs it). */
void orc_partition_rows(uint32_t nrows, const int64_t *rowptr64, int np, int64_t *row_start);

/* test/sblas_mv_rd.c:51-69 semantics, y = alpha*A*x + beta*y, one row at a time, products
 * summed left to right in storage order. */
void orc_csr_spmv(uint32_t nrows, const int *rowptr, const int *col, const double *val,
                  double alpha, const double *x, double beta, double *y);
/* same arithmetic, rows split over OpenMP threads (bench.py cpu_baseline only) */
void orc_csr_spmv_omp(uint32_t nrows, const int *rowptr, const int *col, const double *val,
                      double alpha, const double *x, double beta, double *y);

/* diagonal of a CSR matrix: first stored entry with col==row, 0.0 if absent */
void orc_csr_diag(uint32_t nrows, const int *rowptr, const int *col, const double *val,
                  double *diag);

typedef struct {
  long long iterations;   /* HYPRE_PCGGetNumIterations (solvers.c:765) */
  double rel_res_norm;    /* HYPRE_PCGGetFinalRelativeResidualNorm (solvers.c:766) */
  int converged;          /* 1 = passed all convergence tests */
  int breakdown;          /* 0 none, 1 zero <s,p>, 2 subnormal alpha, 3 subnormal gamma */
} orc_solve_info;

/* PCG with the settings of solvers.c:729-744: two_norm=1, rel_change=1, a_tol=0, no
 * residual recomputation.  precond: 0 identity (solver id 50), 1 diagonal scaling
 * (extension id 51).  rel_norms may be NULL or hold maxit+1 doubles.  use_omp selects
 * the threaded loops (same order of operations per element; dot products then sum in
 * per-thread chunks, so bits may differ from the serial run). */
void orc_pcg(uint32_t n, const int *rowptr, const int *col, const double *val,
             const double *b, double *x, int precond, double tol, int maxit,
             int two_norm, int rel_change, int use_omp, orc_solve_info *info,
             double *rel_norms);

/* plain Jacobi (extension id 70): x <- x + D^-1 (b - A x); stop when
 * ||b - A x_k||_2 / ||b||_2 < tol, tested every check_every iterations. */
void orc_jacobi(uint32_t n, const int *rowptr, const int *col, const double *val,
                const double *b, double *x, double tol, int maxit, int check_every,
                int use_omp, orc_solve_info *info);

/* BiCGSTAB as solvers.c:517-564 drives it (solver id 6: tol, maxit, identity preconditioner); breakdown:
 * 1 <r0,q> ~ 0, 2 res ~ 0, 3 gamma ~ 0.  norms may be NULL or hold maxit+1 doubles (absolute ||r||). */
void orc_bicgstab(uint32_t n, const int *rowptr, const int *col, const double *val, const double *b,
                  double *x, double tol, int maxit, int use_omp, orc_solve_info *info, double *norms);

/* CGNR as solvers.c:864-906 drives it (solver id 40: maxit, tol, identity preconditioner).  rel_norms may be
 * NULL or hold maxit+1 doubles (||r||/||b||). */
void orc_cgnr(uint32_t n, const int *rowptr, const int *col, const double *val, const double *b,
              double *x, double tol, int maxit, int use_omp, orc_solve_info *info, double *rel_norms);

/* GMRES(k_dim) as solvers.c:1183-1245 drives it (solver id 12: maxit, k_dim = 30, tol, identity preconditioner).
 * norms may be NULL or hold maxit+1 doubles (absolute residual estimates). */
void orc_gmres(uint32_t n, const int *rowptr, const int *col, const double *val, const double *b,
               double *x, double tol, int maxit, int k_dim, int use_omp, orc_solve_info *info, double *norms);

/* ---- synthetic matrices (bench.py / tests; same definitions as the device generators) */
uint64_t orc_splitmix64(uint64_t x);
/* R-MAT edge i of a 2^scale graph: row/col bits drawn from counter-based splitmix64 */
void orc_rmat_coo(int scale, uint64_t nedges, uint64_t seed, double a, double b, double c,
                  int *rows, int *cols, double *vals);
/* stencil CSR, lexicographic x fastest: kind 5 (2D 5-pt), 7 (3D 7-pt), 27 (3D 27-pt).
 * returns nnz; pass NULL arrays to count only. */
int64_t orc_stencil_csr(int kind, int nx, int ny, int nz, int *rowptr, int *col, double *val);
/* variable coefficients: D A D with the per-row scale of generate.hip row_scale (13-bit dyadic rational in [1, 5)) */
void orc_row_scale(uint64_t seed, int64_t row_begin, int64_t n, double *s);
void orc_csr_scale_sym(uint32_t n, const int *rowptr, const int *col, double *val, const double *s);
int orc_num_threads(void);
/* banded matrix, half bandwidth hb, diagonal 2*hb+1, off-diagonals -1 */
int64_t orc_banded_csr(int64_t n, int hb, int *rowptr, int *col, double *val);

#ifdef __cplusplus
}
#endif
#endif
