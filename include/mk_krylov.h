/*
 * mk_krylov.h -- C ABI of libmi355_krylov.so: the MI355X (gfx950) SpMV + Krylov hot path.
 *
 * This is the drop-in boundary.  The reference (eemorsi/iterative_solvers) has no plugin
 * interface; what its driver src/solvers.c does for solver id 50 is a fixed sequence of
 * HYPRE calls.  Every entry point below names the reference call site (file:line in
 * /root/reference) it stands in for.  INTEGRATION.md shows the patched solvers.c.
 *
 * Conventions
 *   - plain C, opaque handles, no C++/torch types;
 *   - every function returns 0 on success or a negative mk_status; the text of the last
 *     failure is available from mk_last_error_string().  The library never calls exit()
 *     (the reference's loader does: sparse_matrix.c:43-66);
 *   - host pointers are copied during the call and never kept (HYPRE_IJMatrixSetValues
 *     copies too: the driver frees its COO right after, solvers.c:376-378);
 *   - one caller thread per process; one GPU per process (one rank per GPU);
 *   - local indices are int32, global row offsets and global nnz are int64.
 */
#ifndef MK_KRYLOV_H
#define MK_KRYLOV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Host containers of the reference, field for field (sparse_matrix.h:24-44).  If the
 * reference's own header was included first its definitions are used. */
#ifndef SPARSE_MATRIX_H
#define SPARSE_MATRIX_H
struct SparseMatrixCSR_STRUCT {
  double *values;
  int *column_indices;
  int *row_pointers;
  unsigned int nrows;
  unsigned int ncolumns;
  unsigned int nnz;
};
typedef struct SparseMatrixCSR_STRUCT SparseMatrixCSR;
struct SparseMatrixCOO_STRUCT {
  double *values;
  int *rows;
  int *columns;
  unsigned int nrows;
  unsigned int ncolumns;
  unsigned int nnz;
};
typedef struct SparseMatrixCOO_STRUCT SparseMatrixCOO;
#endif

typedef enum {
  MK_OK = 0,
  MK_ERR_INVALID = -1,      /* bad argument (null, size mismatch, index out of range) */
  MK_ERR_HIP = -2,          /* a HIP runtime call failed */
  MK_ERR_NOT_INIT = -3,     /* mk_init() was not called */
  MK_ERR_NO_DEVICE = -4,    /* no gfx950 device visible */
  MK_ERR_ALLOC = -5,        /* host or device allocation failed */
  MK_ERR_COMM = -6,         /* RCCL failure */
  MK_ERR_UNSUPPORTED = -7
} mk_status;

typedef struct mk_matrix mk_matrix;   /* stands in for HYPRE_IJMatrix / HYPRE_ParCSRMatrix (solvers.c:55-56) */
typedef struct mk_vector mk_vector;   /* HYPRE_IJVector / HYPRE_ParVector (solvers.c:57-60) */
typedef struct mk_coo mk_coo;         /* a COO matrix resident in device memory */
typedef struct mk_pcg mk_pcg;         /* HYPRE_Solver for PCG (solvers.c:65,729) */
typedef struct mk_bicgstab mk_bicgstab;   /* HYPRE_Solver for BiCGSTAB (solvers.c:65,524) */
typedef struct mk_cgnr mk_cgnr;           /* HYPRE_Solver for CGNR (solvers.c:65,869) */
typedef struct mk_gmres mk_gmres;         /* HYPRE_Solver for GMRES (solvers.c:65,1189) */
typedef struct mk_jacobi mk_jacobi;   /* extension: plain Jacobi iteration (no reference counterpart) */

/* ---- life cycle: HYPRE_Init() / HYPRE_Finalize() (solvers.c:79,1560) ------------------- */
int mk_init(int device_id);
int mk_finalize(void);
const char *mk_last_error_string(void);
/* the hipStream_t every kernel of this library is launched on (for HIP-event timing) */
void *mk_get_stream(void);
/* "0000:c1:00.0" of a GPU (device_id < 0: the one mk_init selected); host/power.c finds the hwmon power node with it
 * (the reference samples `nvidia-smi -i <id>`, cpower.h:124-130) */
int mk_get_device_pci_bus_id(int device_id, char *out, int capacity);
int mk_synchronize(void);
/* Device memory the library holds (no reference counterpart: the reference leaves memory to HYPRE).  Freed arrays of 1 MB and
 * more stay in a free list for reuse (hipFree costs 0.2 ms a call): in_use = bytes inside live objects, idle = bytes waiting
 * in the list.  mk_memory_trim() gives the idle blocks back to the driver (a long-running host does this after set-up). */
int mk_memory_info(long long *in_use_bytes, long long *idle_bytes);
int mk_memory_trim(void);
/* how often the library has asked its allocator for a block / given one back since mk_init (whether the request went to
 * hipMalloc / hipFree or to the free list): lets a caller check that a code path allocates nothing */
int mk_memory_calls(long long *alloc_calls, long long *free_calls);
/* HIP-event timers on that stream: start, stop, then elapsed (milliseconds, synchronises) */
int mk_timer_start(void);
int mk_timer_stop(float *elapsed_ms);

/* ---- COO -> sorted COO / CSR on the device --------------------------------------------- */
/* sort_coo_row (sparse_matrix.c:419-529, called at solvers.c:198): stable sort by row.
 * out arrays are allocated with aligned_alloc(64, ...) like the reference's; free() them. */
int mk_sort_coo_row(const SparseMatrixCOO *coo, SparseMatrixCOO *sorted);
/* convert_coo_to_csr (sparse_matrix.c:649-742): bit-identical row_pointers, and
 * column_indices / values in stable-by-row order.  out arrays: aligned_alloc(64, ...). */
int mk_coo_to_csr(const SparseMatrixCOO *coo, SparseMatrixCSR *csr);

/* ---- matrix assembly: HYPRE_IJMatrixCreate..Assemble..GetObject (solvers.c:289-374) ----- */
/* whole matrix on this GPU from the host COO the loader produced (sorted or not) */
int mk_matrix_from_coo(const SparseMatrixCOO *coo, mk_matrix **A);
int mk_matrix_from_csr(const SparseMatrixCSR *csr, mk_matrix **A);
int mk_matrix_from_device_coo(mk_coo *coo, mk_matrix **A);  /* consumes coo's buffers */
int mk_matrix_destroy(mk_matrix *A);                         /* HYPRE_IJMatrixDestroy (solvers.c:1555) */
/* copy the device CSR back (tests; HYPRE_IJMatrixPrint's role at solvers.c:451) */
int mk_matrix_get_csr(const mk_matrix *A, SparseMatrixCSR *csr);

typedef struct {
  int64_t nrows, ncols, nnz;
  int32_t max_row_len, min_row_len;
  double mean_row_len;
  int64_t hist[16];        /* row-length histogram: bucket k holds rows with len in [2^(k-1), 2^k), bucket 0 = empty rows */
  int32_t kernel;          /* mk_spmv_kernel chosen from the histogram */
  int32_t lanes_per_row;   /* for the vector kernels */
  int64_t long_rows;       /* rows of >= 2048 entries (always split over chunks by MK_SPMV_SPLIT) */
  int64_t index_patterns;  /* > 0: column indices are served from a table of that many (column - row) patterns
                              (lossless; the stream kernel then reads 2 B per row instead of 4 B per entry) */
  int64_t value_patterns;  /* > 0: the rows of a pattern also share their values bit for bit (constant-coefficient
                              operators); MK_SPMV_PATTERN then serves offsets AND values from the table */
  int32_t stream_lanes;    /* MK_SPMV_STREAM: lanes per row; a chunk is 256 / stream_lanes consecutive rows */
  int32_t stream_chunk;    /* ... and the entries of its fullest chunk (what the kernel stages in LDS) */
  int64_t shared_columns;  /* > 0: that many entries sit in rows that repeat the column list of the row before them (several
                              unknowns per mesh node); MK_SPMV_STREAM reads such a list once (lossless, same order) */
  int64_t row_groups;      /* > 0: MK_SPMV_STREAM serves the product group by group (runs of up to 4 consecutive rows with one
                              column list): x is gathered once per list and multiplied with every row of the group */
  int32_t group_lanes;     /* ... with this many lanes per group */
  int32_t band_window;     /* > 0: MK_SPMV_STREAM serves the product with a sliding window of x (this many entries) kept in LDS:
                              rows that stay within a bounded distance of the diagonal (band kernel) */
  int32_t band_strips;     /* > 0: ... as that many diagonal strips (entries grouped by column - row), one band-kernel launch each:
                              rows that reach further than one LDS window */
  int32_t reserved1;
} mk_matrix_info;
int mk_matrix_get_info(const mk_matrix *A, mk_matrix_info *info);

typedef enum {
  MK_SPMV_AUTO = 0,
  MK_SPMV_VECTOR = 1,      /* 2^k lanes per row, k from the mean row length (k=6: wavefront per row) */
  MK_SPMV_STREAM = 2,      /* regular rows: 256/G consecutive rows per chunk staged through LDS, one pass per row group */
  MK_SPMV_SPLIT = 3,       /* skewed rows: nnz-balanced chunks (merge-path style), products staged in LDS, rows split
                              inside and across chunks, carries folded in a fixed order by a second small kernel */
  MK_SPMV_PATTERN = 4,     /* rows drawn from a small table of (offsets, values) patterns: one lane per row, 2 B of
                              matrix data per row; only when mk_matrix_info.value_patterns > 0 */
  MK_SPMV_SWEEP = 5,       /* pattern matrices whose rows are all sub-lists of ONE pattern with offsets a*P + b (grids):
                              a block sweeps plane by plane with the x window of three planes in LDS, so every x entry
                              is loaded once per block instead of once per neighbour */
  MK_SPMV_VSWEEP = 6       /* the same sweep for grid operators with VARIABLE coefficients: column offsets from the pattern
                              table (2 B per row instead of col[]), values streamed from val[] through LDS, x through the
                              window ring; rows of at most 8 entries */
} mk_spmv_kernel;
/* override the selector (lanes_per_row: 0 = keep, else 1..64 power of two) */
int mk_matrix_set_spmv_kernel(mk_matrix *A, int kernel, int lanes_per_row);

/* ---- synthetic operators generated in device memory (bench.py; no reference counterpart:
 *      the reference only reads SuiteSparse files, scripts/bench_all_gpu:2) ---------------- */
/* kind 5: 2D 5-pt (nz ignored), 7: 3D 7-pt, 27: 3D 27-pt; lexicographic, x fastest.
 * rows [row_begin,row_end) of the global operator, global column ids. */
int mk_matrix_generate_stencil(int kind, int nx, int ny, int nz, int64_t row_begin, int64_t row_end, mk_matrix **A);
/* the same operator scaled symmetrically, D A D with d_i in [1, 5) drawn per row from a counter-based hash of
 * (seed, global row): every row has its own coefficients (the constant-coefficient kernels do not apply, the matrix
 * is served from its CSR value array like the authors' SuiteSparse inputs), still symmetric positive definite */
int mk_matrix_generate_stencil_scaled(int kind, int nx, int ny, int nz, uint64_t seed, int64_t row_begin, int64_t row_end, mk_matrix **A);
int mk_matrix_generate_banded(int64_t n, int half_bandwidth, int64_t row_begin, int64_t row_end, mk_matrix **A);
/* R-MAT edge list, unsorted, duplicates kept (counter-based splitmix64; oracle/oracle.c holds the same definition) */
int mk_coo_generate_rmat(int scale, uint64_t nedges, uint64_t seed, double a, double b, double c, mk_coo **coo);
int mk_coo_from_host(const SparseMatrixCOO *coo, mk_coo **out);
int mk_coo_download(const mk_coo *coo, SparseMatrixCOO *out);
int mk_coo_destroy(mk_coo *coo);

/* ---- vectors: HYPRE_IJVectorCreate/SetValues/Assemble (solvers.c:382-436) ---------------- */
int mk_vector_create(int64_t n, mk_vector **v);
int mk_vector_set(mk_vector *v, const double *host_values);   /* IJVectorSetValues: copy in */
int mk_vector_fill(mk_vector *v, double value);               /* b = 1 (solvers.c:404-408), x0 = 0 (:420-426) */
int mk_vector_get(const mk_vector *v, double *host_values);
int mk_vector_destroy(mk_vector *v);
int64_t mk_vector_size(const mk_vector *v);

/* ---- SpMV: hypre_ParCSRMatrixMatvec, y = alpha*A*x + beta*y (inside HYPRE; in-repo stand-in
 *      test/sblas_mv_rd.c:51-69) ----------------------------------------------------------- */
int mk_spmv(const mk_matrix *A, double alpha, const mk_vector *x, double beta, mk_vector *y);
/* the fused kernel the CG loop runs: y = A*x together with <y,x> and <x,x>.  With both
 * outputs NULL the kernel is only enqueued (no read-back, no synchronisation). */
int mk_spmv_dot(const mk_matrix *A, const mk_vector *x, mk_vector *y, double *ytx, double *xtx);

/* mk_spmv_dot without any allocation.  No counterpart in the reference: it is for callers that cannot allocate device
 * memory at the point of the call (HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  All scratch
 * of its kernels (reduction partials, the two results, the split kernel's carry pairs) is a file-scope device array of
 * 592 KB per GPU; the call uses no hipMalloc, hipFree or library-allocated scratch.  Conditions: the matrix has been
 * analysed (mk_matrix_get_info, which allocates, was called once; for split-kernel matrices one mk_spmv / mk_spmv_dot
 * has run); split-kernel matrices hold at most 2^26 entries -- larger ones are rejected with MK_ERR_UNSUPPORTED.
 * Results are bitwise those of mk_spmv_dot.  All calls on one GPU share the array: calls that overlap on one GPU race. */
int mk_spmv_dot_static(const mk_matrix *A, const mk_vector *x, mk_vector *y, double *ytx, double *xtx);

/* ---- PCG: HYPRE_ParCSRPCG* / HYPRE_PCG* (solvers.c:729-775) ------------------------------ */
typedef enum { MK_PRECOND_NONE = 0, MK_PRECOND_JACOBI = 1 } mk_precond;
int mk_pcg_create(mk_pcg **solver);                               /* HYPRE_ParCSRPCGCreate :729 */
int mk_pcg_set_max_iter(mk_pcg *solver, int max_iter);            /* HYPRE_PCGSetMaxIter :732 */
int mk_pcg_set_tol(mk_pcg *solver, double tol);                   /* HYPRE_PCGSetTol :734 */
int mk_pcg_set_two_norm(mk_pcg *solver, int two_norm);            /* HYPRE_PCGSetTwoNorm :735 */
int mk_pcg_set_rel_change(mk_pcg *solver, int rel_change);        /* HYPRE_PCGSetRelChange :737 */
int mk_pcg_set_print_level(mk_pcg *solver, int level);            /* HYPRE_PCGSetPrintLevel :742 */
int mk_pcg_set_logging(mk_pcg *solver, int logging);              /* HYPRE_PCGSetLogging :744 */
int mk_pcg_set_precond(mk_pcg *solver, int precond);              /* extension id 51: HYPRE_ParCSRDiagScale semantics */
int mk_pcg_set_check_every(mk_pcg *solver, int iterations);       /* how often the host reads the device-side stop flag */
/* Row-block solves only (mk_dist_pcg_solve), opt-in, default off.  The loop of hypre_PCGSolve that solvers.c:753-754 runs
 * under mpirun reduces twice per iteration (<A p, p> in front of alpha, <r,z> in front of beta: two MPI_Allreduce inside
 * HYPRE).  With on = 1 the step needs ONE all-reduce of five doubles: <A p', p'> for p' = z + beta p comes from <A z, z>
 * by the Chronopoulos-Gear identity, <A z, z> - beta gamma' / alpha_old, so it is known before p' is formed.  Same
 * iterates in exact arithmetic, same stopping rule; in floating point alpha differs in its last bits (tests: iteration
 * counts within 2, residual history to 1e-8 against the two-reduction loop).  Taken only when every rank's diagonal block
 * runs on the sweep kernel with no or a uniform-diagonal Jacobi preconditioner (the ranks agree on that at the start of the
 * solve); otherwise the default step runs.  mk_pcg_get_single_reduction reports what the last row-block solve took. */
int mk_pcg_set_single_reduction(mk_pcg *solver, int on);
int mk_pcg_get_single_reduction(const mk_pcg *solver, int *used);
/* measurement hook (bench.py's roofline figures; SURVEY.md 8(d) "hipEvent around the iterations"): with on = 1 the next
 * mk_pcg_solve records a HIP event in front of every kernel of its iterations 2..33 on the library's stream;
 * get returns, per kernel slot of a step (0: SpMV + dots, 1: x/r update, 2: direction update when it is a kernel of
 * its own), the average duration in milliseconds and the number of launches it was taken over (capacity >= 3) */
int mk_pcg_set_kernel_timing(mk_pcg *solver, int on);
int mk_pcg_get_kernel_timing(const mk_pcg *solver, int *nslots, double *avg_ms, long long *launches, int capacity);
int mk_pcg_setup(mk_pcg *solver, const mk_matrix *A, const mk_vector *b, const mk_vector *x);   /* HYPRE_ParCSRPCGSetup :747 */
int mk_pcg_solve(mk_pcg *solver, const mk_matrix *A, const mk_vector *b, mk_vector *x);         /* HYPRE_ParCSRPCGSolve :754 */
int mk_pcg_get_num_iterations(const mk_pcg *solver, long long *iterations);                    /* HYPRE_PCGGetNumIterations :765 */
int mk_pcg_get_final_relative_residual_norm(const mk_pcg *solver, double *norm);               /* :766 */
int mk_pcg_get_converged(const mk_pcg *solver, int *converged, int *breakdown);
/* relative residual norms 0..iterations (needs logging>0); returns how many were written */
int mk_pcg_get_residual_history(const mk_pcg *solver, double *norms, int capacity);
int mk_pcg_destroy(mk_pcg *solver);                               /* HYPRE_ParCSRPCGDestroy :775 */

/* ---- BiCGSTAB, solver id 6 (solvers.c:517-564), identity preconditioner -------------------
 * HYPRE_ParCSRBiCGSTABCreate :524, SetPrintLevel :527, SetLogging :529, SetTol :532, SetMaxIter :533,
 * Setup :536, Solve :542, GetNumIterations :553, GetFinalRelativeResidualNorm :554, Destroy :563.
 * Stops when ||r|| <= tol * ||b|| holds for the recomputed residual b - A x as well (hypre's rule);
 * breakdown: 1 <r0,Ap> ~ 0, 2 <r0,r> ~ 0, 3 gamma ~ 0 (|.| < 1e-128). */
int mk_bicgstab_create(mk_bicgstab **solver);
int mk_bicgstab_set_max_iter(mk_bicgstab *solver, int max_iter);
int mk_bicgstab_set_tol(mk_bicgstab *solver, double tol);
int mk_bicgstab_set_print_level(mk_bicgstab *solver, int level);
int mk_bicgstab_set_logging(mk_bicgstab *solver, int logging);
int mk_bicgstab_set_check_every(mk_bicgstab *solver, int iterations);   /* host polls the device stop flag this often */
int mk_bicgstab_setup(mk_bicgstab *solver, const mk_matrix *A, const mk_vector *b, const mk_vector *x);
int mk_bicgstab_solve(mk_bicgstab *solver, const mk_matrix *A, const mk_vector *b, mk_vector *x);
int mk_bicgstab_get_num_iterations(const mk_bicgstab *solver, long long *iterations);
int mk_bicgstab_get_final_relative_residual_norm(const mk_bicgstab *solver, double *norm);
int mk_bicgstab_get_converged(const mk_bicgstab *solver, int *converged, int *breakdown);
/* absolute ||r|| after each iteration (needs logging > 0 at setup); returns the number of entries written */
int mk_bicgstab_get_residual_history(const mk_bicgstab *solver, double *norms, int capacity);
int mk_bicgstab_destroy(mk_bicgstab *solver);

/* ---- CGNR, solver id 40 (solvers.c:864-906), identity preconditioner ------------------------
 * HYPRE_ParCSRCGNRCreate :869, HYPRE_CGNRSetMaxIter :872, SetTol :873, SetLogging :874, HYPRE_ParCSRCGNRSetup :877,
 * Solve :884, HYPRE_CGNRGetNumIterations :895, GetFinalRelativeResidualNorm :896, Destroy :905.
 * CG on A^T A x = A^T b; setup builds A^T on the device.  Stops when <r,r> < tol^2 <b,b> holds for the recomputed
 * residual b - A x as well (hypre's rule). */
int mk_cgnr_create(mk_cgnr **solver);
int mk_cgnr_set_max_iter(mk_cgnr *solver, int max_iter);
int mk_cgnr_set_tol(mk_cgnr *solver, double tol);
int mk_cgnr_set_logging(mk_cgnr *solver, int logging);
int mk_cgnr_set_check_every(mk_cgnr *solver, int iterations);
int mk_cgnr_setup(mk_cgnr *solver, const mk_matrix *A, const mk_vector *b, const mk_vector *x);
int mk_cgnr_solve(mk_cgnr *solver, const mk_matrix *A, const mk_vector *b, mk_vector *x);
int mk_cgnr_get_num_iterations(const mk_cgnr *solver, long long *iterations);
int mk_cgnr_get_final_relative_residual_norm(const mk_cgnr *solver, double *norm);
int mk_cgnr_get_converged(const mk_cgnr *solver, int *converged);
/* ||r|| / ||b|| after each iteration (needs logging > 0 at setup); returns the number of entries written */
int mk_cgnr_get_residual_history(const mk_cgnr *solver, double *norms, int capacity);
int mk_cgnr_destroy(mk_cgnr *solver);

/* ---- restarted GMRES, solver id 12 (solvers.c:1183-1245), identity preconditioner -----------
 * HYPRE_ParCSRGMRESCreate :1189, SetMaxIter :1191, SetKDim :1192 (restart = 30, :460), SetTol :1193, SetPrintLevel :1195,
 * SetLogging :1196, Setup :1200, Solve :1217, HYPRE_GMRESGetNumIterations :1231, GetFinalRelativeResidualNorm :1232.
 * k_dim <= 62.  Stops when the residual estimate and then the recomputed residual b - A x are <= tol * ||b||. */
int mk_gmres_create(mk_gmres **solver);
int mk_gmres_set_max_iter(mk_gmres *solver, int max_iter);
int mk_gmres_set_k_dim(mk_gmres *solver, int k_dim);
int mk_gmres_set_tol(mk_gmres *solver, double tol);
int mk_gmres_set_print_level(mk_gmres *solver, int level);
int mk_gmres_set_logging(mk_gmres *solver, int logging);
int mk_gmres_setup(mk_gmres *solver, const mk_matrix *A, const mk_vector *b, const mk_vector *x);
int mk_gmres_solve(mk_gmres *solver, const mk_matrix *A, const mk_vector *b, mk_vector *x);
int mk_gmres_get_num_iterations(const mk_gmres *solver, long long *iterations);
int mk_gmres_get_final_relative_residual_norm(const mk_gmres *solver, double *norm);
int mk_gmres_get_converged(const mk_gmres *solver, int *converged);
/* absolute residual estimate after each iteration (needs logging > 0); returns the number of entries written */
int mk_gmres_get_residual_history(const mk_gmres *solver, double *norms, int capacity);
int mk_gmres_destroy(mk_gmres *solver);

/* ---- plain Jacobi iteration, extension solver id 70 --------------- */
int mk_jacobi_create(mk_jacobi **solver);
int mk_jacobi_set_max_iter(mk_jacobi *solver, int max_iter);
int mk_jacobi_set_tol(mk_jacobi *solver, double tol);
int mk_jacobi_set_check_every(mk_jacobi *solver, int iterations);
int mk_jacobi_setup(mk_jacobi *solver, const mk_matrix *A, const mk_vector *b, const mk_vector *x);
int mk_jacobi_solve(mk_jacobi *solver, const mk_matrix *A, const mk_vector *b, mk_vector *x);
int mk_jacobi_get_num_iterations(const mk_jacobi *solver, long long *iterations);
int mk_jacobi_get_final_relative_residual_norm(const mk_jacobi *solver, double *norm);
int mk_jacobi_get_converged(const mk_jacobi *solver, int *converged);
int mk_jacobi_destroy(mk_jacobi *solver);

/* ---- row-block distribution over GPUs: one rank per GPU per process, RCCL over xGMI ---------
 * Stands in for the MPI side of the reference: the row-block split of solvers.c:214-281, the
 * HYPRE ParCSR diag/offd layout behind HYPRE_IJMatrixAssemble (solvers.c:362) and the halo
 * exchange / MPI_Allreduce inside HYPRE's matvec and inner products.  Unlike solvers.c:263-265
 * the whole matrix is never replicated: each rank hands in its own rows only. */
typedef struct mk_comm mk_comm;
typedef struct mk_dist_matrix mk_dist_matrix;
/* rank 0 draws 256 opaque bytes (two ncclUniqueId) and passes them to every rank by any
 * means (MPI_Bcast in solvers.c, torch.distributed in bench.py); then all ranks call create */
int mk_comm_get_unique_id(void *id256);
int mk_comm_create(const void *id256, int nranks, int rank, mk_comm **comm);   /* MPI_Init + MPI_Comm_rank/size (solvers.c:68-70) */
/* several logical ranks inside ONE process on one GPU (tests): exchange = device copies */
int mk_comm_create_local(int nranks, mk_comm **comms);
/* collective check of every RCCL call the solver relies on (ring send/recv, all-gather, all-reduce) */
int mk_comm_self_test(mk_comm *comm);
/* MPI_Allreduce of a few host doubles (solvers.c:760: the solve time, MPI_MAX).  op: 0 sum, 1 max.  Collective;
 * a local world or a NULL communicator leaves the values as they are. */
int mk_comm_all_reduce_host(mk_comm *comm, double *values, int count, int op);
int mk_comm_destroy(mk_comm *comm);
/* A holds rows [row_start[rank], row_start[rank+1]) of the global operator with GLOBAL column
 * ids (ncols = n_global) and is consumed.  Collective. comm may be NULL for a single rank. */
int mk_dist_matrix_create(mk_matrix *A, int64_t n_global, const int64_t *row_start, mk_comm *comm, mk_dist_matrix **out);
int mk_dist_matrix_destroy(mk_dist_matrix *A);
int mk_dist_matrix_get_info(const mk_dist_matrix *A, int64_t *n_local, int64_t *n_ghost, int64_t *n_offd_rows,
                            int64_t *nnz_diag, int64_t *nnz_offd, int *n_neighbors);
/* mk_matrix_set_spmv_kernel for the diag block (the off-rank block has its own small kernel) */
int mk_dist_matrix_set_spmv_kernel(mk_dist_matrix *A, int kernel, int lanes_per_row);
/* mk_matrix_get_info of the diag block (which SpMV kernel the selector took, pattern counts) */
int mk_dist_matrix_get_diag_info(mk_dist_matrix *A, mk_matrix_info *info);
/* vector of n_local owned entries with room for the ghost entries behind them */
int mk_dist_vector_create(const mk_dist_matrix *A, mk_vector **v);
/* nranks_here = 1 over RCCL, = world size for a local world (arrays in rank order) */
int mk_dist_spmv(int nranks_here, mk_dist_matrix **A, mk_vector **x, mk_vector **y);
/* HYPRE_ParCSRPCGSetup + HYPRE_ParCSRPCGSolve (solvers.c:747-754) on the distributed operator;
 * settings and results through the mk_pcg_set_* / mk_pcg_get_* calls above */
int mk_dist_pcg_solve(int nranks_here, mk_pcg **solvers, mk_dist_matrix **A, mk_vector **b, mk_vector **x);
/* plain Jacobi (extension id 70) on the distributed operator; residual all-reduced every check_every iterations */
int mk_dist_jacobi_solve(int nranks_here, mk_jacobi **solvers, mk_dist_matrix **A, mk_vector **b, mk_vector **x);
/* HYPRE_ParCSRBiCGSTABSetup + Solve (solvers.c:517-564) and HYPRE_ParCSRGMRESSetup + Solve (solvers.c:1183-1245) on the
 * distributed operator; no setup call (the work vectors live for the solve); settings and results through the
 * mk_bicgstab_* / mk_gmres_* setters and getters */
int mk_dist_bicgstab_solve(int nranks_here, mk_bicgstab **solvers, mk_dist_matrix **A, mk_vector **b, mk_vector **x);
int mk_dist_gmres_solve(int nranks_here, mk_gmres **solvers, mk_dist_matrix **A, mk_vector **b, mk_vector **x);
/* HYPRE_ParCSRCGNRSetup + Solve (solvers.c:864-906) on the distributed operator: A^T r through the reverse exchange (the
 * transposed blocks are built at the first call and live with the matrix) */
int mk_dist_cgnr_solve(int nranks_here, mk_cgnr **solvers, mk_dist_matrix **A, mk_vector **b, mk_vector **x);

/* ---- exchange planning, host only (no GPU): the host half of the ParCSR communication package that
 *      HYPRE_IJMatrixAssemble builds (solvers.c:362); mk_dist_matrix_create runs exactly these -------------- */
int mk_halo_check_row_start(int nranks, const int64_t *row_start, int64_t n_global);
/* sorted distinct global column ids outside [row_begin,row_end) of a CSR block with global column ids;
 * ghosts may be NULL to query the count */
int mk_halo_ghost_columns(int64_t nrows, const int *rowptr, const int *col, int64_t row_begin, int64_t row_end, int64_t n_global,
                          int *ghosts, int64_t capacity, int64_t *n_ghost);
/* per owner rank: how many of the (sorted) ghosts it owns and where their run starts (nranks / nranks + 1 values) */
int mk_halo_plan_recv(int nranks, int rank, const int64_t *row_start, const int *ghosts, int64_t n_ghost, int *recv_cnt, int *recv_off);
/* recv_cnt_all[q * nranks + r] = what q receives from r (all-gathered); out: what `rank` sends to each q */
int mk_halo_plan_send(int nranks, int rank, const int *recv_cnt_all, int *send_cnt, int *send_off);

/* ---- row-block partition: solvers.c:214-231,270-281 with defects Q1-Q4 repaired ----------- */
/* host-only; rowptr64 has nrows+1 entries; row_start gets nparts+1 entries */
int mk_partition_rows(int64_t nrows, const int64_t *rowptr64, int nparts, int64_t *row_start);

#ifdef __cplusplus
}
#endif
#endif /* MK_KRYLOV_H */
