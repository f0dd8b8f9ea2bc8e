/*
 * sparse_matrix.h -- host API of the driver: the reference's containers and entry points
 * (src/sparse_matrix.h:24-66) with the heavy lifting moved into libmi355_krylov.so.
 *   fast_load_from_mtx_file   : host text parser (sparse_matrix.c:228-376 semantics)
 *   fast_load_from_array_file : vector files for -b / -x (sparse_matrix.c:34-90)
 *   sort_coo_row              : device stable sort (mk_sort_coo_row)
 *   convert_coo_to_csr        : device conversion (mk_coo_to_csr)
 * Unlike the reference these return an error code instead of calling exit().
 */
#ifndef MK_HOST_SPARSE_MATRIX_H
#define MK_HOST_SPARSE_MATRIX_H
#include "../include/mk_krylov.h" /* SparseMatrixCOO / SparseMatrixCSR */

#ifndef align_size
#define align_size 64 /* the reference builds with -Dalign_size=64 on x86/GPU (src/Makefile_gpu) */
#endif

#ifdef __cplusplus
extern "C" {
#endif
int fast_load_from_mtx_file(const char *mtx_filepath, SparseMatrixCOO *coo_matrix);
int fast_load_from_array_file(const char *mtx_filepath, double **vals, const int ext_factor);   /* the reference's signature; ext_factor must be 1 */
int load_array_file(const char *mtx_filepath, double **vals, unsigned int *count);               /* the same, also returning the number of values */
int sort_coo_row(const SparseMatrixCOO *coo_matrix, SparseMatrixCOO *s_coo_matrix);
int convert_coo_to_csr(const SparseMatrixCOO *coo_matrix, SparseMatrixCSR *csr_matrix, int free_coo);
void free_coo(SparseMatrixCOO *m);
void free_csr(SparseMatrixCSR *m);
size_t get_multiple_of_align_size(size_t size);
/* binary CSR cache beside the .mtx (extension) */
int write_csr_cache(const char *path, const SparseMatrixCSR *m);
int read_csr_cache(const char *path, SparseMatrixCSR *m);
/* the same, remembering / checking size and modification time of the .mtx the cache was made from */
int write_csr_cache_for(const char *path, const SparseMatrixCSR *m, const char *src_mtx);
int read_csr_cache_for(const char *path, SparseMatrixCSR *m, const char *src_mtx);
/* row blocks for np > 1 (solvers.c:214-281): row pointers of an unsorted COO, and one rank's entries */
int coo_row_pointers64(const SparseMatrixCOO *coo, long long *rowptr);
int slice_coo_rows(const SparseMatrixCOO *coo, long long row_lo, long long row_hi, SparseMatrixCOO *out);
int write_stencil_mtx(const char *path, int kind, int nx, int ny, int nz);   /* synthetic test input */
/* -id_file: the communicator id from rank 0 to the others through a shared directory (idfile.c; stands in for MPI_Bcast) */
void mk_idfile_nonce(char *nonce64);
void mk_idfile_clear(const char *path);
int mk_idfile_publish(const char *path, const unsigned char *id256);
int mk_idfile_wait(const char *path, unsigned char *id256, double timeout_s, double max_age_s);
#ifdef __cplusplus
}
#endif
#endif
