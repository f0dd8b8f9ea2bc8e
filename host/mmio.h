/*
 * mmio.h -- the part of the NIST Matrix Market I/O interface the reference's loader uses
 * (src/mmio.h: mm_read_banner, mm_read_mtx_crd_size, mm_read_mtx_array_size,
 * mm_typecode_to_str and the mm_is_* predicates), written from the format description at
 * https://math.nist.gov/MatrixMarket/formats.html.  Sizes are unsigned int as in the
 * reference's variant (src/mmio.h:22-23).
 */
#ifndef MK_HOST_MMIO_H
#define MK_HOST_MMIO_H
#include <stdio.h>

#define MM_MAX_LINE_LENGTH 1025
#define MM_MAX_TOKEN_LENGTH 64
#define MatrixMarketBanner "%%MatrixMarket"

typedef char MM_typecode[4];   /* [0] object M, [1] C(oordinate)/A(rray), [2] R/C/P/I, [3] G/S/K/H */

#define mm_is_matrix(t) ((t)[0] == 'M')
#define mm_is_sparse(t) ((t)[1] == 'C')
#define mm_is_coordinate(t) ((t)[1] == 'C')
#define mm_is_dense(t) ((t)[1] == 'A')
#define mm_is_array(t) ((t)[1] == 'A')
#define mm_is_complex(t) ((t)[2] == 'C')
#define mm_is_real(t) ((t)[2] == 'R')
#define mm_is_pattern(t) ((t)[2] == 'P')
#define mm_is_integer(t) ((t)[2] == 'I')
#define mm_is_symmetric(t) ((t)[3] == 'S')
#define mm_is_general(t) ((t)[3] == 'G')
#define mm_is_skew(t) ((t)[3] == 'K')
#define mm_is_hermitian(t) ((t)[3] == 'H')

#define MM_COULD_NOT_READ_FILE 11
#define MM_PREMATURE_EOF 12
#define MM_NOT_MTX 13
#define MM_NO_HEADER 14
#define MM_UNSUPPORTED_TYPE 15
#define MM_LINE_TOO_LONG 16
#define MM_COULD_NOT_WRITE_FILE 17

#ifdef __cplusplus
extern "C" {
#endif
int mm_read_banner(FILE *f, MM_typecode *matcode);
int mm_read_mtx_crd_size(FILE *f, unsigned int *M, unsigned int *N, unsigned int *nz);
int mm_read_mtx_array_size(FILE *f, unsigned int *M, unsigned int *N);
int mm_write_banner(FILE *f, MM_typecode matcode);
int mm_write_mtx_crd_size(FILE *f, unsigned int M, unsigned int N, unsigned int nz);
char *mm_typecode_to_str(MM_typecode matcode);   /* static buffer, do not free */
int mm_is_valid(MM_typecode matcode);
#ifdef __cplusplus
}
#endif
#endif
