/* mmio.c -- Matrix Market banner / size-line reader (see mmio.h). */
#include "mmio.h"

#include <ctype.h>
#include <string.h>

static void lower(char *s) {
  for (; *s; ++s) *s = (char)tolower((unsigned char)*s);
}

int mm_is_valid(MM_typecode t) {
  if (!mm_is_matrix(t)) return 0;
  if (mm_is_dense(t) && mm_is_pattern(t)) return 0;
  if (mm_is_real(t) && mm_is_hermitian(t)) return 0;
  if (mm_is_pattern(t) && (mm_is_hermitian(t) || mm_is_skew(t))) return 0;
  return 1;
}

int mm_read_banner(FILE *f, MM_typecode *matcode) {
  char line[MM_MAX_LINE_LENGTH];
  char banner[MM_MAX_TOKEN_LENGTH], object[MM_MAX_TOKEN_LENGTH], format[MM_MAX_TOKEN_LENGTH];
  char field[MM_MAX_TOKEN_LENGTH], symmetry[MM_MAX_TOKEN_LENGTH];
  char *t = *matcode;
  t[0] = t[1] = t[2] = ' ';
  t[3] = 'G';
  if (fgets(line, sizeof line, f) == NULL) return MM_PREMATURE_EOF;
  if (sscanf(line, "%63s %63s %63s %63s %63s", banner, object, format, field, symmetry) != 5) return MM_PREMATURE_EOF;
  lower(object); lower(format); lower(field); lower(symmetry);
  if (strncmp(banner, MatrixMarketBanner, strlen(MatrixMarketBanner)) != 0) return MM_NO_HEADER;
  if (strcmp(object, "matrix") != 0) return MM_UNSUPPORTED_TYPE;
  t[0] = 'M';
  if (strcmp(format, "coordinate") == 0) t[1] = 'C';
  else if (strcmp(format, "array") == 0) t[1] = 'A';
  else return MM_UNSUPPORTED_TYPE;
  if (strcmp(field, "real") == 0) t[2] = 'R';
  else if (strcmp(field, "complex") == 0) t[2] = 'C';
  else if (strcmp(field, "pattern") == 0) t[2] = 'P';
  else if (strcmp(field, "integer") == 0) t[2] = 'I';
  else return MM_UNSUPPORTED_TYPE;
  if (strcmp(symmetry, "general") == 0) t[3] = 'G';
  else if (strcmp(symmetry, "symmetric") == 0) t[3] = 'S';
  else if (strcmp(symmetry, "hermitian") == 0) t[3] = 'H';
  else if (strcmp(symmetry, "skew-symmetric") == 0) t[3] = 'K';
  else return MM_UNSUPPORTED_TYPE;
  return 0;
}

/* skip comment lines (leading '%') and blank lines, leave the first data line in `line` */
static int next_data_line(FILE *f, char *line, int cap) {
  do {
    if (fgets(line, cap, f) == NULL) return MM_PREMATURE_EOF;
  } while (line[0] == '%' || line[strspn(line, " \t\r\n")] == '\0');
  return 0;
}

int mm_read_mtx_crd_size(FILE *f, unsigned int *M, unsigned int *N, unsigned int *nz) {
  char line[MM_MAX_LINE_LENGTH];
  *M = *N = *nz = 0;
  int rc = next_data_line(f, line, sizeof line);
  if (rc) return rc;
  if (sscanf(line, "%u %u %u", M, N, nz) != 3) return MM_PREMATURE_EOF;
  return 0;
}

int mm_read_mtx_array_size(FILE *f, unsigned int *M, unsigned int *N) {
  char line[MM_MAX_LINE_LENGTH];
  *M = *N = 0;
  int rc = next_data_line(f, line, sizeof line);
  if (rc) return rc;
  if (sscanf(line, "%u %u", M, N) != 2) return MM_PREMATURE_EOF;
  return 0;
}

char *mm_typecode_to_str(MM_typecode t) {
  static char buf[MM_MAX_LINE_LENGTH];
  const char *fmt = mm_is_sparse(t) ? "coordinate" : mm_is_dense(t) ? "array" : "?";
  const char *fld = mm_is_real(t) ? "real" : mm_is_complex(t) ? "complex" : mm_is_pattern(t) ? "pattern" : mm_is_integer(t) ? "integer" : "?";
  const char *sym = mm_is_general(t) ? "general" : mm_is_symmetric(t) ? "symmetric" : mm_is_hermitian(t) ? "hermitian" : mm_is_skew(t) ? "skew-symmetric" : "?";
  snprintf(buf, sizeof buf, "%s %s %s %s", mm_is_matrix(t) ? "matrix" : "?", fmt, fld, sym);
  return buf;
}

int mm_write_banner(FILE *f, MM_typecode t) {
  return fprintf(f, "%s %s\n", MatrixMarketBanner, mm_typecode_to_str(t)) > 0 ? 0 : MM_COULD_NOT_WRITE_FILE;
}

int mm_write_mtx_crd_size(FILE *f, unsigned int M, unsigned int N, unsigned int nz) {
  return fprintf(f, "%u %u %u\n", M, N, nz) > 0 ? 0 : MM_COULD_NOT_WRITE_FILE;
}
