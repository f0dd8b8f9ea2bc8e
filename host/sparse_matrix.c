/* sparse_matrix.c -- host loader of the driver (see sparse_matrix.h). */
#include "sparse_matrix.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mmio.h"
#ifdef _OPENMP
#include <omp.h>
#endif

size_t get_multiple_of_align_size(size_t size) { return (size + align_size - 1) / align_size * align_size; }

static void *aalloc(size_t bytes) {
  if (bytes == 0) bytes = align_size;
  return aligned_alloc(align_size, get_multiple_of_align_size(bytes));
}

void free_coo(SparseMatrixCOO *m) {
  free(m->values); free(m->rows); free(m->columns);
  m->values = NULL; m->rows = NULL; m->columns = NULL;
}

void free_csr(SparseMatrixCSR *m) {
  free(m->values); free(m->column_indices); free(m->row_pointers);
  m->values = NULL; m->column_indices = NULL; m->row_pointers = NULL;
}

/* ---- entry parsing ---------------------------------------------------------------------
 * Reference semantics kept (sparse_matrix.c:228-376):
 *  - coordinate files with real or pattern field only (:250-255);
 *  - indices turn 0-based (:297-299); pattern entries get 1.0 (:322,367);
 *  - symmetric: an off-diagonal (r,c,v) is followed at once by (c,r,v); the arrays are
 *    sized for 2*entries (:279-312);
 *  - nothing is filtered or merged: explicit zeros and duplicates stay (:372);
 *  - exactly `entries` entries are read, whatever follows is ignored.
 * Differences: errors are returned, not exit()ed; a short file is an error instead of a
 * read past the buffer; and the text is parsed by all host cores (once the
 * solve runs on the GPU, strtoul/strtod over the slurped file is the whole wall time). */

/* the reference's loop: a whitespace-separated token stream, one thread */
static int parse_serial(const char *path, char *text, unsigned int entries, int sym, int pat, SparseMatrixCOO *coo) {
  unsigned int k = 0;
  char *p = text, *q;
  for (unsigned int i = 0; i < entries; i++) {
    const unsigned long r = strtoul(p, &q, 10);
    if (q == p) { fprintf(stderr, "%s: entry %u of %u missing\n", path, i + 1, entries); return 6; }
    p = q;
    const unsigned long c = strtoul(p, &q, 10);
    if (q == p) { fprintf(stderr, "%s: entry %u truncated\n", path, i + 1); return 6; }
    p = q;
    double v = 1.0;
    if (!pat) {
      v = strtod(p, &q);
      if (q == p) { fprintf(stderr, "%s: entry %u has no value\n", path, i + 1); return 6; }
      p = q;
    }
    coo->rows[k] = (int)(r - 1);
    coo->columns[k] = (int)(c - 1);
    coo->values[k] = v;
    k++;
    if (sym && r != c) {
      coo->rows[k] = (int)(c - 1);
      coo->columns[k] = (int)(r - 1);
      coo->values[k] = v;
      k++;
    }
  }
  coo->nnz = k;
  return 0;
}

static int blank_to_eol(const char *p) {
  while (*p == ' ' || *p == '\t' || *p == '\r') p++;
  return *p == '\n' || *p == '\0';
}

/* One entry per line (what every Matrix Market writer produces): the text is cut at line
 * ends into one piece per thread; pass 1 counts entries (and, for symmetric files, how many
 * are off-diagonal) per piece, pass 2 parses each piece into its slice of the output.  Returns
 * -1 when the text is not one-entry-per-line, and the caller falls back to parse_serial. */
static int parse_parallel(char *text, size_t len, unsigned int entries, int sym, int pat, int nthreads, SparseMatrixCOO *coo) {
  size_t *seg = (size_t *)malloc(sizeof(size_t) * (nthreads + 1));
  long long *n_ent = (long long *)calloc(nthreads + 1, sizeof(long long));
  long long *n_out = (long long *)calloc(nthreads + 1, sizeof(long long));
  int bad = 0;
  seg[0] = 0;
  for (int t = 1; t < nthreads; t++) {
    size_t s = len / nthreads * t;
    if (s < seg[t - 1]) s = seg[t - 1];
    while (s < len && text[s] != '\n') s++;
    seg[t] = s < len ? s + 1 : len;
  }
  seg[nthreads] = len;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1) reduction(| : bad)
  for (int t = 0; t < nthreads; t++) {
    long long ne = 0, no = 0;
    char *p = text + seg[t], *end = text + seg[t + 1], *q;
    while (p < end) {
      char *eol = (char *)memchr(p, '\n', (size_t)(end - p));
      if (!eol) eol = end;
      if (!blank_to_eol(p)) {
        ne++;
        if (sym) {
          const unsigned long r = strtoul(p, &q, 10);
          if (q == p || q > eol) { bad = 1; break; }
          p = q;
          const unsigned long c = strtoul(p, &q, 10);
          if (q == p || q > eol) { bad = 1; break; }
          no += (r != c) ? 2 : 1;
        } else {
          no++;
        }
      }
      p = eol + 1;
    }
    n_ent[t + 1] = ne;
    n_out[t + 1] = no;
  }
  if (!bad) {
    for (int t = 0; t < nthreads; t++) { n_ent[t + 1] += n_ent[t]; }
    if (n_ent[nthreads] < (long long)entries) bad = 1;     /* short, or entries span lines: let the token parser decide */
  }
  if (bad) { free(seg); free(n_ent); free(n_out); return -1; }
  /* output offsets: only the first `entries` entries count, so pieces past that point are cut */
  long long *out_off = (long long *)calloc(nthreads + 1, sizeof(long long));
  int exact = 1;
  for (int t = 0; t < nthreads; t++) {
    if (n_ent[t + 1] > (long long)entries) exact = 0;      /* trailing extra lines: offsets of the cut piece unknown */
    out_off[t + 1] = out_off[t] + n_out[t + 1];
  }
  if (!exact && sym) { free(seg); free(n_ent); free(n_out); free(out_off); return -1; }
#pragma omp parallel for num_threads(nthreads) schedule(static, 1) reduction(| : bad)
  for (int t = 0; t < nthreads; t++) {
    long long i = n_ent[t];                       /* global index of this piece's first entry */
    long long k = out_off[t];
    char *p = text + seg[t], *end = text + seg[t + 1], *q;
    while (p < end && i < (long long)entries) {
      char *eol = (char *)memchr(p, '\n', (size_t)(end - p));
      if (!eol) eol = end;
      if (!blank_to_eol(p)) {
        const unsigned long r = strtoul(p, &q, 10);
        if (q == p || q > eol) { bad = 1; break; }
        p = q;
        const unsigned long c = strtoul(p, &q, 10);
        if (q == p || q > eol) { bad = 1; break; }
        p = q;
        double v = 1.0;
        if (!pat) {
          v = strtod(p, &q);
          if (q == p || q > eol) { bad = 1; break; }
          p = q;
        }
        if (!blank_to_eol(p)) { bad = 1; break; }          /* more tokens on the line: not one entry per line */
        coo->rows[k] = (int)(r - 1);
        coo->columns[k] = (int)(c - 1);
        coo->values[k] = v;
        k++;
        if (sym && r != c) {
          coo->rows[k] = (int)(c - 1);
          coo->columns[k] = (int)(r - 1);
          coo->values[k] = v;
          k++;
        }
        i++;
      }
      p = eol + 1;
    }
    n_out[t + 1] = k - out_off[t];                /* what was really written (the cut piece writes less) */
  }
  long long total = 0;
  for (int t = 0; t < nthreads; t++) {
    if (n_out[t + 1] != out_off[t + 1] - out_off[t] && n_ent[t + 1] <= (long long)entries) bad = 1;
    total += n_out[t + 1];
  }
  free(seg); free(n_ent); free(n_out); free(out_off);
  if (bad) return -1;
  coo->nnz = (unsigned int)total;
  return 0;
}

int fast_load_from_mtx_file(const char *path, SparseMatrixCOO *coo) {
  FILE *f = fopen(path, "r");
  if (!f) { fprintf(stderr, "Could not open file: %s \n", path); return 1; }
  MM_typecode tc;
  if (mm_read_banner(f, &tc) != 0) { fprintf(stderr, "Could not process Matrix Market banner.\n"); fclose(f); return 2; }
  if ((!mm_is_real(tc) && !mm_is_pattern(tc)) || !mm_is_matrix(tc) || !mm_is_sparse(tc)) {
    fprintf(stderr, "Market Market type: [%s] not supported\n", mm_typecode_to_str(tc));
    fclose(f);
    return 3;
  }
  unsigned int nrows, ncols, entries;
  if (mm_read_mtx_crd_size(f, &nrows, &ncols, &entries) != 0) {
    fprintf(stderr, "Error while reading matrix dimension sizes.\n");
    fclose(f);
    return 4;
  }
  const long here = ftell(f);
  fseek(f, 0, SEEK_END);
  const long len = ftell(f) - here;
  fseek(f, here, SEEK_SET);
  char *text = (char *)malloc((size_t)len + 1);
  if (!text) { fclose(f); return 5; }
  const size_t got = fread(text, 1, (size_t)len, f);
  text[got] = '\0';
  fclose(f);

  const int sym = mm_is_symmetric(tc), pat = mm_is_pattern(tc);
  const size_t cap = (size_t)entries * (sym ? 2 : 1);
  coo->nrows = nrows;
  coo->ncolumns = ncols;
  coo->nnz = 0;
  coo->rows = (int *)aalloc(cap * sizeof(int));
  coo->columns = (int *)aalloc(cap * sizeof(int));
  coo->values = (double *)aalloc(cap * sizeof(double));
  if (!coo->rows || !coo->columns || !coo->values) { free(text); free_coo(coo); return 5; }

  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  const char *env_t = getenv("MK_PARSE_THREADS");
  if (env_t) nthreads = atoi(env_t);
  const char *env_min = getenv("MK_PARSE_MIN_ENTRIES");
  const unsigned int min_entries = env_min ? (unsigned int)atol(env_min) : 200000u;
  if (nthreads > 64) nthreads = 64;
  int rc = -1;
  if (nthreads > 1 && entries >= min_entries) rc = parse_parallel(text, got, entries, sym, pat, nthreads, coo);
  const int took_parallel = (rc != -1);
  if (rc == -1) rc = parse_serial(path, text, entries, sym, pat, coo);
  if (getenv("MK_PARSE_TRACE"))   /* tests: which parser read the file */
    fprintf(stderr, "Matrix Market parser: %s (%u entries, %d threads)\n", took_parallel ? "parallel" : "serial", entries, took_parallel ? nthreads : 1);
  free(text);
  if (rc != 0) free_coo(coo);
  return rc;
}

/* sparse_matrix.c:34-90: `m` values, one per line, of a Matrix Market array file (only the
 * first column is meaningful in the reference as well: it ignores n, :72-76).  The validity
 * test uses || where the reference has &&. */
int load_array_file(const char *path, double **vals, unsigned int *count) {
  FILE *f = fopen(path, "r");
  if (!f) { fprintf(stderr, "Could not open file: %s \n", path); return 1; }
  MM_typecode tc;
  if (mm_read_banner(f, &tc) != 0) { fprintf(stderr, "Could not process Matrix Market banner.\n"); fclose(f); return 2; }
  if (!mm_is_real(tc) || !mm_is_array(tc)) {
    fprintf(stderr, "Market Market type: [%s] not array\n", mm_typecode_to_str(tc));
    fclose(f);
    return 3;
  }
  unsigned int m, n;
  if (mm_read_mtx_array_size(f, &m, &n) != 0) { fprintf(stderr, "Error while reading matrix dimension sizes.\n"); fclose(f); return 4; }
  *vals = (double *)aalloc((size_t)m * sizeof(double));
  if (!*vals) { fclose(f); return 5; }
  for (unsigned int i = 0; i < m; i++)
    if (fscanf(f, "%lg", &(*vals)[i]) != 1) { fprintf(stderr, "%s: value %u of %u missing\n", path, i + 1, m); fclose(f); free(*vals); *vals = NULL; return 6; }
  if (count) *count = m;
  fclose(f);
  return 0;
}

/* the reference's own signature (sparse_matrix.h:59, called at solvers.c:206-210 with mtx_ext, which the reference fixes at 1,
 * solvers.c:88): an unpatched call site compiles and links against this library.  ext_factor > 1 asked the reference's
 * loader for its "extend" blow-up (sparse_matrix.c:92-109), a dead feature that is refused here. */
int fast_load_from_array_file(const char *path, double **vals, const int ext_factor) {
  if (ext_factor != 1) { fprintf(stderr, "%s: ext_factor %d is not supported (the reference fixes it at 1)\n", path, ext_factor); return 7; }
  return load_array_file(path, vals, NULL);
}

/* device stable sort by row: sparse_matrix.c:419-529 */
int sort_coo_row(const SparseMatrixCOO *coo, SparseMatrixCOO *sorted) {
  int rc = mk_sort_coo_row(coo, sorted);
  if (rc) fprintf(stderr, "sort_coo_row: %s\n", mk_last_error_string());
  return rc;
}

/* device COO -> CSR: sparse_matrix.c:649-742 */
int convert_coo_to_csr(const SparseMatrixCOO *coo, SparseMatrixCSR *csr, int free_coo_flag) {
  int rc = mk_coo_to_csr(coo, csr);
  if (rc) fprintf(stderr, "convert_coo_to_csr: %s\n", mk_last_error_string());
  if (rc == 0 && free_coo_flag) {
    free(coo->values); free(coo->rows); free(coo->columns);
  }
  return rc;
}

/* Synthetic test input: write a 5/7/27-point stencil as a general real coordinate file, rows
 * ascending, columns ascending inside a row (BASELINE config 1 feeds the solver through the
 * .mtx path).  No reference counterpart. */
int write_stencil_mtx(const char *path, int kind, int nx, int ny, int nz) {
  if (kind == 5) nz = 1;
  FILE *f = fopen(path, "w");
  if (!f) return 1;
  const long long n = (long long)nx * ny * nz;
  long long nnz = 0;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%lld %lld %lld\n", n, n, nnz);
    for (long long row = 0; row < n; row++) {
      const int ix = (int)(row % nx), iy = (int)((row / nx) % ny), iz = (int)(row / ((long long)nx * ny));
      for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
          for (int dx = -1; dx <= 1; dx++) {
            const int man = (dx != 0) + (dy != 0) + (dz != 0);
            if (kind != 27 && man > 1) continue;
            if (kind == 5 && dz != 0) continue;
            const int x = ix + dx, y = iy + dy, z = iz + dz;
            if (x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz) continue;
            if (pass == 0) { nnz++; continue; }
            const long long c = ((long long)z * ny + y) * nx + x;
            const double diag = kind == 27 ? 26.0 : kind == 7 ? 6.0 : 4.0;
            fprintf(f, "%lld %lld %g\n", row + 1, c + 1, man == 0 ? diag : -1.0);
          }
    }
  }
  fclose(f);
  return 0;
}

/* ---- binary CSR cache: the converted matrix beside the .mtx, so that a second
 * run skips both the text parse and the sort.  Layout: 64-byte header, then row_pointers
 * (int32[nrows+1]), column_indices (int32[nnz]), values (f64[nnz]), each padded to 64 bytes.
 * The header remembers size and modification time of the text file the cache was made from:
 * read_csr_cache_for() refuses a cache whose source has changed since.  No reference counterpart. */
#include <sys/stat.h>

typedef struct {
  char magic[8];            /* "MKCSR02\0" */
  unsigned int nrows, ncolumns, nnz, endian_mark;
  unsigned long long checksum;   /* wrap-around sum over row pointers, column indices and the value words */
  unsigned long long src_size;   /* bytes of the source .mtx, 0 = not recorded */
  long long src_mtime_ns;        /* its modification time */
  char pad[16];
} csr_cache_header;

static unsigned long long csr_checksum(const SparseMatrixCSR *m) {
  unsigned long long s = 0;
  for (unsigned int i = 0; i <= m->nrows; i++) s += (unsigned int)m->row_pointers[i];
  for (unsigned int k = 0; k < m->nnz; k++) s += (unsigned int)m->column_indices[k];
  for (unsigned int k = 0; k < m->nnz; k++) {   /* the values as 64-bit words, position-weighted so that swaps show */
    unsigned long long w;
    memcpy(&w, &m->values[k], sizeof w);
    s += w * (2ull * k + 1ull);
  }
  return s;
}

static int source_stamp(const char *src, unsigned long long *size, long long *mtime_ns) {
  struct stat st;
  if (!src || stat(src, &st) != 0) return 1;
  *size = (unsigned long long)st.st_size;
  *mtime_ns = (long long)st.st_mtim.tv_sec * 1000000000ll + (long long)st.st_mtim.tv_nsec;
  return 0;
}

static int write_padded(FILE *f, const void *p, size_t bytes) {
  static const char zeros[64] = {0};
  if (bytes && fwrite(p, 1, bytes, f) != bytes) return 1;
  const size_t padn = (64 - bytes % 64) % 64;
  return padn && fwrite(zeros, 1, padn, f) != padn;
}

/* src may be NULL (no source recorded: the cache is then accepted for any source) */
int write_csr_cache_for(const char *path, const SparseMatrixCSR *m, const char *src) {
  FILE *f = fopen(path, "wb");
  if (!f) return 1;
  csr_cache_header h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "MKCSR02", 8);
  h.nrows = m->nrows; h.ncolumns = m->ncolumns; h.nnz = m->nnz; h.endian_mark = 0x01020304u;
  h.checksum = csr_checksum(m);
  if (src && source_stamp(src, &h.src_size, &h.src_mtime_ns) != 0) { h.src_size = 0; h.src_mtime_ns = 0; }
  int bad = fwrite(&h, sizeof h, 1, f) != 1;
  bad = bad || write_padded(f, m->row_pointers, ((size_t)m->nrows + 1) * sizeof(int));
  bad = bad || write_padded(f, m->column_indices, (size_t)m->nnz * sizeof(int));
  bad = bad || write_padded(f, m->values, (size_t)m->nnz * sizeof(double));
  bad = (fclose(f) != 0) || bad;
  if (bad) remove(path);
  return bad;
}

int write_csr_cache(const char *path, const SparseMatrixCSR *m) { return write_csr_cache_for(path, m, NULL); }

static int read_padded(FILE *f, void *p, size_t bytes) {
  if (bytes && fread(p, 1, bytes, f) != bytes) return 1;
  const long padn = (long)((64 - bytes % 64) % 64);
  return padn && fseek(f, padn, SEEK_CUR) != 0;
}

/* returns 0 and fills *m (aligned_alloc'ed arrays), or non-zero when the file is absent (1), short, from
 * another byte order or format version (2), fails its checksum (3), or was made from a source file whose
 * size or modification time differs from src's now (4; only when src is given and the cache recorded one) */
int read_csr_cache_for(const char *path, SparseMatrixCSR *m, const char *src) {
  FILE *f = fopen(path, "rb");
  if (!f) return 1;
  csr_cache_header h;
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "MKCSR02", 8) != 0 || h.endian_mark != 0x01020304u) { fclose(f); return 2; }
  if (src && h.src_size != 0) {
    unsigned long long size = 0;
    long long mtime_ns = 0;
    if (source_stamp(src, &size, &mtime_ns) != 0 || size != h.src_size || mtime_ns != h.src_mtime_ns) { fclose(f); return 4; }
  }
  m->nrows = h.nrows; m->ncolumns = h.ncolumns; m->nnz = h.nnz;
  m->row_pointers = (int *)aalloc(((size_t)h.nrows + 1) * sizeof(int));
  m->column_indices = (int *)aalloc((size_t)h.nnz * sizeof(int));
  m->values = (double *)aalloc((size_t)h.nnz * sizeof(double));
  int bad = !m->row_pointers || !m->column_indices || !m->values;
  bad = bad || read_padded(f, m->row_pointers, ((size_t)h.nrows + 1) * sizeof(int));
  bad = bad || read_padded(f, m->column_indices, (size_t)h.nnz * sizeof(int));
  bad = bad || read_padded(f, m->values, (size_t)h.nnz * sizeof(double));
  fclose(f);
  if (!bad) bad = (m->row_pointers[0] != 0) || ((unsigned int)m->row_pointers[h.nrows] != h.nnz) || (csr_checksum(m) != h.checksum);
  if (bad) { free_csr(m); return 3; }
  return 0;
}

int read_csr_cache(const char *path, SparseMatrixCSR *m) { return read_csr_cache_for(path, m, NULL); }

/* ---- row blocks for np > 1 (solvers.c:214-281).  The reference sorts on rank 0, broadcasts the whole COO
 * and lets every rank pick rows[l_idx..] out of it; here every rank reads the file itself, counts the rows,
 * asks mk_partition_rows for the cuts and keeps only its own entries. */

/* rowptr[r] = number of entries in rows < r (nrows + 1 values); 1 when an index is out of range */
int coo_row_pointers64(const SparseMatrixCOO *coo, long long *rowptr) {
  memset(rowptr, 0, ((size_t)coo->nrows + 1) * sizeof(long long));
  for (unsigned int k = 0; k < coo->nnz; k++) {
    const int r = coo->rows[k];
    if (r < 0 || (unsigned int)r >= coo->nrows) return 1;
    rowptr[r + 1]++;
  }
  for (unsigned int r = 0; r < coo->nrows; r++) rowptr[r + 1] += rowptr[r];
  return 0;
}

/* the entries of rows [row_lo, row_hi) in input order, rows renumbered from 0, global column ids kept */
int slice_coo_rows(const SparseMatrixCOO *coo, long long row_lo, long long row_hi, SparseMatrixCOO *out) {
  if (row_lo < 0 || row_hi < row_lo || row_hi > (long long)coo->nrows) return 1;
  size_t cnt = 0;
  for (unsigned int k = 0; k < coo->nnz; k++) cnt += (coo->rows[k] >= row_lo && coo->rows[k] < row_hi);
  out->nrows = (unsigned int)(row_hi - row_lo);
  out->ncolumns = coo->ncolumns;
  out->nnz = (unsigned int)cnt;
  out->rows = (int *)aalloc(cnt * sizeof(int));
  out->columns = (int *)aalloc(cnt * sizeof(int));
  out->values = (double *)aalloc(cnt * sizeof(double));
  if (!out->rows || !out->columns || !out->values) { free_coo(out); return 2; }
  size_t j = 0;
  for (unsigned int k = 0; k < coo->nnz; k++)
    if (coo->rows[k] >= row_lo && coo->rows[k] < row_hi) {
      out->rows[j] = (int)(coo->rows[k] - row_lo);
      out->columns[j] = coo->columns[k];
      out->values[j] = coo->values[k];
      j++;
    }
  return 0;
}
