// Host-side AddressSanitizer pass over the library: the .hip sources are compiled with
// `-Xarch_host -fsanitize=address` into this one executable (tools/asan_check.sh) and the main entry points are
// driven once through the C ABI.  Checks host heap use only; device code is not instrumented.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../include/mk_krylov.h"

#define CK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, mk_last_error_string()); exit(1); } } while (0)

static void one_matrix(int nx, int ny, int nz) {
  const int64_t n = (int64_t)nx * ny * nz;
  mk_matrix *A = nullptr;
  CK(mk_matrix_generate_stencil(7, nx, ny, nz, 0, n, &A));
  mk_vector *b = nullptr, *x = nullptr;
  CK(mk_vector_create(n, &b)); CK(mk_vector_create(n, &x));
  CK(mk_vector_fill(b, 1.0));
  for (int kernel = 0; kernel <= 5; kernel++) {
    if (mk_matrix_set_spmv_kernel(A, kernel, 0) != 0) continue;
    CK(mk_vector_fill(x, 0.0));
    mk_pcg *S = nullptr;
    CK(mk_pcg_create(&S)); CK(mk_pcg_set_max_iter(S, 40)); CK(mk_pcg_set_tol(S, 1e-9)); CK(mk_pcg_set_logging(S, 1));
    CK(mk_pcg_setup(S, A, b, x)); CK(mk_pcg_solve(S, A, b, x));
    long long it = 0; CK(mk_pcg_get_num_iterations(S, &it));
    std::vector<double> hist(64);
    mk_pcg_get_residual_history(S, hist.data(), 64);
    printf("  kernel %d: pcg %lld iterations\n", kernel, it);
    CK(mk_pcg_destroy(S));
  }
  CK(mk_matrix_set_spmv_kernel(A, 0, 0));
  { mk_bicgstab *S; CK(mk_bicgstab_create(&S)); CK(mk_bicgstab_set_max_iter(S, 20)); CK(mk_bicgstab_set_logging(S, 1)); CK(mk_vector_fill(x, 0.0));
    CK(mk_bicgstab_setup(S, A, b, x)); CK(mk_bicgstab_solve(S, A, b, x)); CK(mk_bicgstab_destroy(S)); }
  { mk_cgnr *S; CK(mk_cgnr_create(&S)); CK(mk_cgnr_set_max_iter(S, 20)); CK(mk_cgnr_set_logging(S, 1)); CK(mk_vector_fill(x, 0.0));
    CK(mk_cgnr_setup(S, A, b, x)); CK(mk_cgnr_solve(S, A, b, x)); CK(mk_cgnr_destroy(S)); }
  { mk_gmres *S; CK(mk_gmres_create(&S)); CK(mk_gmres_set_max_iter(S, 45)); CK(mk_gmres_set_logging(S, 1)); CK(mk_vector_fill(x, 0.0));
    CK(mk_gmres_setup(S, A, b, x)); CK(mk_gmres_solve(S, A, b, x)); CK(mk_gmres_destroy(S)); }
  { mk_jacobi *S; CK(mk_jacobi_create(&S)); CK(mk_jacobi_set_max_iter(S, 30)); CK(mk_vector_fill(x, 0.0));
    CK(mk_jacobi_setup(S, A, b, x)); CK(mk_jacobi_solve(S, A, b, x)); CK(mk_jacobi_destroy(S)); }
  SparseMatrixCSR csr = {};
  CK(mk_matrix_get_csr(A, &csr));
  mk_matrix *B = nullptr;
  CK(mk_matrix_from_csr(&csr, &B));
  free(csr.row_pointers); free(csr.column_indices); free(csr.values);
  CK(mk_matrix_destroy(B));
  CK(mk_vector_destroy(b)); CK(mk_vector_destroy(x)); CK(mk_matrix_destroy(A));
}

static void rmat() {
  mk_coo *coo = nullptr;
  CK(mk_coo_generate_rmat(14, 1u << 18, 7, 0.57, 0.19, 0.19, &coo));
  mk_matrix *A = nullptr;
  CK(mk_matrix_from_device_coo(coo, &A));
  CK(mk_coo_destroy(coo));
  mk_matrix_info info; CK(mk_matrix_get_info(A, &info));
  mk_vector *x, *y; CK(mk_vector_create(info.ncols, &x)); CK(mk_vector_create(info.nrows, &y)); CK(mk_vector_fill(x, 1.0));
  for (int kernel = 0; kernel <= 3; kernel++) { if (mk_matrix_set_spmv_kernel(A, kernel, 0) == 0) CK(mk_spmv(A, 1.0, x, 0.0, y)); }
  double a, b2; CK(mk_spmv_dot(A, x, y, &a, &b2));
  CK(mk_vector_destroy(x)); CK(mk_vector_destroy(y)); CK(mk_matrix_destroy(A));
}

// variable coefficients (vsweep), 27-point rows (boundary-pairs-first table, half batch), banded (planes of chunks), static entry
static void more_kernels() {
  struct Case { int kind, nx, ny, nz; unsigned long long seed; int kernel; } cases[] = {
      {7, 96, 80, 40, 3, 6}, {27, 48, 40, 36, 0, 5}, {5, 300, 200, 1, 0, 5}, {7, 64, 64, 64, 9, 6}};
  for (const Case &c : cases) {
    const int64_t n = (int64_t)c.nx * c.ny * c.nz;
    mk_matrix *A = nullptr;
    if (c.seed) CK(mk_matrix_generate_stencil_scaled(c.kind, c.nx, c.ny, c.nz, c.seed, 0, n, &A));
    else CK(mk_matrix_generate_stencil(c.kind, c.nx, c.ny, c.nz, 0, n, &A));
    const bool forced = mk_matrix_set_spmv_kernel(A, c.kernel, 0) == 0;
    mk_vector *x, *y; CK(mk_vector_create(n, &x)); CK(mk_vector_create(n, &y)); CK(mk_vector_fill(x, 1.0));
    CK(mk_spmv(A, 2.0, x, 0.0, y)); CK(mk_spmv(A, 1.0, x, -1.0, y));
    double a, b2; CK(mk_spmv_dot(A, x, y, &a, &b2)); CK(mk_spmv_dot_static(A, x, y, &a, &b2));
    mk_pcg *S; CK(mk_pcg_create(&S)); CK(mk_pcg_set_max_iter(S, 25)); CK(mk_pcg_set_tol(S, 0.0)); CK(mk_pcg_set_precond(S, 1));
    CK(mk_vector_fill(y, 0.0)); CK(mk_pcg_setup(S, A, x, y)); CK(mk_pcg_solve(S, A, x, y)); CK(mk_pcg_destroy(S));
    mk_jacobi *J; CK(mk_jacobi_create(&J)); CK(mk_jacobi_set_max_iter(J, 12)); CK(mk_vector_fill(y, 0.0));
    CK(mk_jacobi_setup(J, A, x, y)); CK(mk_jacobi_solve(J, A, x, y)); CK(mk_jacobi_destroy(J));
    printf("  kind %d %dx%dx%d seed %llu kernel %d%s\n", c.kind, c.nx, c.ny, c.nz, c.seed, c.kernel, forced ? "" : " (not eligible, default kernel)");
    CK(mk_vector_destroy(x)); CK(mk_vector_destroy(y)); CK(mk_matrix_destroy(A));
  }
  mk_matrix *B = nullptr;
  CK(mk_matrix_generate_banded(300003, 3, 0, 300003, &B));
  mk_vector *x, *y; CK(mk_vector_create(300003, &x)); CK(mk_vector_create(300003, &y)); CK(mk_vector_fill(x, 1.0));
  CK(mk_spmv(B, 1.0, x, 0.0, y));
  mk_jacobi *J; CK(mk_jacobi_create(&J)); CK(mk_jacobi_set_max_iter(J, 12)); CK(mk_vector_fill(y, 0.0));
  CK(mk_jacobi_setup(J, B, x, y)); CK(mk_jacobi_solve(J, B, x, y)); CK(mk_jacobi_destroy(J));
  CK(mk_vector_destroy(x)); CK(mk_vector_destroy(y)); CK(mk_matrix_destroy(B));
  long long in_use = 0, idle = 0;
  CK(mk_memory_info(&in_use, &idle)); CK(mk_memory_trim()); CK(mk_memory_info(&in_use, &idle));
  if (idle != 0) { fprintf(stderr, "memory trim left %lld idle bytes\n", idle); exit(1); }
}

// communicators destroyed before the matrices created on them (a host language's finaliser order)
// three unknowns per node: a grid of nodes (index patterns with a shape table), random neighbour nodes (shared column lists),
// and short rows with a few long ones (split kernel, pieces summed by a wave)
static void product_all_modes(int n, std::vector<int> &rp, std::vector<int> &ci, std::vector<double> &v, const char *what) {
  SparseMatrixCSR csr = {};
  csr.values = v.data(); csr.row_pointers = rp.data(); csr.column_indices = ci.data();
  csr.nrows = (unsigned)n; csr.ncolumns = (unsigned)n; csr.nnz = (unsigned)ci.size();
  mk_matrix *A = nullptr;
  CK(mk_matrix_from_csr(&csr, &A));
  mk_matrix_info info; CK(mk_matrix_get_info(A, &info));
  printf("  %s: kernel %d, %lld index patterns, %lld shared column entries, lanes %d chunk %d\n", what, info.kernel,
         (long long)info.index_patterns, (long long)info.shared_columns, info.stream_lanes, info.stream_chunk);
  mk_vector *x, *y, *b; CK(mk_vector_create(n, &x)); CK(mk_vector_create(n, &y)); CK(mk_vector_create(n, &b));
  CK(mk_vector_fill(x, 1.0)); CK(mk_vector_fill(b, 1.0));
  CK(mk_spmv(A, 1.0, x, 0.0, y)); CK(mk_spmv(A, 0.5, x, 2.0, y));
  double a, b2; CK(mk_spmv_dot(A, x, y, &a, &b2));
  { mk_jacobi *S; CK(mk_jacobi_create(&S)); CK(mk_jacobi_set_max_iter(S, 5)); CK(mk_vector_fill(x, 0.0));
    CK(mk_jacobi_setup(S, A, b, x)); CK(mk_jacobi_solve(S, A, b, x)); CK(mk_jacobi_destroy(S)); }
  CK(mk_vector_destroy(x)); CK(mk_vector_destroy(y)); CK(mk_vector_destroy(b)); CK(mk_matrix_destroy(A));
}

static void fem_kernels() {
  unsigned long long st = 12345;
  auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (unsigned)(st >> 33); };
  {  // 27 node neighbours x 3x3 blocks on a 12 x 9 x 8 grid of nodes
    const int nx = 12, ny = 9, nz = 8, nn = nx * ny * nz;
    std::vector<int> rp(1, 0), ci; std::vector<double> v;
    for (int node = 0; node < nn; node++) {
      const int ix = node % nx, iy = (node / nx) % ny, iz = node / (nx * ny);
      for (int d = 0; d < 3; d++) {
        for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
          if (ix + dx < 0 || ix + dx >= nx || iy + dy < 0 || iy + dy >= ny || iz + dz < 0 || iz + dz >= nz) continue;
          const int nb = node + dx + dy * nx + dz * nx * ny;
          for (int e = 0; e < 3; e++) { ci.push_back(3 * nb + e); v.push_back(nb == node && e == d ? 200.0 : -1.0 - 0.001 * (rnd() % 1000)); }
        }
        rp.push_back((int)ci.size());
      }
    }
    product_all_modes(3 * nn, rp, ci, v, "node grid");
  }
  {  // random neighbour nodes: the three rows of a node share one column list, no two nodes a pattern
    const int nn = 3000;
    std::vector<int> rp(1, 0), ci; std::vector<double> v;
    for (int node = 0; node < nn; node++) {
      std::vector<int> nb(1, node);
      const int k = 8 + rnd() % 20;
      for (int t = 0; t < k; t++) { int j = node - 40 + (int)(rnd() % 81); if (j >= 0 && j < nn) nb.push_back(j); }
      std::sort(nb.begin(), nb.end()); nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
      for (int d = 0; d < 3; d++) {
        for (int j : nb) for (int e = 0; e < 3; e++) { ci.push_back(3 * j + e); v.push_back(j == node && e == d ? 300.0 : -1.0 - 0.001 * (rnd() % 1000)); }
        rp.push_back((int)ci.size());
      }
    }
    product_all_modes(3 * nn, rp, ci, v, "node blocks, random neighbours");
  }
  {  // short rows, every 3000th row 500 entries long
    const int n = 20000;
    std::vector<int> rp(1, 0), ci; std::vector<double> v;
    for (int r = 0; r < n; r++) {
      const int len = (r % 3000 == 777) ? 500 : 3 + (int)(rnd() % 6);
      std::vector<int> c(1, r);
      for (int t = 1; t < len; t++) c.push_back((int)(rnd() % n));
      std::sort(c.begin(), c.end());
      for (int j : c) { ci.push_back(j); v.push_back(j == r ? 1000.0 : -1.0); }
      rp.push_back((int)ci.size());
    }
    product_all_modes(n, rp, ci, v, "short rows with long ones");
  }
}

static void comm_first() {
  const int R = 3, nx = 16, ny = 12, nz = 18;
  const int64_t n = (int64_t)nx * ny * nz;
  std::vector<int64_t> rs(R + 1);
  for (int r = 0; r <= R; r++) rs[r] = (int64_t)nx * ny * ((int64_t)nz * r / R);
  std::vector<mk_comm *> comms(R);
  CK(mk_comm_create_local(R, comms.data()));
  std::vector<mk_dist_matrix *> D(R);
  for (int r = 0; r < R; r++) {
    mk_matrix *blk = nullptr;
    CK(mk_matrix_generate_stencil(7, nx, ny, nz, rs[r], rs[r + 1], &blk));
    CK(mk_dist_matrix_create(blk, n, rs.data(), comms[r], &D[r]));
  }
  for (int r = 0; r < R; r++) CK(mk_comm_destroy(comms[r]));
  for (int r = R - 1; r >= 0; r--) CK(mk_dist_matrix_destroy(D[r]));
}

static void world(int nx, int ny, int nz, int R, bool rccl, int precond) {
  const int64_t n = (int64_t)nx * ny * nz;
  std::vector<int64_t> rs(R + 1);
  for (int r = 0; r <= R; r++) rs[r] = (int64_t)nx * ny * ((int64_t)nz * r / R);
  std::vector<mk_comm *> comms(R);
  if (rccl) {
    char id[256]; CK(mk_comm_get_unique_id(id)); CK(mk_comm_create(id, 1, 0, &comms[0])); CK(mk_comm_self_test(comms[0]));
  } else CK(mk_comm_create_local(R, comms.data()));
  std::vector<mk_dist_matrix *> D(R);
  std::vector<mk_vector *> b(R), x(R);
  std::vector<mk_pcg *> S(R);
  std::vector<mk_jacobi *> J(R);
  for (int r = 0; r < R; r++) {
    mk_matrix *blk = nullptr;
    CK(mk_matrix_generate_stencil(7, nx, ny, nz, rs[r], rs[r + 1], &blk));
    CK(mk_dist_matrix_create(blk, n, rs.data(), comms[r], &D[r]));
  }
  for (int r = 0; r < R; r++) {
    CK(mk_dist_vector_create(D[r], &b[r])); CK(mk_dist_vector_create(D[r], &x[r]));
    CK(mk_vector_fill(b[r], 1.0)); CK(mk_vector_fill(x[r], 0.0));
    CK(mk_pcg_create(&S[r])); CK(mk_pcg_set_max_iter(S[r], 60)); CK(mk_pcg_set_tol(S[r], 1e-9)); CK(mk_pcg_set_logging(S[r], 1));
    CK(mk_pcg_set_precond(S[r], precond));
    CK(mk_jacobi_create(&J[r])); CK(mk_jacobi_set_max_iter(J[r], 25));
  }
  CK(mk_dist_spmv(R, D.data(), b.data(), x.data()));
  for (int r = 0; r < R; r++) CK(mk_vector_fill(x[r], 0.0));
  CK(mk_dist_pcg_solve(R, S.data(), D.data(), b.data(), x.data()));
  long long it = 0; CK(mk_pcg_get_num_iterations(S[0], &it));
  printf("  world of %d (%s, precond %d): pcg %lld iterations\n", R, rccl ? "rccl" : "local", precond, it);
  for (int r = 0; r < R; r++) CK(mk_vector_fill(x[r], 0.0));
  CK(mk_dist_jacobi_solve(R, J.data(), D.data(), b.data(), x.data()));
  for (int r = 0; r < R; r++) {
    CK(mk_pcg_destroy(S[r])); CK(mk_jacobi_destroy(J[r])); CK(mk_vector_destroy(b[r])); CK(mk_vector_destroy(x[r]));
    CK(mk_dist_matrix_destroy(D[r]));
  }
  for (int r = 0; r < R; r++) CK(mk_comm_destroy(comms[r]));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  CK(mk_init(0));
  printf("single matrix\n");
  one_matrix(64, 64, 64);
  one_matrix(19, 11, 13);
  one_matrix(128, 128, 96);
  printf("rmat\n");
  rmat();
  printf("more kernels\n");
  more_kernels();
  printf("finite-element shapes\n");
  fem_kernels();
  printf("row blocks\n");
  comm_first();
  world(64, 64, 64, 4, false, 0);
  world(128, 128, 128, 8, false, 0);
  world(128, 128, 128, 8, false, 1);
  world(24, 20, 9, 3, false, 0);
  world(64, 64, 32, 1, true, 0);
  CK(mk_finalize());
  printf("asan_check: done\n");
  return 0;
}
