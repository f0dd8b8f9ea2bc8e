// generate.hip -- synthetic operators built directly in device memory (bench.py and tests).
// No reference counterpart: the reference only reads SuiteSparse .mtx files
// (scripts/bench_all_gpu:2).  Definitions match oracle/oracle.c bit for bit.
#include "mk_internal.h"

namespace mk {

struct StencilGeom {
  int kind, nx, ny, nz;
};

__device__ __forceinline__ void row_to_xyz(const StencilGeom g, int64_t row, int &ix, int &iy, int &iz) {
  ix = (int)(row % g.nx);
  iy = (int)((row / g.nx) % g.ny);
  iz = (int)(row / ((int64_t)g.nx * g.ny));
}

__device__ __forceinline__ int stencil_row_len(const StencilGeom g, int ix, int iy, int iz) {
  const int cx = 1 + (ix > 0) + (ix < g.nx - 1);
  const int cy = 1 + (iy > 0) + (iy < g.ny - 1);
  const int cz = 1 + (iz > 0) + (iz < g.nz - 1);
  if (g.kind == 27) return cx * cy * cz;
  if (g.kind == 7) return cx + cy + cz - 2;
  return cx + cy - 1;
}

__global__ void stencil_count_kernel(StencilGeom g, int64_t row_begin, int64_t nlocal, uint32_t *__restrict__ counts) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nlocal; r += (int64_t)gridDim.x * blockDim.x) {
    int ix, iy, iz;
    row_to_xyz(g, row_begin + r, ix, iy, iz);
    counts[r] = (uint32_t)stencil_row_len(g, ix, iy, iz);
  }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x);

// Row scale of the variable-coefficient operators: s_i = (1024 + 12 hash bits) / 1024 in [1, 5), a dyadic rational
// of 13 bits, so that (a_ij * s_i) * s_j is exact in fp64 (no rounding, no dependence on contraction) and symmetric.
// D A D with D = diag(s) > 0 keeps A symmetric positive definite.  oracle/oracle.c orc_row_scale is the same.
__device__ __forceinline__ double row_scale(uint64_t seed, int64_t row) {
  const uint64_t h = splitmix64(seed ^ ((uint64_t)row * 0xD1B54A32D192ED03ULL));
  return (double)(1024 + (int)(h >> 52)) * (1.0 / 1024.0);
}

__global__ void scale_sym_kernel(int64_t nlocal, int64_t row_begin, uint64_t seed, const int *__restrict__ rowptr,
                                 const int *__restrict__ col, double *__restrict__ val) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nlocal; r += (int64_t)gridDim.x * blockDim.x) {
    const double si = row_scale(seed, row_begin + r);
    for (int k = rowptr[r]; k < rowptr[r + 1]; k++) val[k] = (val[k] * si) * row_scale(seed, col[k]);
  }
}

__global__ void stencil_fill_kernel(StencilGeom g, int64_t row_begin, int64_t nlocal, const int *__restrict__ rowptr,
                                    int *__restrict__ col, double *__restrict__ val) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nlocal; r += (int64_t)gridDim.x * blockDim.x) {
    int ix, iy, iz;
    row_to_xyz(g, row_begin + r, ix, iy, iz);
    int k = rowptr[r];
    if (g.kind == 27) {
      for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
          for (int dx = -1; dx <= 1; dx++) {
            const int x = ix + dx, y = iy + dy, z = iz + dz;
            if (x < 0 || x >= g.nx || y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
            col[k] = (int)(((int64_t)z * g.ny + y) * g.nx + x);
            val[k] = (dx == 0 && dy == 0 && dz == 0) ? 26.0 : -1.0;
            k++;
          }
    } else {
      const double diag = (g.kind == 5) ? 4.0 : 6.0;
      // ascending columns: z-1, y-1, x-1, self, x+1, y+1, z+1
      if (g.kind == 7 && iz > 0) { col[k] = (int)(((int64_t)(iz - 1) * g.ny + iy) * g.nx + ix); val[k++] = -1.0; }
      if (iy > 0) { col[k] = (int)(((int64_t)iz * g.ny + iy - 1) * g.nx + ix); val[k++] = -1.0; }
      if (ix > 0) { col[k] = (int)(((int64_t)iz * g.ny + iy) * g.nx + ix - 1); val[k++] = -1.0; }
      col[k] = (int)(((int64_t)iz * g.ny + iy) * g.nx + ix); val[k++] = diag;
      if (ix < g.nx - 1) { col[k] = (int)(((int64_t)iz * g.ny + iy) * g.nx + ix + 1); val[k++] = -1.0; }
      if (iy < g.ny - 1) { col[k] = (int)(((int64_t)iz * g.ny + iy + 1) * g.nx + ix); val[k++] = -1.0; }
      if (g.kind == 7 && iz < g.nz - 1) { col[k] = (int)(((int64_t)(iz + 1) * g.ny + iy) * g.nx + ix); val[k++] = -1.0; }
    }
  }
}

__global__ void banded_count_kernel(int64_t n, int hb, int64_t row_begin, int64_t nlocal, uint32_t *__restrict__ counts) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nlocal; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = row_begin + r;
    const int64_t lo = max((int64_t)0, row - hb), hi = min(n - 1, row + hb);
    counts[r] = (uint32_t)(hi - lo + 1);
  }
}

__global__ void banded_fill_kernel(int64_t n, int hb, int64_t row_begin, int64_t nlocal, const int *__restrict__ rowptr,
                                   int *__restrict__ col, double *__restrict__ val) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nlocal; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = row_begin + r;
    const int64_t lo = max((int64_t)0, row - hb), hi = min(n - 1, row + hb);
    int k = rowptr[r];
    for (int64_t cc = lo; cc <= hi; cc++, k++) {
      col[k] = (int)cc;
      val[k] = (cc == row) ? (double)(2 * hb + 1) : -1.0;
    }
  }
}

__global__ void set_last_rowptr_kernel(int *rowptr, int64_t nlocal, const uint32_t *total) { rowptr[nlocal] = (int)*total; }

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

__global__ void rmat_kernel(int scale, uint64_t nedges, uint64_t seed, uint32_t ta, uint32_t tb, uint32_t tc,
                            int *__restrict__ rows, int *__restrict__ cols, double *__restrict__ vals) {
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nedges; e += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t base = splitmix64(seed ^ (e * 0xD1B54A32D192ED03ULL));
    uint32_t r = 0, cc = 0;
    uint64_t w = 0;
    for (int l = 0; l < scale; l++) {
      if ((l & 3) == 0) w = splitmix64(base + (uint64_t)(l >> 2));
      const uint32_t u = (uint32_t)(w >> (16 * (l & 3))) & 0xFFFFu;
      const uint32_t rb = (u >= tb), cb = ((u >= ta && u < tb) || (u >= tc));
      r = (r << 1) | rb;
      cc = (cc << 1) | cb;
    }
    const uint64_t wv = splitmix64(base + 15);
    rows[e] = (int)r;
    cols[e] = (int)cc;
    vals[e] = (double)(wv >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
  }
}

static int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192)); }

// counts -> rowptr (exclusive scan), then allocate col/val
static int finish_structure(mk_matrix *M, uint32_t *counts, int64_t nlocal) {
  Context &c = ctx();
  uint32_t *d_total = nullptr, h_total = 0;
  MK_TRY(dev_alloc(&d_total, 1));
  MK_TRY(dev_alloc(&M->rowptr, (size_t)nlocal + 1));
  MK_TRY(exclusive_scan_u32(counts, reinterpret_cast<uint32_t *>(M->rowptr), (size_t)nlocal, d_total));
  hipLaunchKernelGGL(set_last_rowptr_kernel, dim3(1), dim3(1), 0, c.stream, M->rowptr, nlocal, d_total);
  MK_HIP(hipMemcpyAsync(&h_total, d_total, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
  MK_HIP(hipStreamSynchronize(c.stream));
  dev_free(d_total);
  if (h_total > 0x7fffffffu) return fail(MK_ERR_UNSUPPORTED, "generated block has %u entries: exceeds int32 row pointers", h_total);
  M->nnz = h_total;
  MK_TRY(dev_alloc(&M->col, (size_t)M->nnz));
  MK_TRY(dev_alloc(&M->val, (size_t)M->nnz));
  return MK_OK;
}

}  // namespace mk

using namespace mk;

extern "C" {

int mk_matrix_generate_stencil(int kind, int nx, int ny, int nz, int64_t row_begin, int64_t row_end, mk_matrix **A) {
  MK_REQUIRE_INIT();
  if (!A) return fail(MK_ERR_INVALID, "mk_matrix_generate_stencil: null output");
  if (kind != 5 && kind != 7 && kind != 27) return fail(MK_ERR_INVALID, "mk_matrix_generate_stencil: kind must be 5, 7 or 27");
  if (kind == 5) nz = 1;
  if (nx < 1 || ny < 1 || nz < 1) return fail(MK_ERR_INVALID, "mk_matrix_generate_stencil: bad grid");
  const int64_t n = (int64_t)nx * ny * nz;
  if (n > 0x7fffffffLL) return fail(MK_ERR_UNSUPPORTED, "mk_matrix_generate_stencil: more than 2^31-1 columns");
  if (row_begin < 0 || row_end > n || row_begin > row_end) return fail(MK_ERR_INVALID, "mk_matrix_generate_stencil: bad row range");
  Context &c = ctx();
  const int64_t nlocal = row_end - row_begin;
  StencilGeom g{kind, nx, ny, nz};
  mk_matrix *M = new mk_matrix();
  M->nrows = nlocal; M->ncols = n;
  uint32_t *counts = nullptr;
  int rc = dev_alloc(&counts, (size_t)nlocal);
  if (rc == MK_OK) {
    hipLaunchKernelGGL(stencil_count_kernel, dim3(grid_for(nlocal)), dim3(256), 0, c.stream, g, row_begin, nlocal, counts);
    rc = finish_structure(M, counts, nlocal);
  }
  if (rc == MK_OK) {
    hipLaunchKernelGGL(stencil_fill_kernel, dim3(grid_for(nlocal)), dim3(256), 0, c.stream, g, row_begin, nlocal, M->rowptr, M->col, M->val);
    hipError_t e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) rc = fail(MK_ERR_HIP, "stencil fill: %s", hipGetErrorString(e));
  }
  dev_free(counts);
  if (rc != MK_OK) { mk_matrix_destroy(M); return rc; }
  *A = M;
  return MK_OK;
}

int mk_matrix_generate_stencil_scaled(int kind, int nx, int ny, int nz, uint64_t seed, int64_t row_begin, int64_t row_end, mk_matrix **A) {
  mk_matrix *M = nullptr;
  MK_TRY(mk_matrix_generate_stencil(kind, nx, ny, nz, row_begin, row_end, &M));
  if (M->nrows > 0) {
    hipLaunchKernelGGL(scale_sym_kernel, dim3(grid_for(M->nrows)), dim3(256), 0, ctx().stream, M->nrows, row_begin, seed, M->rowptr, M->col, M->val);
    hipError_t e = hipStreamSynchronize(ctx().stream);
    if (e != hipSuccess) { mk_matrix_destroy(M); return fail(MK_ERR_HIP, "stencil scaling: %s", hipGetErrorString(e)); }
  }
  *A = M;
  return MK_OK;
}

int mk_matrix_generate_banded(int64_t n, int hb, int64_t row_begin, int64_t row_end, mk_matrix **A) {
  MK_REQUIRE_INIT();
  if (!A || n < 1 || hb < 0 || n > 0x7fffffffLL) return fail(MK_ERR_INVALID, "mk_matrix_generate_banded: bad argument");
  if (row_begin < 0 || row_end > n || row_begin > row_end) return fail(MK_ERR_INVALID, "mk_matrix_generate_banded: bad row range");
  Context &c = ctx();
  const int64_t nlocal = row_end - row_begin;
  mk_matrix *M = new mk_matrix();
  M->nrows = nlocal; M->ncols = n;
  uint32_t *counts = nullptr;
  int rc = dev_alloc(&counts, (size_t)nlocal);
  if (rc == MK_OK) {
    hipLaunchKernelGGL(banded_count_kernel, dim3(grid_for(nlocal)), dim3(256), 0, c.stream, n, hb, row_begin, nlocal, counts);
    rc = finish_structure(M, counts, nlocal);
  }
  if (rc == MK_OK) {
    hipLaunchKernelGGL(banded_fill_kernel, dim3(grid_for(nlocal)), dim3(256), 0, c.stream, n, hb, row_begin, nlocal, M->rowptr, M->col, M->val);
    hipError_t e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) rc = fail(MK_ERR_HIP, "banded fill: %s", hipGetErrorString(e));
  }
  dev_free(counts);
  if (rc != MK_OK) { mk_matrix_destroy(M); return rc; }
  *A = M;
  return MK_OK;
}

int mk_coo_generate_rmat(int scale, uint64_t nedges, uint64_t seed, double a, double b, double cc, mk_coo **coo) {
  MK_REQUIRE_INIT();
  if (!coo || scale < 1 || scale > 30) return fail(MK_ERR_INVALID, "mk_coo_generate_rmat: scale must be 1..30");
  if (nedges > 0x7fffffffULL) return fail(MK_ERR_UNSUPPORTED, "mk_coo_generate_rmat: more than 2^31-1 edges");
  if (!(a >= 0 && b >= 0 && cc >= 0 && a + b + cc <= 1.0)) return fail(MK_ERR_INVALID, "mk_coo_generate_rmat: bad probabilities");
  Context &c = ctx();
  mk_coo *d = new mk_coo();
  d->nrows = d->ncols = (int64_t)1 << scale;
  d->nnz = (int64_t)nedges;
  int rc = dev_alloc(&d->rows, (size_t)nedges);
  if (rc == MK_OK) rc = dev_alloc(&d->cols, (size_t)nedges);
  if (rc == MK_OK) rc = dev_alloc(&d->vals, (size_t)nedges);
  if (rc != MK_OK) { mk_coo_destroy(d); return rc; }
  const uint32_t ta = (uint32_t)(a * 65536.0);
  const uint32_t tb = ta + (uint32_t)(b * 65536.0);
  const uint32_t tc = tb + (uint32_t)(cc * 65536.0);
  if (nedges) {
    hipLaunchKernelGGL(rmat_kernel, dim3(grid_for((int64_t)nedges)), dim3(256), 0, c.stream, scale, nedges, seed, ta, tb, tc, d->rows, d->cols, d->vals);
    hipError_t e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) { mk_coo_destroy(d); return fail(MK_ERR_HIP, "rmat: %s", hipGetErrorString(e)); }
  }
  *coo = d;
  return MK_OK;
}

}  // extern "C"
