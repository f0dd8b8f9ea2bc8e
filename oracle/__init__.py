"""ctypes front-end of the test oracle (oracle/liboracle.so) and of the reference's own
conversion code compiled into oracle/_ref/libref_sparse.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  The product (iterative_solvers_amd) never imports this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_sparse.so")

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """compile liboracle.so (and _ref when /root/reference is mounted)"""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_longlong), ("rel_res_norm", C.c_double),
                ("converged", C.c_int), ("breakdown", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_sort_coo_row.argtypes = [C.c_uint32, C.c_uint32, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p]
        L.orc_coo_to_csr.argtypes = [C.c_uint32, C.c_uint32, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p]
        L.orc_partition_rows.argtypes = [C.c_uint32, _i64p, C.c_int, _i64p]
        for f in (L.orc_csr_spmv, L.orc_csr_spmv_omp):
            f.argtypes = [C.c_uint32, _i32p, _i32p, _f64p, C.c_double, _f64p, C.c_double, _f64p]
        L.orc_csr_diag.argtypes = [C.c_uint32, _i32p, _i32p, _f64p, _f64p]
        L.orc_pcg.argtypes = [C.c_uint32, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_int, C.c_double,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SolveInfo), C.c_void_p]
        L.orc_jacobi.argtypes = [C.c_uint32, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_int,
                                 C.c_int, C.c_int, C.POINTER(SolveInfo)]
        L.orc_bicgstab.argtypes = [C.c_uint32, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_int, C.c_int,
                                   C.POINTER(SolveInfo), C.c_void_p]
        L.orc_cgnr.argtypes = L.orc_bicgstab.argtypes
        L.orc_gmres.argtypes = [C.c_uint32, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_int, C.c_int, C.c_int,
                                C.POINTER(SolveInfo), C.c_void_p]
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_rmat_coo.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_double,
                                   _i32p, _i32p, _f64p]
        L.orc_stencil_csr.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_stencil_csr.restype = C.c_int64
        L.orc_row_scale.argtypes = [C.c_uint64, C.c_int64, C.c_int64, _f64p]
        L.orc_csr_scale_sym.argtypes = [C.c_uint32, _i32p, _i32p, _f64p, _f64p]
        L.orc_banded_csr.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_banded_csr.restype = C.c_int64
        _lib = L
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def sort_coo_row(nrows, rows, cols, vals):
    rows, cols, vals = _c(rows, np.int32), _c(cols, np.int32), _c(vals, np.float64)
    nnz = rows.size
    sr, sc, sv = np.empty(nnz, np.int32), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
    lib().orc_sort_coo_row(nnz, nrows, rows, cols, vals, sr, sc, sv)
    return sr, sc, sv


def coo_to_csr(nrows, rows, cols, vals):
    rows, cols, vals = _c(rows, np.int32), _c(cols, np.int32), _c(vals, np.float64)
    nnz = rows.size
    rp, ci, v = np.empty(nrows + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
    lib().orc_coo_to_csr(nnz, nrows, rows, cols, vals, rp, ci, v)
    return rp, ci, v


def partition_rows(rowptr, nparts):
    rp = _c(rowptr, np.int64)
    out = np.empty(nparts + 1, np.int64)
    lib().orc_partition_rows(rp.size - 1, rp, nparts, out)
    return out


def csr_spmv(rowptr, col, val, x, alpha=1.0, beta=0.0, y=None, omp=False):
    rowptr, col, val, x = _c(rowptr, np.int32), _c(col, np.int32), _c(val, np.float64), _c(x, np.float64)
    n = rowptr.size - 1
    y = np.zeros(n) if y is None else _c(y, np.float64).copy()
    (lib().orc_csr_spmv_omp if omp else lib().orc_csr_spmv)(n, rowptr, col, val, alpha, x, beta, y)
    return y


def csr_diag(rowptr, col, val):
    rowptr, col, val = _c(rowptr, np.int32), _c(col, np.int32), _c(val, np.float64)
    d = np.empty(rowptr.size - 1)
    lib().orc_csr_diag(rowptr.size - 1, rowptr, col, val, d)
    return d


def pcg(rowptr, col, val, b, x0=None, precond=0, tol=1e-6, maxit=1000, two_norm=1, rel_change=1,
        omp=False, want_norms=False):
    rowptr, col, val, b = _c(rowptr, np.int32), _c(col, np.int32), _c(val, np.float64), _c(b, np.float64)
    n = rowptr.size - 1
    x = np.zeros(n) if x0 is None else _c(x0, np.float64).copy()
    info = SolveInfo()
    norms = np.full(maxit + 1, np.nan) if want_norms else None
    lib().orc_pcg(n, rowptr, col, val, b, x, precond, tol, maxit, two_norm, rel_change, int(omp),
                  C.byref(info), norms.ctypes.data if want_norms else None)
    out = dict(x=x, iterations=info.iterations, rel_res_norm=info.rel_res_norm,
               converged=info.converged, breakdown=info.breakdown)
    if want_norms:
        out["rel_norms"] = norms[: info.iterations + 1]
    return out


def jacobi(rowptr, col, val, b, x0=None, tol=1e-6, maxit=1000, check_every=1, omp=False):
    rowptr, col, val, b = _c(rowptr, np.int32), _c(col, np.int32), _c(val, np.float64), _c(b, np.float64)
    n = rowptr.size - 1
    x = np.zeros(n) if x0 is None else _c(x0, np.float64).copy()
    info = SolveInfo()
    lib().orc_jacobi(n, rowptr, col, val, b, x, tol, maxit, check_every, int(omp), C.byref(info))
    return dict(x=x, iterations=info.iterations, rel_res_norm=info.rel_res_norm,
                converged=info.converged)


def bicgstab(rowptr, col, val, b, x0=None, tol=1e-6, maxit=1000, omp=False, want_norms=False):
    rowptr, col, val, b = _c(rowptr, np.int32), _c(col, np.int32), _c(val, np.float64), _c(b, np.float64)
    n = rowptr.size - 1
    x = np.zeros(n) if x0 is None else _c(x0, np.float64).copy()
    info = SolveInfo()
    norms = np.full(maxit + 1, np.nan) if want_norms else None
    lib().orc_bicgstab(n, rowptr, col, val, b, x, tol, maxit, int(omp), C.byref(info),
                       norms.ctypes.data if want_norms else None)
    out = dict(x=x, iterations=info.iterations, rel_res_norm=info.rel_res_norm, converged=info.converged,
               breakdown=info.breakdown)
    if want_norms:
        out["norms"] = norms[: info.iterations + 1]
    return out


def cgnr(rowptr, col, val, b, x0=None, tol=1e-6, maxit=1000, omp=False, want_norms=False):
    rowptr, col, val, b = _c(rowptr, np.int32), _c(col, np.int32), _c(val, np.float64), _c(b, np.float64)
    n = rowptr.size - 1
    x = np.zeros(n) if x0 is None else _c(x0, np.float64).copy()
    info = SolveInfo()
    norms = np.full(maxit + 1, np.nan) if want_norms else None
    lib().orc_cgnr(n, rowptr, col, val, b, x, tol, maxit, int(omp), C.byref(info),
                   norms.ctypes.data if want_norms else None)
    out = dict(x=x, iterations=info.iterations, rel_res_norm=info.rel_res_norm, converged=info.converged)
    if want_norms:
        out["rel_norms"] = norms[: info.iterations + 1]
    return out


def gmres(rowptr, col, val, b, x0=None, tol=1e-6, maxit=1000, k_dim=30, omp=False, want_norms=False):
    rowptr, col, val, b = _c(rowptr, np.int32), _c(col, np.int32), _c(val, np.float64), _c(b, np.float64)
    n = rowptr.size - 1
    x = np.zeros(n) if x0 is None else _c(x0, np.float64).copy()
    info = SolveInfo()
    norms = np.full(maxit + 1, np.nan) if want_norms else None
    lib().orc_gmres(n, rowptr, col, val, b, x, tol, maxit, k_dim, int(omp), C.byref(info),
                    norms.ctypes.data if want_norms else None)
    out = dict(x=x, iterations=info.iterations, rel_res_norm=info.rel_res_norm, converged=info.converged)
    if want_norms:
        out["norms"] = norms[: info.iterations + 1]
    return out


def rmat_coo(scale, nedges, seed=42, a=0.57, b=0.19, c=0.19):
    r, cc, v = np.empty(nedges, np.int32), np.empty(nedges, np.int32), np.empty(nedges, np.float64)
    lib().orc_rmat_coo(scale, nedges, seed, a, b, c, r, cc, v)
    return r, cc, v


def row_scale(seed, n, row_begin=0):
    s = np.empty(n)
    lib().orc_row_scale(seed, row_begin, n, s)
    return s


def num_threads():
    return int(lib().orc_num_threads())


def stencil_csr(kind, nx, ny, nz=1, scale_seed=None):
    """scale_seed: D A D with the per-row scale of mk_matrix_generate_stencil_scaled (variable coefficients)"""
    nnz = lib().orc_stencil_csr(kind, nx, ny, nz, None, None, None)
    n = nx * ny * (1 if kind == 5 else nz)
    rp, ci, v = np.empty(n + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
    lib().orc_stencil_csr(kind, nx, ny, nz, rp.ctypes.data, ci.ctypes.data, v.ctypes.data)
    if scale_seed is not None:
        lib().orc_csr_scale_sym(n, rp, ci, v, row_scale(scale_seed, n))
    return rp, ci, v


def banded_csr(n, hb=3):
    nnz = lib().orc_banded_csr(n, hb, None, None, None)
    rp, ci, v = np.empty(n + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
    lib().orc_banded_csr(n, hb, rp.ctypes.data, ci.ctypes.data, v.ctypes.data)
    return rp, ci, v


# ---------------------------------------------------------------- the real reference ----

class _COO(C.Structure):  # sparse_matrix.h:35-44
    _fields_ = [("values", C.POINTER(C.c_double)), ("rows", C.POINTER(C.c_int)),
                ("columns", C.POINTER(C.c_int)), ("nrows", C.c_uint), ("ncolumns", C.c_uint),
                ("nnz", C.c_uint)]


class _CSR(C.Structure):  # sparse_matrix.h:24-33
    _fields_ = [("values", C.POINTER(C.c_double)), ("column_indices", C.POINTER(C.c_int)),
                ("row_pointers", C.POINTER(C.c_int)), ("nrows", C.c_uint), ("ncolumns", C.c_uint),
                ("nnz", C.c_uint)]


def have_ref():
    return os.path.exists(_REF)


_ref = None


def ref():
    """the reference's sparse_matrix.c/mmio.c compiled by oracle/Makefile (this container only)"""
    global _ref
    if _ref is None:
        build()
        _ref = C.CDLL(_REF)
    return _ref


def _coo_from_np(nrows, ncols, rows, cols, vals):
    rows, cols, vals = _c(rows, np.int32), _c(cols, np.int32), _c(vals, np.float64)
    m = _COO(vals.ctypes.data_as(C.POINTER(C.c_double)), rows.ctypes.data_as(C.POINTER(C.c_int)),
             cols.ctypes.data_as(C.POINTER(C.c_int)), nrows, ncols, rows.size)
    m._keep = (rows, cols, vals)
    return m


def ref_load_mtx(path):
    """fast_load_from_mtx_file (sparse_matrix.c:228-376); exits the process on bad input"""
    m = _COO()
    ref().fast_load_from_mtx_file(path.encode(), C.byref(m))
    nnz = m.nnz
    out = (m.nrows, m.ncolumns, np.ctypeslib.as_array(m.rows, (nnz,)).copy().astype(np.int32),
           np.ctypeslib.as_array(m.columns, (nnz,)).copy().astype(np.int32),
           np.ctypeslib.as_array(m.values, (nnz,)).copy()) if nnz else (
        m.nrows, m.ncolumns, np.empty(0, np.int32), np.empty(0, np.int32), np.empty(0))
    return out


def ref_sort_coo_row(nrows, ncols, rows, cols, vals):
    src = _coo_from_np(nrows, ncols, rows, cols, vals)
    dst = _COO()
    ref().sort_coo_row(C.byref(src), C.byref(dst))
    nnz = dst.nnz
    if nnz == 0:
        return np.empty(0, np.int32), np.empty(0, np.int32), np.empty(0)
    return (np.ctypeslib.as_array(dst.rows, (nnz,)).copy().astype(np.int32),
            np.ctypeslib.as_array(dst.columns, (nnz,)).copy().astype(np.int32),
            np.ctypeslib.as_array(dst.values, (nnz,)).copy())


def ref_coo_to_csr(nrows, ncols, rows, cols, vals):
    src = _coo_from_np(nrows, ncols, rows, cols, vals)
    dst = _CSR()
    ref().convert_coo_to_csr(C.byref(src), C.byref(dst), 0)
    nnz = dst.nnz
    rp = np.ctypeslib.as_array(dst.row_pointers, (nrows + 1,)).copy().astype(np.int32)
    if nnz == 0:
        return rp, np.empty(0, np.int32), np.empty(0)
    return (rp, np.ctypeslib.as_array(dst.column_indices, (nnz,)).copy().astype(np.int32),
            np.ctypeslib.as_array(dst.values, (nnz,)).copy())
