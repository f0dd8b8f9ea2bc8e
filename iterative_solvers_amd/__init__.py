"""iterative_solvers_amd -- thin ctypes binding of libmi355_krylov.so (include/mk_krylov.h).

The product is the HIP library; this module only marshals numpy arrays into the C ABI so
that tests and bench.py can drive it.  There is no CPU fallback: if the shared library is
missing, or no gfx950 GPU is visible when a compute call is made, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MK_LIB_PATH") or os.path.join(_HERE, "lib", "libmi355_krylov.so")
CSRC = os.path.join(_HERE, "csrc")


class MkError(RuntimeError):
    pass


class COO(C.Structure):  # sparse_matrix.h:35-44
    _fields_ = [("values", C.POINTER(C.c_double)), ("rows", C.POINTER(C.c_int)),
                ("columns", C.POINTER(C.c_int)), ("nrows", C.c_uint), ("ncolumns", C.c_uint),
                ("nnz", C.c_uint)]


class CSR(C.Structure):  # sparse_matrix.h:24-33
    _fields_ = [("values", C.POINTER(C.c_double)), ("column_indices", C.POINTER(C.c_int)),
                ("row_pointers", C.POINTER(C.c_int)), ("nrows", C.c_uint), ("ncolumns", C.c_uint),
                ("nnz", C.c_uint)]


class MatrixInfo(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64),
                ("max_row_len", C.c_int32), ("min_row_len", C.c_int32), ("mean_row_len", C.c_double),
                ("hist", C.c_int64 * 16), ("kernel", C.c_int32), ("lanes_per_row", C.c_int32),
                ("long_rows", C.c_int64), ("index_patterns", C.c_int64), ("value_patterns", C.c_int64),
                ("stream_lanes", C.c_int32), ("stream_chunk", C.c_int32), ("shared_columns", C.c_int64),
                ("row_groups", C.c_int64), ("group_lanes", C.c_int32), ("band_window", C.c_int32), ("band_strips", C.c_int32), ("reserved1", C.c_int32)]


SPMV_AUTO, SPMV_VECTOR, SPMV_STREAM, SPMV_SPLIT, SPMV_PATTERN, SPMV_SWEEP, SPMV_VSWEEP = 0, 1, 2, 3, 4, 5, 6
PRECOND_NONE, PRECOND_JACOBI = 0, 1

# every symbol include/mk_krylov.h declares (tests check the library exports all of them)
EXPORTS = [
    "mk_init", "mk_finalize", "mk_last_error_string", "mk_get_stream", "mk_get_device_pci_bus_id", "mk_synchronize",
    "mk_memory_info", "mk_memory_trim", "mk_memory_calls",
    "mk_timer_start", "mk_timer_stop",
    "mk_sort_coo_row", "mk_coo_to_csr",
    "mk_matrix_from_coo", "mk_matrix_from_csr", "mk_matrix_from_device_coo", "mk_matrix_destroy",
    "mk_matrix_get_csr", "mk_matrix_get_info", "mk_matrix_set_spmv_kernel",
    "mk_matrix_generate_stencil", "mk_matrix_generate_stencil_scaled", "mk_matrix_generate_banded", "mk_coo_generate_rmat",
    "mk_coo_from_host", "mk_coo_download", "mk_coo_destroy",
    "mk_vector_create", "mk_vector_set", "mk_vector_fill", "mk_vector_get", "mk_vector_destroy",
    "mk_vector_size", "mk_spmv", "mk_spmv_dot", "mk_spmv_dot_static",
    "mk_pcg_create", "mk_pcg_set_max_iter", "mk_pcg_set_tol", "mk_pcg_set_two_norm",
    "mk_pcg_set_rel_change", "mk_pcg_set_print_level", "mk_pcg_set_logging", "mk_pcg_set_precond",
    "mk_pcg_set_check_every", "mk_pcg_set_single_reduction", "mk_pcg_get_single_reduction", "mk_pcg_set_kernel_timing", "mk_pcg_get_kernel_timing", "mk_pcg_setup", "mk_pcg_solve", "mk_pcg_get_num_iterations",
    "mk_pcg_get_final_relative_residual_norm", "mk_pcg_get_converged",
    "mk_pcg_get_residual_history", "mk_pcg_destroy",
    "mk_bicgstab_create", "mk_bicgstab_set_max_iter", "mk_bicgstab_set_tol", "mk_bicgstab_set_print_level",
    "mk_bicgstab_set_logging", "mk_bicgstab_set_check_every", "mk_bicgstab_setup", "mk_bicgstab_solve",
    "mk_bicgstab_get_num_iterations", "mk_bicgstab_get_final_relative_residual_norm", "mk_bicgstab_get_converged",
    "mk_bicgstab_get_residual_history", "mk_bicgstab_destroy",
    "mk_cgnr_create", "mk_cgnr_set_max_iter", "mk_cgnr_set_tol", "mk_cgnr_set_logging", "mk_cgnr_set_check_every",
    "mk_cgnr_setup", "mk_cgnr_solve", "mk_cgnr_get_num_iterations", "mk_cgnr_get_final_relative_residual_norm",
    "mk_cgnr_get_converged", "mk_cgnr_get_residual_history", "mk_cgnr_destroy",
    "mk_gmres_create", "mk_gmres_set_max_iter", "mk_gmres_set_k_dim", "mk_gmres_set_tol", "mk_gmres_set_print_level",
    "mk_gmres_set_logging", "mk_gmres_setup", "mk_gmres_solve", "mk_gmres_get_num_iterations",
    "mk_gmres_get_final_relative_residual_norm", "mk_gmres_get_converged", "mk_gmres_get_residual_history",
    "mk_gmres_destroy",
    "mk_jacobi_create", "mk_jacobi_set_max_iter", "mk_jacobi_set_tol", "mk_jacobi_set_check_every",
    "mk_jacobi_setup", "mk_jacobi_solve", "mk_jacobi_get_num_iterations",
    "mk_jacobi_get_final_relative_residual_norm", "mk_jacobi_get_converged", "mk_jacobi_destroy",
    "mk_partition_rows", "mk_halo_check_row_start", "mk_halo_ghost_columns", "mk_halo_plan_recv", "mk_halo_plan_send",
    "mk_comm_get_unique_id", "mk_comm_create", "mk_comm_create_local", "mk_comm_self_test", "mk_comm_all_reduce_host", "mk_comm_destroy",
    "mk_dist_matrix_create", "mk_dist_matrix_destroy", "mk_dist_matrix_get_info", "mk_dist_matrix_get_diag_info", "mk_dist_matrix_set_spmv_kernel",
    "mk_dist_vector_create", "mk_dist_spmv", "mk_dist_pcg_solve", "mk_dist_jacobi_solve", "mk_dist_bicgstab_solve", "mk_dist_gmres_solve", "mk_dist_cgnr_solve",
]

_lib = None
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def load():
    """dlopen the HIP library (no GPU needed for this step)"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MkError(f"{LIB_PATH} is missing: build it with `make -C {CSRC}` "
                      "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    L.mk_last_error_string.restype = C.c_char_p
    L.mk_get_stream.restype = C.c_void_p
    L.mk_vector_size.restype = C.c_int64
    P = C.c_void_p
    sig = {
        "mk_init": [C.c_int], "mk_timer_stop": [C.POINTER(C.c_float)],
        "mk_sort_coo_row": [C.POINTER(COO), C.POINTER(COO)],
        "mk_coo_to_csr": [C.POINTER(COO), C.POINTER(CSR)],
        "mk_matrix_from_coo": [C.POINTER(COO), C.POINTER(P)],
        "mk_matrix_from_csr": [C.POINTER(CSR), C.POINTER(P)],
        "mk_matrix_from_device_coo": [P, C.POINTER(P)],
        "mk_matrix_destroy": [P], "mk_matrix_get_csr": [P, C.POINTER(CSR)],
        "mk_matrix_get_info": [P, C.POINTER(MatrixInfo)],
        "mk_dist_matrix_get_diag_info": [P, C.POINTER(MatrixInfo)],
        "mk_matrix_set_spmv_kernel": [P, C.c_int, C.c_int],
        "mk_matrix_generate_stencil": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.POINTER(P)],
        "mk_matrix_generate_stencil_scaled": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.POINTER(P)],
        "mk_matrix_generate_banded": [C.c_int64, C.c_int, C.c_int64, C.c_int64, C.POINTER(P)],
        "mk_coo_generate_rmat": [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_double, C.POINTER(P)],
        "mk_coo_from_host": [C.POINTER(COO), C.POINTER(P)], "mk_coo_download": [P, C.POINTER(COO)],
        "mk_coo_destroy": [P],
        "mk_vector_create": [C.c_int64, C.POINTER(P)], "mk_vector_set": [P, P],
        "mk_vector_fill": [P, C.c_double], "mk_vector_get": [P, P], "mk_vector_destroy": [P],
        "mk_vector_size": [P],
        "mk_spmv": [P, C.c_double, P, C.c_double, P],
        "mk_spmv_dot": [P, P, P, C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "mk_spmv_dot_static": [P, P, P, C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "mk_pcg_create": [C.POINTER(P)], "mk_pcg_set_max_iter": [P, C.c_int], "mk_pcg_set_tol": [P, C.c_double],
        "mk_pcg_set_two_norm": [P, C.c_int], "mk_pcg_set_rel_change": [P, C.c_int],
        "mk_pcg_set_print_level": [P, C.c_int], "mk_pcg_set_logging": [P, C.c_int],
        "mk_pcg_set_precond": [P, C.c_int], "mk_pcg_set_check_every": [P, C.c_int],
        "mk_pcg_set_kernel_timing": [P, C.c_int],
        "mk_pcg_set_single_reduction": [P, C.c_int], "mk_pcg_get_single_reduction": [P, C.POINTER(C.c_int)],
        "mk_pcg_get_kernel_timing": [P, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int],
        "mk_pcg_setup": [P, P, P, P], "mk_pcg_solve": [P, P, P, P],
        "mk_pcg_get_num_iterations": [P, C.POINTER(C.c_longlong)],
        "mk_pcg_get_final_relative_residual_norm": [P, C.POINTER(C.c_double)],
        "mk_pcg_get_converged": [P, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "mk_pcg_get_residual_history": [P, P, C.c_int], "mk_pcg_destroy": [P],
        "mk_bicgstab_create": [C.POINTER(P)], "mk_bicgstab_set_max_iter": [P, C.c_int], "mk_bicgstab_set_tol": [P, C.c_double],
        "mk_bicgstab_set_print_level": [P, C.c_int], "mk_bicgstab_set_logging": [P, C.c_int],
        "mk_bicgstab_set_check_every": [P, C.c_int],
        "mk_bicgstab_setup": [P, P, P, P], "mk_bicgstab_solve": [P, P, P, P],
        "mk_bicgstab_get_num_iterations": [P, C.POINTER(C.c_longlong)],
        "mk_bicgstab_get_final_relative_residual_norm": [P, C.POINTER(C.c_double)],
        "mk_bicgstab_get_converged": [P, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "mk_bicgstab_get_residual_history": [P, P, C.c_int], "mk_bicgstab_destroy": [P],
        "mk_cgnr_create": [C.POINTER(P)], "mk_cgnr_set_max_iter": [P, C.c_int], "mk_cgnr_set_tol": [P, C.c_double],
        "mk_cgnr_set_logging": [P, C.c_int], "mk_cgnr_set_check_every": [P, C.c_int],
        "mk_cgnr_setup": [P, P, P, P], "mk_cgnr_solve": [P, P, P, P],
        "mk_cgnr_get_num_iterations": [P, C.POINTER(C.c_longlong)],
        "mk_cgnr_get_final_relative_residual_norm": [P, C.POINTER(C.c_double)],
        "mk_cgnr_get_converged": [P, C.POINTER(C.c_int)],
        "mk_cgnr_get_residual_history": [P, P, C.c_int], "mk_cgnr_destroy": [P],
        "mk_gmres_create": [C.POINTER(P)], "mk_gmres_set_max_iter": [P, C.c_int], "mk_gmres_set_k_dim": [P, C.c_int],
        "mk_gmres_set_tol": [P, C.c_double], "mk_gmres_set_print_level": [P, C.c_int], "mk_gmres_set_logging": [P, C.c_int],
        "mk_gmres_setup": [P, P, P, P], "mk_gmres_solve": [P, P, P, P],
        "mk_gmres_get_num_iterations": [P, C.POINTER(C.c_longlong)],
        "mk_gmres_get_final_relative_residual_norm": [P, C.POINTER(C.c_double)],
        "mk_gmres_get_converged": [P, C.POINTER(C.c_int)],
        "mk_gmres_get_residual_history": [P, P, C.c_int], "mk_gmres_destroy": [P],
        "mk_jacobi_create": [C.POINTER(P)], "mk_jacobi_set_max_iter": [P, C.c_int],
        "mk_jacobi_set_tol": [P, C.c_double], "mk_jacobi_set_check_every": [P, C.c_int],
        "mk_jacobi_setup": [P, P, P, P], "mk_jacobi_solve": [P, P, P, P],
        "mk_jacobi_get_num_iterations": [P, C.POINTER(C.c_longlong)],
        "mk_jacobi_get_final_relative_residual_norm": [P, C.POINTER(C.c_double)],
        "mk_jacobi_get_converged": [P, C.POINTER(C.c_int)], "mk_jacobi_destroy": [P],
        "mk_partition_rows": [C.c_int64, P, C.c_int, P],
    }
    for name, args in sig.items():
        getattr(L, name).argtypes = args
    _lib = L
    return L


def _ck(rc):
    if rc < 0:
        raise MkError(f"mk status {rc}: {load().mk_last_error_string().decode()}")
    return rc


_inited = None


def init(device=0):
    global _inited
    if _inited is None:
        _ck(load().mk_init(device))
        _inited = device
    return _inited


def synchronize():
    _ck(load().mk_synchronize())


def stream():
    return load().mk_get_stream()


def memory_info():
    """(bytes inside live objects, bytes idle in the allocator's free list)"""
    a, b = C.c_longlong(), C.c_longlong()
    _ck(load().mk_memory_info(C.byref(a), C.byref(b)))
    return a.value, b.value


def memory_calls():
    """(requests to, releases into) the library's allocator since mk_init"""
    a, b = C.c_longlong(), C.c_longlong()
    _ck(load().mk_memory_calls(C.byref(a), C.byref(b)))
    return a.value, b.value


def memory_trim():
    _ck(load().mk_memory_trim())


def timer_start():
    _ck(load().mk_timer_start())


def timer_stop():
    ms = C.c_float()
    _ck(load().mk_timer_stop(C.byref(ms)))
    return ms.value


def _a(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


def _coo_struct(nrows, ncols, rows, cols, vals):
    rows, cols, vals = _a(rows, np.int32), _a(cols, np.int32), _a(vals, np.float64)
    if not (rows.size == cols.size == vals.size):
        raise MkError("COO arrays differ in length")
    s = COO(vals.ctypes.data_as(C.POINTER(C.c_double)), rows.ctypes.data_as(C.POINTER(C.c_int)),
            cols.ctypes.data_as(C.POINTER(C.c_int)), nrows, ncols, rows.size)
    s._keep = (rows, cols, vals)
    return s


def _take(ptr, n, dt):
    """copy n items out of a malloc'ed C array and free it"""
    if n:
        out = np.ctypeslib.as_array(ptr, (n,)).astype(dt, copy=True)
    else:
        out = np.empty(0, dt)
    _libc.free(C.cast(ptr, C.c_void_p))
    return out


def sort_coo_row(nrows, ncols, rows, cols, vals):
    """sort_coo_row (sparse_matrix.c:419) on the device -> (rows, cols, vals) host arrays"""
    init()
    src, dst = _coo_struct(nrows, ncols, rows, cols, vals), COO()
    _ck(load().mk_sort_coo_row(C.byref(src), C.byref(dst)))
    n = dst.nnz
    return _take(dst.rows, n, np.int32), _take(dst.columns, n, np.int32), _take(dst.values, n, np.float64)


def coo_to_csr(nrows, ncols, rows, cols, vals):
    """convert_coo_to_csr (sparse_matrix.c:649) on the device -> (rowptr, col, val)"""
    init()
    src, dst = _coo_struct(nrows, ncols, rows, cols, vals), CSR()
    _ck(load().mk_coo_to_csr(C.byref(src), C.byref(dst)))
    return (_take(dst.row_pointers, dst.nrows + 1, np.int32), _take(dst.column_indices, dst.nnz, np.int32),
            _take(dst.values, dst.nnz, np.float64))


def partition_rows(rowptr, nparts):
    rp = _a(rowptr, np.int64)
    out = np.empty(nparts + 1, np.int64)
    _ck(load().mk_partition_rows(rp.size - 1, rp.ctypes.data, nparts, out.ctypes.data))
    return out


class Vector:
    def __init__(self, n, fill=None, values=None):
        init()
        self.h = C.c_void_p()
        _ck(load().mk_vector_create(int(n), C.byref(self.h)))
        self.n = int(n)
        if values is not None:
            self.set(values)
        elif fill is not None:
            self.fill(fill)

    def set(self, values):
        v = _a(values, np.float64)
        if v.size != self.n:
            raise MkError("vector length mismatch")
        _ck(load().mk_vector_set(self.h, v.ctypes.data))

    def fill(self, value):
        _ck(load().mk_vector_fill(self.h, float(value)))

    def get(self):
        out = np.empty(self.n)
        _ck(load().mk_vector_get(self.h, out.ctypes.data))
        return out

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mk_vector_destroy(self.h)
            self.h = None


class DeviceCOO:
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def rmat(cls, scale, nedges, seed=42, a=0.57, b=0.19, c=0.19):
        init()
        h = C.c_void_p()
        _ck(load().mk_coo_generate_rmat(scale, nedges, seed, a, b, c, C.byref(h)))
        return cls(h)

    @classmethod
    def from_host(cls, nrows, ncols, rows, cols, vals):
        init()
        h = C.c_void_p()
        s = _coo_struct(nrows, ncols, rows, cols, vals)
        _ck(load().mk_coo_from_host(C.byref(s), C.byref(h)))
        return cls(h)

    def download(self):
        d = COO()
        _ck(load().mk_coo_download(self.h, C.byref(d)))
        n = d.nnz
        return (d.nrows, d.ncolumns, _take(d.rows, n, np.int32), _take(d.columns, n, np.int32),
                _take(d.values, n, np.float64))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mk_coo_destroy(self.h)
            self.h = None


class Matrix:
    def __init__(self, handle):
        self.h = handle
        i = self.info()
        self.nrows, self.ncols, self.nnz = i.nrows, i.ncols, i.nnz

    @classmethod
    def from_coo(cls, nrows, ncols, rows, cols, vals):
        init()
        h = C.c_void_p()
        s = _coo_struct(nrows, ncols, rows, cols, vals)
        _ck(load().mk_matrix_from_coo(C.byref(s), C.byref(h)))
        return cls(h)

    @classmethod
    def from_csr(cls, nrows, ncols, rowptr, col, val):
        init()
        rowptr, col, val = _a(rowptr, np.int32), _a(col, np.int32), _a(val, np.float64)
        if rowptr.size != nrows + 1 or col.size != val.size:
            raise MkError("CSR arrays have inconsistent lengths")
        s = CSR(val.ctypes.data_as(C.POINTER(C.c_double)), col.ctypes.data_as(C.POINTER(C.c_int)),
                rowptr.ctypes.data_as(C.POINTER(C.c_int)), nrows, ncols, col.size)
        h = C.c_void_p()
        _ck(load().mk_matrix_from_csr(C.byref(s), C.byref(h)))
        return cls(h)

    @classmethod
    def from_device_coo(cls, coo):
        h = C.c_void_p()
        _ck(load().mk_matrix_from_device_coo(coo.h, C.byref(h)))
        return cls(h)

    @classmethod
    def stencil(cls, kind, nx, ny, nz=1, row_begin=0, row_end=None, scale_seed=None):
        """scale_seed: D A D with a per-row scale in [1, 5) (variable coefficients), None: the plain stencil"""
        init()
        n = nx * ny * (1 if kind == 5 else nz)
        h = C.c_void_p()
        if scale_seed is None:
            _ck(load().mk_matrix_generate_stencil(kind, nx, ny, nz, row_begin, n if row_end is None else row_end, C.byref(h)))
        else:
            _ck(load().mk_matrix_generate_stencil_scaled(kind, nx, ny, nz, scale_seed, row_begin, n if row_end is None else row_end, C.byref(h)))
        return cls(h)

    @classmethod
    def banded(cls, n, hb=3, row_begin=0, row_end=None):
        init()
        h = C.c_void_p()
        _ck(load().mk_matrix_generate_banded(n, hb, row_begin, n if row_end is None else row_end, C.byref(h)))
        return cls(h)

    def info(self):
        i = MatrixInfo()
        _ck(load().mk_matrix_get_info(self.h, C.byref(i)))
        return i

    def set_spmv_kernel(self, kernel=SPMV_AUTO, lanes_per_row=0):
        _ck(load().mk_matrix_set_spmv_kernel(self.h, kernel, lanes_per_row))

    def to_csr(self):
        d = CSR()
        _ck(load().mk_matrix_get_csr(self.h, C.byref(d)))
        return (_take(d.row_pointers, d.nrows + 1, np.int32), _take(d.column_indices, d.nnz, np.int32),
                _take(d.values, d.nnz, np.float64))

    def spmv(self, x, y, alpha=1.0, beta=0.0):
        _ck(load().mk_spmv(self.h, alpha, x.h, beta, y.h))

    def spmv_dot(self, x, y, want=True):
        if not want:   # enqueue only (timing loops)
            _ck(load().mk_spmv_dot(self.h, x.h, y.h, None, None))
            return None
        a, b = C.c_double(), C.c_double()
        _ck(load().mk_spmv_dot(self.h, x.h, y.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def spmv_dot_static(self, x, y):
        """mk_spmv_dot_static: the same product and dots with all scratch in the library's static device array"""
        a, b = C.c_double(), C.c_double()
        _ck(load().mk_spmv_dot_static(self.h, x.h, y.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mk_matrix_destroy(self.h)
            self.h = None


class PCG:
    """HYPRE_ParCSRPCG* life cycle as solvers.c:729-775 drives it"""

    def __init__(self, max_iter=1000, tol=1e-6, two_norm=1, rel_change=1, precond=PRECOND_NONE,
                 logging=1, print_level=0, check_every=16):
        init()
        self.h = C.c_void_p()
        L = load()
        _ck(L.mk_pcg_create(C.byref(self.h)))
        _ck(L.mk_pcg_set_max_iter(self.h, max_iter))
        _ck(L.mk_pcg_set_tol(self.h, tol))
        _ck(L.mk_pcg_set_two_norm(self.h, two_norm))
        _ck(L.mk_pcg_set_rel_change(self.h, rel_change))
        _ck(L.mk_pcg_set_print_level(self.h, print_level))
        _ck(L.mk_pcg_set_logging(self.h, logging))
        _ck(L.mk_pcg_set_precond(self.h, precond))
        _ck(L.mk_pcg_set_check_every(self.h, check_every))

    def setup(self, A, b, x):
        _ck(load().mk_pcg_setup(self.h, A.h, b.h, x.h))

    def solve(self, A, b, x):
        L = load()
        _ck(L.mk_pcg_solve(self.h, A.h, b.h, x.h))
        it, nrm, cv, bd = C.c_longlong(), C.c_double(), C.c_int(), C.c_int()
        _ck(L.mk_pcg_get_num_iterations(self.h, C.byref(it)))
        _ck(L.mk_pcg_get_final_relative_residual_norm(self.h, C.byref(nrm)))
        _ck(L.mk_pcg_get_converged(self.h, C.byref(cv), C.byref(bd)))
        return dict(iterations=it.value, rel_res_norm=nrm.value, converged=cv.value, breakdown=bd.value)

    def residual_history(self, capacity):
        out = np.empty(capacity)
        n = _ck(load().mk_pcg_get_residual_history(self.h, out.ctypes.data, capacity))
        return out[:n]

    def set_kernel_timing(self, on=True):
        _ck(load().mk_pcg_set_kernel_timing(self.h, int(on)))

    def set_single_reduction(self, on=True):
        """row-block solves: one all-reduce per iteration (Chronopoulos-Gear form of alpha); opt-in"""
        _ck(load().mk_pcg_set_single_reduction(self.h, int(on)))

    def single_reduction_used(self):
        u = C.c_int()
        _ck(load().mk_pcg_get_single_reduction(self.h, C.byref(u)))
        return bool(u.value)

    def kernel_timing(self):
        """[(average ms, launches)] per kernel slot of a CG step, measured by HIP events inside the last solve"""
        ns, ms, cnt = C.c_int(), (C.c_double * 3)(), (C.c_longlong * 3)()
        _ck(load().mk_pcg_get_kernel_timing(self.h, C.byref(ns), ms, cnt, 3))
        return [(ms[k], cnt[k]) for k in range(ns.value)]

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mk_pcg_destroy(self.h)
            self.h = None


class BiCGSTAB:
    """HYPRE_ParCSRBiCGSTAB* life cycle as solvers.c:517-564 drives it (solver id 6)"""

    def __init__(self, max_iter=1000, tol=1e-6, logging=1, print_level=0, check_every=16):
        init()
        self.h = C.c_void_p()
        L = load()
        _ck(L.mk_bicgstab_create(C.byref(self.h)))
        _ck(L.mk_bicgstab_set_max_iter(self.h, max_iter))
        _ck(L.mk_bicgstab_set_tol(self.h, tol))
        _ck(L.mk_bicgstab_set_print_level(self.h, print_level))
        _ck(L.mk_bicgstab_set_logging(self.h, logging))
        _ck(L.mk_bicgstab_set_check_every(self.h, check_every))

    def setup(self, A, b, x):
        _ck(load().mk_bicgstab_setup(self.h, A.h, b.h, x.h))

    def solve(self, A, b, x):
        L = load()
        _ck(L.mk_bicgstab_solve(self.h, A.h, b.h, x.h))
        it, nrm, cv, bd = C.c_longlong(), C.c_double(), C.c_int(), C.c_int()
        _ck(L.mk_bicgstab_get_num_iterations(self.h, C.byref(it)))
        _ck(L.mk_bicgstab_get_final_relative_residual_norm(self.h, C.byref(nrm)))
        _ck(L.mk_bicgstab_get_converged(self.h, C.byref(cv), C.byref(bd)))
        return dict(iterations=it.value, rel_res_norm=nrm.value, converged=cv.value, breakdown=bd.value)

    def residual_history(self, capacity):
        out = np.empty(capacity)
        n = _ck(load().mk_bicgstab_get_residual_history(self.h, out.ctypes.data, capacity))
        return out[:n]

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mk_bicgstab_destroy(self.h)
            self.h = None


class CGNR:
    """HYPRE_ParCSRCGNR* life cycle as solvers.c:864-906 drives it (solver id 40)"""

    def __init__(self, max_iter=1000, tol=1e-6, logging=1, check_every=16):
        init()
        self.h = C.c_void_p()
        L = load()
        _ck(L.mk_cgnr_create(C.byref(self.h)))
        _ck(L.mk_cgnr_set_max_iter(self.h, max_iter))
        _ck(L.mk_cgnr_set_tol(self.h, tol))
        _ck(L.mk_cgnr_set_logging(self.h, logging))
        _ck(L.mk_cgnr_set_check_every(self.h, check_every))

    def setup(self, A, b, x):
        _ck(load().mk_cgnr_setup(self.h, A.h, b.h, x.h))

    def solve(self, A, b, x):
        L = load()
        _ck(L.mk_cgnr_solve(self.h, A.h, b.h, x.h))
        it, nrm, cv = C.c_longlong(), C.c_double(), C.c_int()
        _ck(L.mk_cgnr_get_num_iterations(self.h, C.byref(it)))
        _ck(L.mk_cgnr_get_final_relative_residual_norm(self.h, C.byref(nrm)))
        _ck(L.mk_cgnr_get_converged(self.h, C.byref(cv)))
        return dict(iterations=it.value, rel_res_norm=nrm.value, converged=cv.value)

    def residual_history(self, capacity):
        out = np.empty(capacity)
        n = _ck(load().mk_cgnr_get_residual_history(self.h, out.ctypes.data, capacity))
        return out[:n]

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mk_cgnr_destroy(self.h)
            self.h = None


class GMRES:
    """HYPRE_ParCSRGMRES* life cycle as solvers.c:1183-1245 drives it (solver id 12, restart 30)"""

    def __init__(self, max_iter=1000, tol=1e-6, k_dim=30, logging=1, print_level=0):
        init()
        self.h = C.c_void_p()
        L = load()
        _ck(L.mk_gmres_create(C.byref(self.h)))
        _ck(L.mk_gmres_set_max_iter(self.h, max_iter))
        _ck(L.mk_gmres_set_k_dim(self.h, k_dim))
        _ck(L.mk_gmres_set_tol(self.h, tol))
        _ck(L.mk_gmres_set_print_level(self.h, print_level))
        _ck(L.mk_gmres_set_logging(self.h, logging))

    def setup(self, A, b, x):
        _ck(load().mk_gmres_setup(self.h, A.h, b.h, x.h))

    def solve(self, A, b, x):
        L = load()
        _ck(L.mk_gmres_solve(self.h, A.h, b.h, x.h))
        it, nrm, cv = C.c_longlong(), C.c_double(), C.c_int()
        _ck(L.mk_gmres_get_num_iterations(self.h, C.byref(it)))
        _ck(L.mk_gmres_get_final_relative_residual_norm(self.h, C.byref(nrm)))
        _ck(L.mk_gmres_get_converged(self.h, C.byref(cv)))
        return dict(iterations=it.value, rel_res_norm=nrm.value, converged=cv.value)

    def residual_history(self, capacity):
        out = np.empty(capacity)
        n = _ck(load().mk_gmres_get_residual_history(self.h, out.ctypes.data, capacity))
        return out[:n]

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mk_gmres_destroy(self.h)
            self.h = None


class Jacobi:
    def __init__(self, max_iter=1000, tol=1e-6, check_every=10):
        init()
        self.h = C.c_void_p()
        L = load()
        _ck(L.mk_jacobi_create(C.byref(self.h)))
        _ck(L.mk_jacobi_set_max_iter(self.h, max_iter))
        _ck(L.mk_jacobi_set_tol(self.h, tol))
        _ck(L.mk_jacobi_set_check_every(self.h, check_every))

    def setup(self, A, b, x):
        _ck(load().mk_jacobi_setup(self.h, A.h, b.h, x.h))

    def solve(self, A, b, x):
        L = load()
        _ck(L.mk_jacobi_solve(self.h, A.h, b.h, x.h))
        it, nrm, cv = C.c_longlong(), C.c_double(), C.c_int()
        _ck(L.mk_jacobi_get_num_iterations(self.h, C.byref(it)))
        _ck(L.mk_jacobi_get_final_relative_residual_norm(self.h, C.byref(nrm)))
        _ck(L.mk_jacobi_get_converged(self.h, C.byref(cv)))
        return dict(iterations=it.value, rel_res_norm=nrm.value, converged=cv.value)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mk_jacobi_destroy(self.h)
            self.h = None
