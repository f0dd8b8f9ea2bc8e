"""Host side of the boundary: the loader (host/sparse_matrix.c, host/mmio.c) against the golden
COO arrays the reference's own loader produced, and the driver binary (host/solvers.c) with the
reference's flags and output lines."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import iterative_solvers_amd as mk
import oracle
from conftest import GOLDEN, GOLDEN_CASES, ROOT

HOST = os.path.join(ROOT, "host")


def _host():
    if not os.path.exists(os.path.join(HOST, "libmkhost.so")):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    mk.load()
    L = C.CDLL(os.path.join(HOST, "libmkhost.so"))
    L.fast_load_from_mtx_file.argtypes = [C.c_char_p, C.POINTER(mk.COO)]
    L.load_array_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_uint)]
    L.fast_load_from_array_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_double)), C.c_int]   # the reference's signature
    return L


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_loader_matches_reference_loader(name):
    L = _host()
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = mk.COO()
    assert L.fast_load_from_mtx_file(os.path.join(GOLDEN, name + ".mtx").encode(), C.byref(m)) == 0
    assert (m.nrows, m.ncolumns, m.nnz) == (int(g["nrows"]), int(g["ncols"]), g["coo_rows"].size)
    n = m.nnz
    if n:
        assert np.array_equal(np.ctypeslib.as_array(m.rows, (n,)), g["coo_rows"])
        assert np.array_equal(np.ctypeslib.as_array(m.columns, (n,)), g["coo_cols"])
        assert np.array_equal(np.ctypeslib.as_array(m.values, (n,)).view(np.uint64), g["coo_vals"].view(np.uint64))


def _load(L, path):
    m = mk.COO()
    rc = L.fast_load_from_mtx_file(str(path).encode(), C.byref(m))
    if rc != 0:
        return rc, None
    n = m.nnz
    take = lambda p, dt: np.ctypeslib.as_array(p, (n,)).astype(dt, copy=True) if n else np.empty(0, dt)
    return 0, (m.nrows, m.ncolumns, take(m.rows, np.int32), take(m.columns, np.int32), take(m.values, np.float64))


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_parallel_parser_matches_reference_loader(name, monkeypatch):
    # the multi-threaded entry parser, forced on for the small fixtures
    L = _host()
    monkeypatch.setenv("MK_PARSE_THREADS", "5")
    monkeypatch.setenv("MK_PARSE_MIN_ENTRIES", "0")
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rc, got = _load(L, os.path.join(GOLDEN, name + ".mtx"))
    assert rc == 0
    assert np.array_equal(got[2], g["coo_rows"]) and np.array_equal(got[3], g["coo_cols"])
    assert np.array_equal(got[4].view(np.uint64), g["coo_vals"].view(np.uint64))


@pytest.mark.parametrize("field,sym", [("real", "general"), ("real", "symmetric"), ("pattern", "general"), ("pattern", "symmetric")])
def test_parallel_parser_equals_serial_parser(tmp_path, monkeypatch, field, sym):
    L = _host()
    rng = np.random.default_rng(11)
    n, m = 5000, 60000
    r = rng.integers(1, n + 1, m)
    c = rng.integers(1, n + 1, m)
    if sym == "symmetric":
        r, c = np.maximum(r, c), np.minimum(r, c)
    v = rng.normal(size=m) * 10.0 ** rng.integers(-200, 200, m)
    p = tmp_path / "big.mtx"
    with open(p, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate {field} {sym}\n% comment line\n{n} {n} {m}\n")
        for i in range(m):
            pad = " " * int(rng.integers(1, 4))
            f.write(f"{r[i]}{pad}{c[i]}\n" if field == "pattern" else f"{r[i]}{pad}{c[i]}\t{float(v[i])!r}\r\n" if i % 97 == 0 else f"{r[i]} {c[i]} {v[i]:.17g}\n")
            if i % 5000 == 0:
                f.write("\n")                       # blank lines are legal
    monkeypatch.setenv("MK_PARSE_MIN_ENTRIES", "0")
    monkeypatch.setenv("MK_PARSE_THREADS", "1")
    rc1, a = _load(L, p)
    for t in ("2", "7", "16"):
        monkeypatch.setenv("MK_PARSE_THREADS", t)
        rc2, b = _load(L, p)
        assert rc1 == 0 and rc2 == 0 and a[:2] == b[:2]
        for u, w in zip(a[2:], b[2:]):
            assert np.array_equal(u.view(np.uint8), w.view(np.uint8))
    if oracle.have_ref():                         # and the serial parser equals the reference's loader
        rr = oracle.ref_load_mtx(str(p))
        assert np.array_equal(rr[2], a[2]) and np.array_equal(rr[3], a[3]) and np.array_equal(rr[4].view(np.uint64), a[4].view(np.uint64))


def test_parallel_parser_falls_back_on_odd_layouts(tmp_path, monkeypatch):
    L = _host()
    monkeypatch.setenv("MK_PARSE_MIN_ENTRIES", "0")
    monkeypatch.setenv("MK_PARSE_THREADS", "4")
    # entries spread over several lines / several entries per line: legal token stream for the reference
    p = tmp_path / "tokens.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 4\n1 1\n1.5 2 2 2.5\n3 3 3.5 1\n3 -1\n")
    rc, got = _load(L, p)
    assert rc == 0 and got[2].tolist() == [0, 1, 2, 0] and got[3].tolist() == [0, 1, 2, 2] and got[4].tolist() == [1.5, 2.5, 3.5, -1.0]
    # more lines than the header announces: the first `entries` count, the rest is ignored
    p = tmp_path / "extra.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n2 2 2.0\n3 3 3.0\n3 1 4.0\n")
    rc, got = _load(L, p)
    assert rc == 0 and got[4].tolist() == [1.0, 2.0]
    # fewer lines than announced: an error, not a crash
    p = tmp_path / "short.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 5\n1 1 1.0\n2 2 2.0\n")
    rc, got = _load(L, p)
    assert rc != 0


def test_csr_cache_round_trip(tmp_path):
    L = _host()
    L.write_csr_cache.argtypes = [C.c_char_p, C.POINTER(mk.CSR)]
    L.read_csr_cache.argtypes = [C.c_char_p, C.POINTER(mk.CSR)]
    g = np.load(os.path.join(GOLDEN, "random_70k.npz"))
    rp, ci, v = (np.ascontiguousarray(g[k]) for k in ("csr_rowptr", "csr_col", "csr_val"))
    m = mk.CSR(v.ctypes.data_as(C.POINTER(C.c_double)), ci.ctypes.data_as(C.POINTER(C.c_int)),
               rp.ctypes.data_as(C.POINTER(C.c_int)), int(g["nrows"]), int(g["ncols"]), ci.size)
    path = str(tmp_path / "a.mkcsr").encode()
    assert L.write_csr_cache(path, C.byref(m)) == 0
    r = mk.CSR()
    assert L.read_csr_cache(path, C.byref(r)) == 0
    assert (r.nrows, r.ncolumns, r.nnz) == (m.nrows, m.ncolumns, m.nnz)
    assert np.array_equal(np.ctypeslib.as_array(r.row_pointers, (r.nrows + 1,)), rp)
    assert np.array_equal(np.ctypeslib.as_array(r.column_indices, (r.nnz,)), ci)
    assert np.array_equal(np.ctypeslib.as_array(r.values, (r.nnz,)).view(np.uint64), v.view(np.uint64))
    raw = bytearray(open(path, "rb").read())
    raw[64 + 40] ^= 1                                     # flip one bit of a row pointer
    open(path, "wb").write(raw)
    assert L.read_csr_cache(path, C.byref(r)) != 0        # checksum catches it
    assert L.read_csr_cache(str(tmp_path / "absent").encode(), C.byref(r)) != 0


def test_csr_cache_checks_values_and_source_file(tmp_path):
    """the checksum covers the value bytes, and a cache made from a .mtx that has changed since is refused"""
    L = _host()
    for f in (L.write_csr_cache_for, L.read_csr_cache_for):
        f.argtypes = [C.c_char_p, C.POINTER(mk.CSR), C.c_char_p]
    g = np.load(os.path.join(GOLDEN, "poisson5_6x5_sorted.npz"))
    rp, ci, v = (np.ascontiguousarray(g[k]) for k in ("csr_rowptr", "csr_col", "csr_val"))
    m = mk.CSR(v.ctypes.data_as(C.POINTER(C.c_double)), ci.ctypes.data_as(C.POINTER(C.c_int)),
               rp.ctypes.data_as(C.POINTER(C.c_int)), int(g["nrows"]), int(g["ncols"]), ci.size)
    src = tmp_path / "a.mtx"
    src.write_text(open(os.path.join(GOLDEN, "poisson5_6x5_sorted.mtx")).read())
    path = str(tmp_path / "a.mtx.mkcsr").encode()
    assert L.write_csr_cache_for(path, C.byref(m), str(src).encode()) == 0
    r = mk.CSR()
    assert L.read_csr_cache_for(path, C.byref(r), str(src).encode()) == 0
    assert L.read_csr_cache_for(path, C.byref(r), None) == 0
    # same size, later modification time: stale
    st = os.stat(src)
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns + 2_000_000_000))
    assert L.read_csr_cache_for(path, C.byref(r), str(src).encode()) == 4
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert L.read_csr_cache_for(path, C.byref(r), str(src).encode()) == 0
    src.write_text(src.read_text() + "\n")                 # other size
    assert L.read_csr_cache_for(path, C.byref(r), str(src).encode()) == 4
    # one flipped bit in a value, two values swapped: both fail the checksum
    raw = bytearray(open(path, "rb").read())
    pad64 = lambda b: (b + 63) // 64 * 64
    voff = 64 + pad64(4 * (m.nrows + 1)) + pad64(4 * m.nnz)
    bad = bytearray(raw)
    bad[voff + 8 * 3] ^= 1
    open(path, "wb").write(bad)
    assert L.read_csr_cache_for(path, C.byref(r), None) == 3
    bad = bytearray(raw)
    bad[voff:voff + 8], bad[voff + 16:voff + 24] = raw[voff + 16:voff + 24], raw[voff:voff + 8]    # -1 and 4: differ
    assert bad != raw
    open(path, "wb").write(bad)
    assert L.read_csr_cache_for(path, C.byref(r), None) == 3


@pytest.mark.parametrize("name", ["random_70k", "ragged_dups_empty_rows", "skewed_one_long_row", "empty_matrix"])
@pytest.mark.parametrize("np_ranks", [1, 2, 5])
def test_row_block_slicing_of_the_np_driver(name, np_ranks):
    """host side of `solver -np N` (solvers.c:214-281 without its defects): row pointers of the unsorted COO, the cuts
    of mk_partition_rows, and each rank's entries in input order with rows renumbered from 0"""
    L = _host()
    L.coo_row_pointers64.argtypes = [C.POINTER(mk.COO), np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")]
    L.slice_coo_rows.argtypes = [C.POINTER(mk.COO), C.c_longlong, C.c_longlong, C.POINTER(mk.COO)]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rows, cols, vals = g["coo_rows"], g["coo_cols"], g["coo_vals"]
    n = int(g["nrows"])
    src = mk._coo_struct(n, int(g["ncols"]), rows, cols, vals)
    rp = np.zeros(n + 1, np.int64)
    assert L.coo_row_pointers64(C.byref(src), rp) == 0
    assert np.array_equal(rp, g["csr_rowptr"].astype(np.int64))              # the reference's own row pointers
    rs = mk.partition_rows(rp, np_ranks)
    assert np.array_equal(rs, oracle.partition_rows(rp, np_ranks)) and rs[0] == 0 and rs[-1] == n
    seen = 0
    for r in range(np_ranks):
        lo, hi = int(rs[r]), int(rs[r + 1])
        out = mk.COO()
        assert L.slice_coo_rows(C.byref(src), lo, hi, C.byref(out)) == 0
        keep = (rows >= lo) & (rows < hi)
        assert (out.nrows, out.ncolumns, out.nnz) == (hi - lo, int(g["ncols"]), int(keep.sum()))
        k = out.nnz
        if k:
            assert np.array_equal(np.ctypeslib.as_array(out.rows, (k,)), rows[keep] - lo)
            assert np.array_equal(np.ctypeslib.as_array(out.columns, (k,)), cols[keep])
            assert np.array_equal(np.ctypeslib.as_array(out.values, (k,)).view(np.uint64), vals[keep].view(np.uint64))
        seen += k
    assert seen == rows.size
    bad = mk.COO()
    assert L.slice_coo_rows(C.byref(src), 3, 2, C.byref(bad)) != 0 and L.slice_coo_rows(C.byref(src), 0, n + 1, C.byref(bad)) != 0


def test_loader_errors_are_returned_not_fatal(tmp_path):
    L = _host()
    m = mk.COO()
    assert L.fast_load_from_mtx_file(str(tmp_path / "missing.mtx").encode(), C.byref(m)) != 0
    p = tmp_path / "complex.mtx"
    p.write_text("%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1.0 0.0\n")
    assert L.fast_load_from_mtx_file(str(p).encode(), C.byref(m)) != 0
    p = tmp_path / "short.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 4\n1 1 1.0\n2 2 2.0\n")
    assert L.fast_load_from_mtx_file(str(p).encode(), C.byref(m)) != 0
    p = tmp_path / "array.mtx"
    p.write_text("%%MatrixMarket matrix array real general\n% rhs\n3 1\n1.5\n-2\n3e2\n")
    v, cnt = C.POINTER(C.c_double)(), C.c_uint()
    assert L.load_array_file(str(p).encode(), C.byref(v), C.byref(cnt)) == 0
    assert cnt.value == 3 and [v[i] for i in range(3)] == [1.5, -2.0, 300.0]
    v2 = C.POINTER(C.c_double)()
    assert L.fast_load_from_array_file(str(p).encode(), C.byref(v2), 1) == 0 and [v2[i] for i in range(3)] == [1.5, -2.0, 300.0]
    assert L.fast_load_from_array_file(str(p).encode(), C.byref(v2), 2) != 0          # the reference's dead 'extend' factor


def _write_poisson(path, nx, ny, shuffle_seed=None):
    rp, ci, v = oracle.stencil_csr(5, nx, ny)
    n = rp.size - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    order = np.arange(rows.size) if shuffle_seed is None else np.random.default_rng(shuffle_seed).permutation(rows.size)
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{n} {n} {rows.size}\n")
        for k in order:
            f.write(f"{rows[k] + 1} {ci[k] + 1} {v[k]:.1f}\n")
    return rp, ci, v


@pytest.mark.gpu
@pytest.mark.parametrize("solver,precond", [(50, 0), (51, 1)])
def test_driver_pcg_output_lines(tmp_path, solver, precond):
    # BASELINE config 1 shape scaled down: synthetic .mtx through the mmio path, solver id 50
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    mtx = tmp_path / "poisson.mtx"
    rp, ci, v = _write_poisson(mtx, 64, 48, shuffle_seed=1)
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", str(solver), "-mtx", str(mtx), "-tol", "1e-8",
                          "-maxit", "500", "-niter", "1", "-log", "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert re.search(r"^PCG Solve time = [0-9.]+ seconds$", out.stdout, re.M)
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    nrm = float(re.search(r"^Final Relative Residual Norm = ([0-9.e+-]+)$", out.stdout, re.M).group(1))
    o = oracle.pcg(rp, ci, v, np.ones(rp.size - 1), tol=1e-8, precond=precond, maxit=500)
    assert abs(it - o["iterations"]) <= 2 and nrm < 1e-8
    assert "Finish reading and sorting of COO data" in out.stderr


@pytest.mark.gpu
def test_driver_pcg_prints_the_iteration_table_at_the_default_print_level(tmp_path):
    """the reference driver's default is prnt_info = 2 (solvers.c: HYPRE_PCGSetPrintLevel), at which hypre_PCGSolve prints one
    row per iteration: Iters, ||r||_2, conv.rate, ||r||_2/||b||_2.  Parity unpinned (no HYPRE here): the layout follows the
    published pcg.c, the numbers are checked against the oracle's history."""
    mtx = tmp_path / "poisson.mtx"
    rp, ci, v = _write_poisson(mtx, 40, 30, shuffle_seed=2)
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", "50", "-mtx", str(mtx), "-tol", "1e-8", "-maxit", "500"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Iters       ||r||_2     conv.rate  ||r||_2/||b||_2\n-----    ------------    ---------  ------------ \n" in out.stdout
    rows = re.findall(r"^ *(\d+)    ([0-9.e+-]+)    ([0-9.]+)    ([0-9.e+-]+)$", out.stdout, re.M)
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    assert [int(r[0]) for r in rows] == list(range(1, it + 1))
    n = rp.size - 1
    o = oracle.pcg(rp, ci, v, np.ones(n), tol=1e-8, maxit=500, want_norms=True)
    rel = np.array([float(r[3]) for r in rows])
    k = min(it, o["iterations"])
    assert np.allclose(rel[:k], o["rel_norms"][1:k + 1], rtol=1e-5)
    absn = np.array([float(r[1]) for r in rows])
    assert np.allclose(absn, rel * np.sqrt(n), rtol=1e-5)                      # ||b||_2 = sqrt(n) for b = 1
    rate = np.array([float(r[2]) for r in rows])
    assert np.allclose(rate[1:], absn[1:] / absn[:-1], atol=2e-6 + 1e-5 * rate[1:])


@pytest.mark.gpu
def test_driver_bicgstab_output_lines(tmp_path):
    # solver id 6 (solvers.c:517-564) on the same synthetic .mtx path
    mtx = tmp_path / "poisson.mtx"
    rp, ci, v = _write_poisson(mtx, 48, 40, shuffle_seed=3)
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", "6", "-mtx", str(mtx), "-tol", "1e-8",
                          "-maxit", "500", "-log", "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert re.search(r"^ParCSRBiCGSTABC Solve time = [0-9.]+ seconds$", out.stdout, re.M)     # the reference's spelling (:550)
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    nrm = float(re.search(r"^Final Relative Residual Norm = ([0-9.e+-]+)$", out.stdout, re.M).group(1))
    o = oracle.bicgstab(rp, ci, v, np.ones(rp.size - 1), tol=1e-8, maxit=500)
    # BiCGSTAB's irregular convergence magnifies the different summation order of the dot products:
    # the bar for the count is 5 % (at least 3 iterations), the residual itself must meet the tolerance
    assert abs(it - o["iterations"]) <= max(3, 0.05 * o["iterations"]) and nrm <= 1e-8


@pytest.mark.gpu
def test_driver_power_sampling(tmp_path):
    """-p <file>: one number (watts) per line while the solver runs, as the reference's nvidia-smi loop writes
    (cpower.h:124-130); read from the amdgpu hwmon node here.  A box that hides the node makes the driver say
    so and solve without sampling."""
    mtx = tmp_path / "poisson.mtx"
    _write_poisson(mtx, 200, 200)
    pw = tmp_path / "power.txt"
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", "50", "-mtx", str(mtx), "-tol", "1e-12", "-maxit", "20000",
                          "-niter", "3", "-p", str(pw)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert re.search(r"^Iterations = \d+$", out.stdout, re.M)
    if "power sampling:" in out.stderr:
        pytest.skip("no readable hwmon power node on this box: " + out.stderr.strip().splitlines()[-1])
    m = re.search(r"^Power samples = (\d+), mean board power = ([0-9.]+) W, energy = ([0-9.]+) J$", out.stdout, re.M)
    assert m and int(m.group(1)) >= 1
    vals = [float(l) for l in open(pw).read().split()]
    assert len(vals) == int(m.group(1)) and all(10.0 < v < 2000.0 for v in vals)
    assert abs(np.mean(vals) - float(m.group(2))) < 0.1


@pytest.mark.gpu
def test_driver_cgnr_output_lines(tmp_path):
    # solver id 40 (solvers.c:864-906)
    mtx = tmp_path / "poisson.mtx"
    rp, ci, v = _write_poisson(mtx, 24, 20, shuffle_seed=4)
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", "40", "-mtx", str(mtx), "-tol", "1e-8",
                          "-maxit", "5000"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert re.search(r"^CGNR Solve time = [0-9.]+ seconds$", out.stdout, re.M)
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    nrm = float(re.search(r"^Final Relative Residual Norm = ([0-9.e+-]+)$", out.stdout, re.M).group(1))
    o = oracle.cgnr(rp, ci, v, np.ones(rp.size - 1), tol=1e-8, maxit=5000)
    assert abs(it - o["iterations"]) <= max(2, 0.02 * o["iterations"]) and nrm < 1e-8


@pytest.mark.gpu
def test_driver_gmres_output_lines(tmp_path):
    # solver id 12 (solvers.c:1183-1245), restart 30
    mtx = tmp_path / "poisson.mtx"
    rp, ci, v = _write_poisson(mtx, 48, 40, shuffle_seed=5)
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", "12", "-mtx", str(mtx), "-tol", "1e-8",
                          "-maxit", "2000", "-log", "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert re.search(r"^GMRES Setup time = [0-9.]+ seconds$", out.stdout, re.M)
    assert re.search(r"^GMRES Solve time = [0-9.]+ seconds$", out.stdout, re.M)
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    nrm = float(re.search(r"^Final Relative Residual Norm = ([0-9.e+-]+)$", out.stdout, re.M).group(1))
    o = oracle.gmres(rp, ci, v, np.ones(rp.size - 1), tol=1e-8, maxit=2000, k_dim=30)
    assert abs(it - o["iterations"]) <= max(2, 0.02 * o["iterations"]) and nrm <= 1e-8


@pytest.mark.gpu
def test_driver_binary_cache(tmp_path):
    mtx = tmp_path / "poisson.mtx"
    _write_poisson(mtx, 40, 40, shuffle_seed=2)
    outs = []
    for _ in range(2):
        o = subprocess.run([os.path.join(HOST, "solver"), "-solver", "50", "-mtx", str(mtx), "-cache", "-tol", "1e-9"],
                           capture_output=True, text=True, timeout=120)
        assert o.returncode == 0, o.stderr
        outs.append(o)
    assert os.path.exists(str(mtx) + ".mkcsr") and "Loaded binary CSR cache" in outs[1].stderr
    pick = lambda o: re.findall(r"^(Iterations = \d+|Final Relative Residual Norm = \S+)$", o.stdout, re.M)
    assert pick(outs[0]) == pick(outs[1]) and len(pick(outs[0])) == 2


@pytest.mark.gpu
def test_driver_rejects_other_solver_ids(tmp_path):
    mtx = tmp_path / "p.mtx"
    _write_poisson(mtx, 8, 8)
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", "1", "-mtx", str(mtx)], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "Invalid solver id specified." in out.stdout


@pytest.mark.gpu
def test_driver_jacobi_and_vector_files(tmp_path):
    n = 500
    rp, ci, v = oracle.banded_csr(n, 3)
    mtx = tmp_path / "band.mtx"
    rows = np.repeat(np.arange(n), np.diff(rp))
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{n} {n} {rows.size}\n")
        for k in range(rows.size):
            f.write(f"{rows[k] + 1} {ci[k] + 1} {v[k]}\n")
    bh = np.linspace(-1, 1, n)
    bfile = tmp_path / "b.mtx"
    bfile.write_text("%%MatrixMarket matrix array real general\n" + f"{n} 1\n" + "\n".join(repr(float(t)) for t in bh) + "\n")
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", "70", "-mtx", str(mtx), "-b", str(bfile), "-tol", "1e-9",
                          "-maxit", "300"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    o = oracle.jacobi(rp, ci, v, bh, tol=1e-9, maxit=300, check_every=10)
    assert it == o["iterations"]


@pytest.mark.gpu
@pytest.mark.parametrize("solver,precond", [(50, 0), (51, 1), (70, None)])
def test_driver_row_block_path_one_rank(tmp_path, solver, precond):
    """`solver -dist`: the np > 1 code path of the driver (slice, mk_dist_matrix_create over a real RCCL communicator,
    mk_dist_*_solve, all-reduced wall time) with a world of one rank -- RCCL refuses two ranks on one GPU"""
    mtx = tmp_path / "poisson.mtx"
    if solver == 70:
        n = 2000
        rp, ci, v = oracle.banded_csr(n, 3)
        rows = np.repeat(np.arange(n), np.diff(rp))
        with open(mtx, "w") as f:
            f.write("%%MatrixMarket matrix coordinate real general\n" + f"{n} {n} {rows.size}\n")
            for k in np.random.default_rng(0).permutation(rows.size):
                f.write(f"{rows[k] + 1} {ci[k] + 1} {v[k]}\n")
        o = oracle.jacobi(rp, ci, v, np.ones(n), tol=1e-8, maxit=500, check_every=10)
    else:
        rp, ci, v = _write_poisson(mtx, 56, 40, shuffle_seed=6)
        o = oracle.pcg(rp, ci, v, np.ones(rp.size - 1), tol=1e-8, precond=precond, maxit=500)
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", str(solver), "-mtx", str(mtx), "-tol", "1e-8", "-maxit", "500",
                          "-log", "0", "-dist", "-id_file", str(tmp_path / "id.bin"), "-print_system"],
                         capture_output=True, text=True, timeout=180, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    assert re.search(r"^(PCG|Jacobi) Solve time = [0-9.]+ seconds$", out.stdout, re.M) and "Ranks = 1" in out.stdout
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    nrm = float(re.search(r"^Final Relative Residual Norm = ([0-9.e+-]+)$", out.stdout, re.M).group(1))
    assert abs(it - o["iterations"]) <= (0 if solver == 70 else 2) and nrm < 1e-8
    assert not os.path.exists(tmp_path / "id.bin")          # rank 0 removes the rendezvous file
    # -print_system: HYPRE's IJ print layout, one file per rank (solvers.c:450-454)
    a = open(tmp_path / "IJ.out.A.00000").read().split("\n")
    n = rp.size - 1
    assert a[0].split() == ["0", str(n - 1), "0", str(n - 1)] and len(a) - 2 == ci.size
    trip = np.array([l.split() for l in a[1:-1]], dtype=float)
    # the driver hands the COO of the file to the device sort: stable by row, so row r holds the file's order
    assert np.array_equal(trip[:, 0].astype(int), np.repeat(np.arange(n), np.diff(rp)))
    bfile = open(tmp_path / "IJ.out.b.00000").read().split("\n")
    assert bfile[0].split() == ["0", str(n - 1)] and float(bfile[1].split()[1]) == 1.0 and len(bfile) - 2 == n
    assert os.path.exists(tmp_path / "IJ.out.x.00000")


@pytest.mark.gpu
@pytest.mark.parametrize("solver,precond,dims,auto_min", [(50, 0, (256, 200), "1"), (51, 1, (256, 200), "1"), (51, 1, (56, 40), None)])
def test_driver_single_reduction_flag(tmp_path, solver, precond, dims, auto_min):
    """`solver -dist -single_reduction`: mk_pcg_set_single_reduction behind the driver; with the diag block on the sweep kernel
    (MK_SWEEP_AUTO_MIN=1: grids this small would take the pattern kernel) the step is taken, else declined and the default
    step runs: same answer either way"""
    mtx = tmp_path / "poisson.mtx"
    rp, ci, v = _write_poisson(mtx, *dims)
    o = oracle.pcg(rp, ci, v, np.ones(rp.size - 1), tol=1e-8, precond=precond, maxit=3000)
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", str(solver), "-mtx", str(mtx), "-tol", "1e-8", "-maxit", "3000",
                          "-log", "0", "-dist", "-single_reduction", "-id_file", str(tmp_path / "id.bin")],
                         capture_output=True, text=True, timeout=180, cwd=tmp_path,
                         env=dict(os.environ, **({"MK_SWEEP_AUTO_MIN": auto_min} if auto_min else {})))
    assert out.returncode == 0, out.stderr
    m = re.search(r"^CG step = (single|two) reduction", out.stdout, re.M)
    assert m and (m.group(1) == "single") == (auto_min is not None), out.stdout
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    nrm = float(re.search(r"^Final Relative Residual Norm = ([0-9.e+-]+)$", out.stdout, re.M).group(1))
    assert abs(it - o["iterations"]) <= 2 and nrm < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("solver,label", [(6, "ParCSRBiCGSTABC"), (12, "GMRES"), (40, "CGNR")])
def test_driver_row_block_path_one_rank_other_krylov_solvers(tmp_path, solver, label):
    """`solver -dist -solver 6|12|40`: mk_dist_bicgstab_solve / mk_dist_gmres_solve / mk_dist_cgnr_solve behind the np > 1 code
    path of the driver, over a real one-rank RCCL communicator"""
    mtx = tmp_path / "poisson.mtx"
    dims = (24, 20) if solver == 40 else (48, 40)
    rp, ci, v = _write_poisson(mtx, *dims, shuffle_seed=7)
    ones = np.ones(rp.size - 1)
    o = {6: lambda: oracle.bicgstab(rp, ci, v, ones, tol=1e-8, maxit=5000),
         12: lambda: oracle.gmres(rp, ci, v, ones, tol=1e-8, maxit=5000, k_dim=30),
         40: lambda: oracle.cgnr(rp, ci, v, ones, tol=1e-8, maxit=5000)}[solver]()
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", str(solver), "-mtx", str(mtx), "-tol", "1e-8", "-maxit", "5000",
                          "-log", "0", "-dist", "-id_file", str(tmp_path / "id.bin")], capture_output=True, text=True, timeout=180, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    assert re.search(rf"^{label} Solve time = [0-9.]+ seconds$", out.stdout, re.M) and "Ranks = 1" in out.stdout
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    nrm = float(re.search(r"^Final Relative Residual Norm = ([0-9.e+-]+)$", out.stdout, re.M).group(1))
    assert abs(it - o["iterations"]) <= max(3, 0.05 * o["iterations"]) and nrm <= 1e-8


@pytest.mark.gpu
def test_driver_rejects_stale_cache(tmp_path):
    mtx = tmp_path / "poisson.mtx"
    _write_poisson(mtx, 20, 20)
    run = lambda: subprocess.run([os.path.join(HOST, "solver"), "-solver", "50", "-mtx", str(mtx), "-cache", "-tol", "1e-9", "-log", "0"],
                                 capture_output=True, text=True, timeout=120)
    first = run()
    assert first.returncode == 0 and os.path.exists(str(mtx) + ".mkcsr")
    _write_poisson(mtx, 24, 20)                             # another matrix under the same name
    second = run()
    assert second.returncode == 0 and "Loaded binary CSR cache" not in second.stderr
    third = run()
    assert "Loaded binary CSR cache" in third.stderr
    pick = lambda o: re.findall(r"^(Iterations = \d+|Final Relative Residual Norm = \S+)$", o.stdout, re.M)
    assert pick(second) == pick(third) != pick(first)


@pytest.mark.gpu
def test_bench_cpu_configs_mode_runs_config_1():
    """bench.py --cpu-configs c1: the oracle's CPU solve and the GPU solve of BASELINE configs[0] side by side;
    both must take the same number of iterations to the same residual"""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-configs", "c1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["rows"] == 1024 * 1024 and rec["omp_threads"] >= 1 and rec["cpu_model"]
    for tol in ("tol1e-06", "tol1e-10"):
        t = rec[tol]
        assert abs(t["cpu_iterations"] - t["gpu_iterations"]) <= 2
        assert t["gpu_it_per_s"] > t["cpu_it_per_s"] > 0
        assert t["gpu_rel_res"] < float(tol[3:]) and t["cpu_rel_res"] < float(tol[3:])


@pytest.mark.gpu
def test_driver_config_1_full_size_through_the_mtx_path(tmp_path):
    """BASELINE configs[0] as the reference runs it (sparse_matrix.c:228-376 -> solvers.c -solver 50): 2D 5-point Poisson
    1024^2 written as a Matrix Market file (5.2 M entries), read back by the loader -- large enough that the PARALLEL parser
     is the one that runs, which the trace line proves --, sorted and converted on the device, solved by
    unpreconditioned CG at the reference's default tolerance; iteration count within 2 of the oracle's orc_pcg on the same
    operator (parity unpinned: no HYPRE here), residual below the tolerance."""
    L = _host()
    L.write_stencil_mtx.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    mtx = tmp_path / "poisson1024.mtx"
    assert L.write_stencil_mtx(str(mtx).encode(), 5, 1024, 1024, 1) == 0
    env = dict(os.environ, MK_PARSE_TRACE="1")
    env.pop("MK_PARSE_MIN_ENTRIES", None)
    env.pop("MK_PARSE_THREADS", None)
    env.setdefault("OMP_NUM_THREADS", "8")
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", "50", "-mtx", str(mtx), "-tol", "1e-6", "-maxit", "5000", "-log", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert re.search(r"Matrix Market parser: parallel \(5238784 entries, \d+ threads\)", out.stderr), out.stderr[-2000:]
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    nrm = float(re.search(r"^Final Relative Residual Norm = ([0-9.e+-]+)$", out.stdout, re.M).group(1))
    rp, ci, v = oracle.stencil_csr(5, 1024, 1024, 1)
    o = oracle.pcg(rp, ci, v, np.ones(rp.size - 1), tol=1e-6, maxit=5000, omp=True)
    assert abs(it - o["iterations"]) <= 2, (it, o["iterations"])
    assert nrm < 1e-6 and abs(nrm - o["rel_res_norm"]) <= 1e-3 * o["rel_res_norm"]


@pytest.mark.gpu
def test_driver_reads_a_shuffled_symmetric_file_with_the_parallel_parser(tmp_path):
    """the parallel parser inside the GPU suite on a file the serial one would also take: symmetric storage (lower triangle,
    shuffled), 359 k stored entries -> 599 k after expansion; the driver's answer must be the one the oracle gives for the
    expanded operator."""
    nx, ny = 400, 300
    rp, ci, v = oracle.stencil_csr(5, nx, ny)
    n = rp.size - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    low = ci <= rows
    r, c, vv = rows[low], ci[low], v[low]
    order = np.random.default_rng(5).permutation(r.size)
    mtx = tmp_path / "sym.mtx"
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n")
        f.write(f"{n} {n} {r.size}\n")
        f.write("".join(f"{r[k] + 1} {c[k] + 1} {vv[k]:.1f}\n" for k in order))
    env = dict(os.environ, MK_PARSE_TRACE="1", MK_PARSE_THREADS="6")
    env.pop("MK_PARSE_MIN_ENTRIES", None)
    out = subprocess.run([os.path.join(HOST, "solver"), "-solver", "51", "-mtx", str(mtx), "-tol", "1e-8", "-maxit", "3000", "-log", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Matrix Market parser: parallel" in out.stderr
    it = int(re.search(r"^Iterations = (\d+)$", out.stdout, re.M).group(1))
    o = oracle.pcg(rp, ci, v, np.ones(n), tol=1e-8, precond=1, maxit=3000)
    assert abs(it - o["iterations"]) <= 2, (it, o["iterations"])


def test_reference_call_sites_compile_against_the_host_header(tmp_path):
    """the four loader / conversion calls the reference driver and its dead alternative driver make (solvers.c:196-210,
    main.c:64: results ignored, third argument of the array loader an int) compile without a diagnostic against
    host/sparse_matrix.h -- the host API is the reference's, argument for argument"""
    src = tmp_path / "calls.c"
    src.write_text('#include "sparse_matrix.h"\n'
                   "void driver(const char *m, const char *b) {\n"
                   "  SparseMatrixCOO t, c; SparseMatrixCSR r; double *rhs; const int mtx_ext = 1;\n"
                   "  fast_load_from_mtx_file(m, &t);\n"
                   "  sort_coo_row(&t, &c);\n"
                   "  fast_load_from_array_file(b, &rhs, mtx_ext);\n"
                   "  convert_coo_to_csr(&c, &r, 1);\n"
                   "  (void)get_multiple_of_align_size(100);\n"
                   "}\n")
    out = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-Dalign_size=64", "-I", HOST, "-I", os.path.join(ROOT, "include"),
                          "-c", str(src), "-o", str(tmp_path / "calls.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
