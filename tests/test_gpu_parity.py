"""GPU parity tests: every call goes through the C ABI of libmi355_krylov.so and is checked
against the CPU oracle (oracle/) and the golden fixtures the reference produced.
Integer/byte results: bit-exact.  SpMV: 1e-12 relative.  CG: iteration count within +-2."""
import os

import numpy as np
import pytest

import iterative_solvers_amd as mk
import oracle
from conftest import GOLDEN, GOLDEN_CASES

pytestmark = pytest.mark.gpu

SPMV_RTOL = 1e-12   # north_star: "SpMV y matches the reference within 1e-12 relative in fp64"
CG_ITER_TOL = 2     # north_star: "within +-2 iterations"


def _same_bits(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


# ------------------------------------------------------------------ conversion: bit-exact
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_sort_coo_row_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sr, sc, sv = mk.sort_coo_row(int(g["nrows"]), int(g["ncols"]), g["coo_rows"], g["coo_cols"], g["coo_vals"])
    assert np.array_equal(sr, g["sorted_rows"])
    assert np.array_equal(sc, g["sorted_cols"])
    assert _same_bits(sv, g["sorted_vals"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_coo_to_csr_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rp, ci, v = mk.coo_to_csr(int(g["nrows"]), int(g["ncols"]), g["coo_rows"], g["coo_cols"], g["coo_vals"])
    assert np.array_equal(rp, g["csr_rowptr"])
    assert np.array_equal(ci, g["csr_col"])
    assert _same_bits(v, g["csr_val"])


@pytest.mark.parametrize("nrows,nnz,skew", [(1, 100, 0), (3, 50000, 0), (255, 10000, 0), (257, 100000, 0),
                                            (65536, 300000, 1), (65537, 300000, 0), (1 << 20, 2000000, 1),
                                            ((1 << 24) + 5, 3000000, 0),
                                            # entry counts at and around whole sort tiles (4096) and whole rounds of 8 tiles
                                            (300, 4096, 0), (300, 4097, 0), (300, 4095, 0), (70000, 65536, 0), (70000, 8 * 4096 + 1, 1),
                                            (70000, 9 * 4096, 0), (1 << 17, 24 * 4096 - 1, 1)])
def test_coo_to_csr_random_vs_oracle(nrows, nnz, skew):
    rng = np.random.default_rng(nrows + nnz)
    if skew:
        rows = np.minimum((rng.pareto(0.6, nnz)).astype(np.int64), nrows - 1).astype(np.int32)
    else:
        rows = rng.integers(0, nrows, nnz).astype(np.int32)
    cols = rng.integers(0, nrows, nnz).astype(np.int32)
    vals = rng.normal(size=nnz)
    rp, ci, v = mk.coo_to_csr(nrows, nrows, rows, cols, vals)
    orp, oci, ov = oracle.coo_to_csr(nrows, rows, cols, vals)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and _same_bits(v, ov)
    sr, sc, sv = mk.sort_coo_row(nrows, nrows, rows, cols, vals)
    osr, osc, osv = oracle.sort_coo_row(nrows, rows, cols, vals)
    assert np.array_equal(sr, osr) and np.array_equal(sc, osc) and _same_bits(sv, osv)


def test_coo_to_csr_degenerate_orders():
    """already sorted input (no sort pass runs), reverse order, every entry in one row, in the last row: bit-exact"""
    rng = np.random.default_rng(7)
    nrows, nnz = 5000, 20000
    cols = rng.integers(0, nrows, nnz).astype(np.int32)
    vals = rng.normal(size=nnz)
    for rows in (np.sort(rng.integers(0, nrows, nnz)).astype(np.int32), np.sort(rng.integers(0, nrows, nnz))[::-1].astype(np.int32).copy(),
                 np.full(nnz, 17, np.int32), np.full(nnz, nrows - 1, np.int32)):
        rp, ci, v = mk.coo_to_csr(nrows, nrows, rows, cols, vals)
        orp, oci, ov = oracle.coo_to_csr(nrows, rows, cols, vals)
        assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and _same_bits(v, ov)


def test_coo_rejects_bad_indices():
    with pytest.raises(mk.MkError):
        mk.coo_to_csr(4, 4, [0, 4], [0, 0], [1.0, 1.0])
    with pytest.raises(mk.MkError):
        mk.coo_to_csr(4, 4, [0, 1], [0, -1], [1.0, 1.0])


def test_rmat_generator_matches_oracle_and_converts():
    scale, ne = 12, 1 << 16
    d = mk.DeviceCOO.rmat(scale, ne, seed=42)
    nr, nc, r, c, v = d.download()
    orr, occ, ovv = oracle.rmat_coo(scale, ne, seed=42)
    assert nr == 1 << scale and np.array_equal(r, orr) and np.array_equal(c, occ) and _same_bits(v, ovv)
    A = mk.Matrix.from_device_coo(d)
    rp, ci, val = A.to_csr()
    orp, oci, ov = oracle.coo_to_csr(nr, orr, occ, ovv)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and _same_bits(val, ov)


# ------------------------------------------------------------------ generators
@pytest.mark.parametrize("kind,dims", [(5, (37, 19, 1)), (7, (13, 9, 11)), (27, (7, 8, 9)), (27, (1, 1, 5)), (7, (64, 64, 64))])
def test_stencil_generator(kind, dims):
    A = mk.Matrix.stencil(kind, *dims)
    rp, ci, v = A.to_csr()
    orp, oci, ov = oracle.stencil_csr(kind, *dims)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and _same_bits(v, ov)


def test_stencil_generator_row_block():
    orp, oci, ov = oracle.stencil_csr(7, 10, 11, 12)
    A = mk.Matrix.stencil(7, 10, 11, 12, row_begin=333, row_end=999)
    rp, ci, v = A.to_csr()
    assert np.array_equal(rp, orp[333:1000] - orp[333])
    assert np.array_equal(ci, oci[orp[333]:orp[999]]) and _same_bits(v, ov[orp[333]:orp[999]])


def test_banded_generator():
    A = mk.Matrix.banded(5000, 3)
    rp, ci, v = A.to_csr()
    orp, oci, ov = oracle.banded_csr(5000, 3)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and _same_bits(v, ov)


# ------------------------------------------------------------------ SpMV
def _check_spmv(A, rp, ci, v, n, ncols, rng, alpha=1.0, beta=0.0):
    xh = rng.normal(size=ncols)
    yh = rng.normal(size=n)
    x, y = mk.Vector(ncols, values=xh), mk.Vector(n, values=yh)
    A.spmv(x, y, alpha, beta)
    got = y.get()
    ref = oracle.csr_spmv(rp, ci, v, xh, alpha, beta, yh)
    # scale per row by sum |a_ij x_j| so that cancellation does not hide or fake an error
    mag = oracle.csr_spmv(rp, ci, np.abs(v), np.abs(xh)) * abs(alpha) + abs(beta) * np.abs(yh)
    err = np.abs(got - ref) / np.maximum(mag, 1e-300)
    assert err.max() <= SPMV_RTOL, err.max()
    assert np.linalg.norm(got - ref) <= SPMV_RTOL * np.linalg.norm(ref) + 1e-300


def test_spmv_known_answer():
    k = np.load(os.path.join(GOLDEN, "sblas_mv_kat.npz"))   # test/sblas_mv_rd.c:15-32
    A = mk.Matrix.from_csr(5, 7, k["iaptr"], k["iaind"], k["aval"])
    x, y = mk.Vector(7, values=k["x"]), mk.Vector(5, fill=0.0)
    A.spmv(x, y)
    assert np.allclose(y.get(), [3.5, 11.3, 66.9, 21.7, 77.0], rtol=1e-14)


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 32, 64])
def test_spmv_vector_all_group_widths(lanes):
    rng = np.random.default_rng(lanes)
    rp, ci, v = oracle.stencil_csr(27, 9, 10, 11)
    v = v * rng.uniform(0.5, 1.5, v.size)
    n = rp.size - 1
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    A.set_spmv_kernel(mk.SPMV_VECTOR, lanes)
    _check_spmv(A, rp, ci, v, n, n, rng)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=-1.0, beta=1.0)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=0.5, beta=-2.0)


@pytest.mark.parametrize("kind,dims", [(5, (33, 17, 1)), (7, (19, 11, 13)), (27, (9, 10, 11)), (7, (70, 70, 70))])
def test_spmv_stream_kernel_stencils(kind, dims):
    rng = np.random.default_rng(kind)
    rp, ci, v = oracle.stencil_csr(kind, *dims)
    v = v * rng.uniform(0.5, 1.5, v.size)
    n = rp.size - 1
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    assert A.info().kernel == mk.SPMV_STREAM          # regular rows -> LDS-staged kernel
    assert 1 <= A.info().index_patterns <= 27         # stencil rows share a few (column - row) patterns
    _check_spmv(A, rp, ci, v, n, n, rng)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=-1.0, beta=1.0)
    if kind != 27:
        # one lane per row sums left to right like the serial reference: same bits
        xh = rng.normal(size=n)
        x, y = mk.Vector(n, values=xh), mk.Vector(n)
        A.spmv(x, y)
        assert np.array_equal(y.get(), oracle.csr_spmv(rp, ci, v, xh))
    A.set_spmv_kernel(mk.SPMV_VECTOR)
    _check_spmv(A, rp, ci, v, n, n, rng)


@pytest.mark.parametrize("kind,dims", [(5, (33, 17, 1)), (7, (19, 11, 13)), (27, (9, 10, 11)), (7, (64, 64, 40))])
def test_spmv_pattern_kernel_constant_coefficients(kind, dims):
    """rows of a constant-coefficient stencil share offsets AND values: the pattern kernel reads 2 B of
    matrix data per row and must give the serial reference's bits (one lane per row, left to right)"""
    rng = np.random.default_rng(kind + 100)
    rp, ci, v = oracle.stencil_csr(kind, *dims)
    n = rp.size - 1
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    i = A.info()
    assert i.kernel == mk.SPMV_PATTERN and 1 <= i.value_patterns <= 27 and i.value_patterns == i.index_patterns
    xh = rng.normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    A.spmv(x, y)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    assert np.array_equal(y.get(), ref)
    yx, xx = A.spmv_dot(x, y)
    assert np.array_equal(y.get(), ref)
    assert abs(yx - ref @ xh) <= 1e-12 * np.abs(ref * xh).sum() and abs(xx - xh @ xh) <= 1e-12 * (xh @ xh)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=-1.0, beta=1.0)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=0.5, beta=-2.0)
    # the CSR arrays are still there: other kernels on the same matrix, and the download
    for k in (mk.SPMV_STREAM, mk.SPMV_VECTOR, mk.SPMV_SPLIT, mk.SPMV_PATTERN):
        A.set_spmv_kernel(k)
        _check_spmv(A, rp, ci, v, n, n, rng)
    rp2, ci2, v2 = A.to_csr()
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(v2.view(np.uint64), v.view(np.uint64))


def test_spmv_pattern_kernel_value_edge_cases():
    """-0.0 and +0.0 are different table entries (bit compare), rows with distinct values fall back to
    index-only compression, and the pattern kernel is refused for them"""
    n = 4096
    rp = np.arange(0, 2 * n + 1, 2, dtype=np.int32)
    ci = np.stack([np.arange(n), (np.arange(n) + 1) % n]).T.copy()
    ci[-1] = [n - 1, n - 1]
    ci = ci.reshape(-1).astype(np.int32)
    v = np.tile([2.0, 0.0], n)
    v[1::4] = -0.0                                        # every other row stores -0.0
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    i = A.info()
    assert i.kernel == mk.SPMV_PATTERN and i.value_patterns == 3      # interior rows with +0.0, with -0.0, and the last row
    xh = -np.ones(n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    A.spmv(x, y)
    assert np.array_equal(y.get().view(np.uint64), oracle.csr_spmv(rp, ci, v, xh).view(np.uint64))
    v3 = np.random.default_rng(0).normal(size=2 * n)
    B = mk.Matrix.from_csr(n, n, rp, ci, v3)
    assert B.info().kernel == mk.SPMV_STREAM and B.info().value_patterns == 0 and B.info().index_patterns == 2
    with pytest.raises(mk.MkError):
        B.set_spmv_kernel(mk.SPMV_PATTERN)


@pytest.mark.parametrize("kind,dims", [(7, (64, 64, 40)), (7, (30, 22, 10)), (27, (16, 18, 9)), (27, (40, 40, 6)),
                                       (5, (128, 96, 1)), (7, (512, 4, 5)), (7, (24, 16, 33)),
                                       (7, (640, 4, 5)), (7, (1024, 3, 4)), (27, (700, 3, 3))])   # wide rows: 2048-row chunks
def test_spmv_sweep_kernel(kind, dims):
    """grid operators whose rows are sub-lists of one pattern: plane-by-plane sweep with the x window in
    LDS.  Same (column, value) pairs in the same order as the CSR arrays -> the serial reference's bits."""
    rng = np.random.default_rng(kind + 200)
    rp, ci, v = oracle.stencil_csr(kind, *dims)
    n = rp.size - 1
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    A.set_spmv_kernel(mk.SPMV_SWEEP)
    assert A.info().kernel == mk.SPMV_SWEEP
    xh = rng.normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    A.spmv(x, y)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    assert np.array_equal(y.get(), ref)
    yx, xx = A.spmv_dot(x, y)
    assert np.array_equal(y.get(), ref)
    assert abs(yx - ref @ xh) <= 1e-12 * np.abs(ref * xh).sum() and abs(xx - xh @ xh) <= 1e-12 * (xh @ xh)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=-1.0, beta=1.0)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=0.5, beta=-2.0)
    # whole solves through the sweep kernel agree with the pattern kernel's
    res = {}
    for k in (mk.SPMV_SWEEP, mk.SPMV_PATTERN):
        A.set_spmv_kernel(k)
        b, x0 = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
        S = mk.PCG(max_iter=300, tol=1e-9, precond=mk.PRECOND_JACOBI)
        S.setup(A, b, x0)
        r = S.solve(A, b, x0)
        J = mk.Jacobi(max_iter=40, tol=1e-30, check_every=7)
        xj = mk.Vector(n, fill=0.0)
        J.setup(A, b, xj)
        rj = J.solve(A, b, xj)
        res[k] = (r, x0.get(), rj, xj.get())
    assert abs(res[mk.SPMV_SWEEP][0]["iterations"] - res[mk.SPMV_PATTERN][0]["iterations"]) <= 1   # fused step: other dot order
    assert np.allclose(res[mk.SPMV_SWEEP][1], res[mk.SPMV_PATTERN][1], rtol=1e-7, atol=1e-10)
    assert np.array_equal(res[mk.SPMV_SWEEP][3], res[mk.SPMV_PATTERN][3])      # Jacobi sweeps: same bits per row


@pytest.mark.parametrize("n,hb", [(1, 1), (2, 1), (7, 3), (511, 2), (513, 3), (1025, 1), (100001, 3), (4099, 15), (300003, 3)])   # 300003: planes of 64 chunks, ragged last plane
def test_spmv_pattern_and_sweep_ragged_sizes(n, hb):
    """odd row counts (a lane's second row missing), fewer rows than one chunk, rows shorter than the band at
    both ends: pattern and sweep kernels against the serial reference, bit for bit; then whole solves"""
    rp, ci, v = oracle.banded_csr(n, hb)
    rng = np.random.default_rng(n)
    xh = rng.normal(size=n)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    kernels = [mk.SPMV_PATTERN, mk.SPMV_SWEEP] if A.info().value_patterns > 0 else []
    assert kernels or n < 40                     # tiny matrices may have one pattern per row: no table
    for k in kernels:
        try:
            A.set_spmv_kernel(k)
        except mk.MkError:
            assert k == mk.SPMV_SWEEP and n < 2 * hb + 1      # no interior row: no common base pattern
            continue
        x, y = mk.Vector(n, values=xh), mk.Vector(n, fill=-3.0)
        A.spmv(x, y)
        assert np.array_equal(y.get(), ref)
        yx, xx = A.spmv_dot(x, y)
        assert np.array_equal(y.get(), ref) and abs(yx - ref @ xh) <= 1e-12 * np.abs(ref * xh).sum()
        _check_spmv(A, rp, ci, v, n, n, rng, alpha=-0.5, beta=2.0)
        o = oracle.pcg(rp, ci, v, np.ones(n), tol=1e-10, precond=1, maxit=500)
        b, x0 = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
        S = mk.PCG(max_iter=500, tol=1e-10, precond=mk.PRECOND_JACOBI)
        S.setup(A, b, x0)
        r = S.solve(A, b, x0)
        assert r["converged"] == o["converged"] == 1 and abs(r["iterations"] - o["iterations"]) <= CG_ITER_TOL
        assert np.linalg.norm(x0.get() - o["x"]) <= 1e-7 * np.linalg.norm(o["x"])
        oj = oracle.jacobi(rp, ci, v, np.ones(n), tol=1e-30, maxit=25, check_every=5)
        J = mk.Jacobi(max_iter=25, tol=1e-30, check_every=5)
        xj = mk.Vector(n, fill=0.0)
        J.setup(A, b, xj)
        J.solve(A, b, xj)
        assert np.allclose(xj.get(), oj["x"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("seed,P,K,nbase,full_frac", [(1, 640, 9, 9, 0.6), (2, 1024, 5, 14, 0.6), (3, 96, 40, 6, 0.6), (4, 2048, 12, 12, 0.6),
                                                       # long patterns, nearly all rows complete: the lanes of a chunk are dealt
                                                       # boundary pairs first, complete waves run select-free batches, the last
                                                       # 1-4 entries take the half batch (27 = 3 x 8 + 3, 17 = 2 x 8 + 1, 20, 32)
                                                       (5, 512, 12, 27, 0.995), (6, 768, 9, 20, 0.99), (7, 256, 30, 32, 0.98),
                                                       (8, 640, 8, 17, 0.999), (9, 1024, 6, 27, 1.0)])
def test_spmv_sweep_kernel_random_patterns(seed, P, K, nbase, full_frac):
    """not a grid: a random base pattern with offsets a*P + b (a in {-1,0,1}, unsymmetric, unsorted), every row a
    random sub-list of it chosen among ~20 masks, row-dependent sub-lists at the matrix ends.  Pattern and sweep
    kernels must reproduce the serial reference bit for bit (same pairs, same order)."""
    rng = np.random.default_rng(seed)
    n = P * K
    bmax = min(P // 2 - 1, 300)
    offs = []
    while len(offs) < nbase:
        o = int(rng.integers(-1, 2)) * P + int(rng.integers(-bmax, bmax + 1))
        if o not in offs:
            offs.append(o)
    if 0 not in offs:
        offs[int(rng.integers(0, nbase))] = 0            # keep a diagonal (Jacobi modes)
    offs = np.array(offs)
    vals = rng.normal(size=nbase)
    vals[offs == 0] = 4.0 + nbase
    masks = [np.ones(nbase, bool)] + [rng.random(nbase) < 0.7 for _ in range(19)]
    for m in masks:
        m[offs == 0] = True
    which = rng.integers(0, len(masks), n)
    which[rng.random(n) < full_frac] = 0                  # most rows carry the full pattern
    rows_c, rows_v = [], []
    for i in range(n):
        keep = masks[which[i]] & (i + offs >= 0) & (i + offs < n)
        rows_c.append(i + offs[keep])
        rows_v.append(vals[keep])
    lens = np.array([c.size for c in rows_c])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate(rows_c).astype(np.int32)
    v = np.concatenate(rows_v)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    assert A.info().value_patterns > 0
    xh = rng.normal(size=n)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    for k in (mk.SPMV_PATTERN, mk.SPMV_SWEEP):
        A.set_spmv_kernel(k)
        x, y = mk.Vector(n, values=xh), mk.Vector(n, fill=9.0)
        A.spmv(x, y)
        assert np.array_equal(y.get(), ref), k
        _check_spmv(A, rp, ci, v, n, n, rng, alpha=1.25, beta=-0.5)
        oj = oracle.jacobi(rp, ci, v, np.ones(n), tol=1e-30, maxit=8, check_every=4)
        J = mk.Jacobi(max_iter=8, tol=1e-30, check_every=4)
        b, xj = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
        J.setup(A, b, xj)
        J.solve(A, b, xj)
        assert np.allclose(xj.get(), oj["x"], rtol=1e-12, atol=0), k


def _vsweep_checks(A, rp, ci, v, n, rng, jacobi=True):
    """MK_SPMV_VSWEEP against the serial reference: product, product + dots, alpha/beta forms, Jacobi sweeps"""
    A.set_spmv_kernel(mk.SPMV_VSWEEP)
    assert A.info().kernel == mk.SPMV_VSWEEP and A.info().value_patterns == 0 and A.info().index_patterns > 0
    xh = rng.normal(size=n)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    x, y = mk.Vector(n, values=xh), mk.Vector(n, fill=7.0)
    A.spmv(x, y)
    assert np.array_equal(y.get(), ref)                    # same (value, x) pairs in the same order: the reference's bits
    y.fill(-1.0)
    yx, xx = A.spmv_dot(x, y)
    assert np.array_equal(y.get(), ref)
    assert abs(yx - ref @ xh) <= 1e-12 * np.abs(ref * xh).sum() and abs(xx - xh @ xh) <= 1e-12 * (xh @ xh)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=-1.0, beta=1.0)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=0.5, beta=-2.0)
    if jacobi:
        oj = oracle.jacobi(rp, ci, v, np.ones(n), tol=1e-30, maxit=9, check_every=4)
        J = mk.Jacobi(max_iter=9, tol=1e-30, check_every=4)
        b, xj = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
        J.setup(A, b, xj)
        J.solve(A, b, xj)
        assert np.allclose(xj.get(), oj["x"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("kind,dims", [(7, (64, 64, 40)), (7, (30, 22, 10)), (5, (128, 96, 1)), (7, (512, 4, 5)), (7, (24, 16, 33)),
                                       (7, (640, 4, 5)), (7, (1024, 3, 9)), (5, (1000, 700, 1)), (7, (6, 6, 100)),
                                       # rows of 27 entries: one row per lane, 512-row chunks
                                       (27, (12, 12, 12)), (27, (64, 40, 20)), (27, (256, 6, 5)), (27, (500, 4, 7)), (27, (34, 30, 31)), (27, (620, 4, 5))])
def test_spmv_vsweep_kernel_variable_coefficients(kind, dims):
    """grid operators with per-row coefficients (D A D): column offsets from the pattern table, values streamed through LDS,
    x through the window ring (vsweep.hip)"""
    rng = np.random.default_rng(kind + 300)
    rp, ci, v = oracle.stencil_csr(kind, *dims, scale_seed=11)
    n = rp.size - 1
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    _vsweep_checks(A, rp, ci, v, n, rng)
    # a whole CG solve through it (K1 on the vsweep kernel, K2/K3 the vector kernels) against the oracle
    bh = rng.normal(size=n)
    o = oracle.pcg(rp, ci, v, bh, tol=1e-9, precond=1, maxit=400, want_norms=True)     # the large 2D grid stops at 400 unconverged
    b, x0 = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=400, tol=1e-9, precond=mk.PRECOND_JACOBI, logging=1)
    S.setup(A, b, x0)
    r = S.solve(A, b, x0)
    assert r["converged"] == o["converged"] and abs(r["iterations"] - o["iterations"]) <= CG_ITER_TOL
    assert np.linalg.norm(x0.get() - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
    hist = S.residual_history(4000)
    m = min(hist.size, o["rel_norms"].size, 30)
    assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)


@pytest.mark.parametrize("kind,dims", [(7, (64, 64, 40)), (7, (30, 22, 10)), (5, (128, 96, 1)), (7, (640, 4, 6)), (7, (1024, 3, 9)), (7, (6, 6, 100)),
                                       (27, (20, 18, 16)), (27, (260, 4, 6))])
@pytest.mark.parametrize("precond", [0, 1])
@pytest.mark.parametrize("rel_change", [0, 1])
def test_pcg_vsweep_fused_k3_matches_oracle(kind, dims, precond, rel_change):
    """CG on variable-coefficient grids: from the second iteration on K3 rides in the window fill of the
    vsweep product (p' = z + beta p formed while the window is loaded, own rows to the other p buffer, with the Jacobi
    preconditioner 1/diag as a third window stream), K2 unchanged.  Same recurrences and stopping rule as the three-kernel
    loop: iteration counts, residual history, exact-k solutions and the zero right-hand side against orc_pcg (parity
    unpinned: hypre itself is not available)."""
    rp, ci, v = oracle.stencil_csr(kind, *dims, scale_seed=23)
    n = rp.size - 1
    tol = 1e-9
    bh = np.random.default_rng(n).normal(size=n)
    o = oracle.pcg(rp, ci, v, bh, tol=tol, precond=precond, maxit=600, want_norms=True, rel_change=rel_change)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    A.set_spmv_kernel(mk.SPMV_VSWEEP)
    for check_every in (1, 7):
        b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
        S = mk.PCG(max_iter=600, tol=tol, two_norm=1, rel_change=rel_change, precond=precond, logging=1, check_every=check_every)
        S.setup(A, b, x)
        res = S.solve(A, b, x)
        assert res["converged"] == o["converged"]
        assert abs(res["iterations"] - o["iterations"]) <= CG_ITER_TOL, (res, o["iterations"])
        xs = x.get()
        assert np.linalg.norm(xs - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
        hist = S.residual_history(2000)
        assert hist.size == res["iterations"] + 1
        m = min(hist.size, o["rel_norms"].size, 40)
        assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)
        if res["converged"]:
            assert res["rel_res_norm"] < tol
            assert np.linalg.norm(bh - oracle.csr_spmv(rp, ci, v, xs)) / np.linalg.norm(bh) < tol * 1.5
    for k in (1, 2, 3, 6):      # exactly k iterations; p ping-pongs between two buffers from the second one on
        b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
        S = mk.PCG(max_iter=k, tol=1e-30, rel_change=rel_change, precond=precond)
        S.setup(A, b, x)
        res = S.solve(A, b, x)
        ok = oracle.pcg(rp, ci, v, bh, tol=1e-30, precond=precond, maxit=k, rel_change=rel_change)
        assert res["iterations"] == k and res["converged"] == 0
        assert np.isclose(res["rel_res_norm"], ok["rel_res_norm"], rtol=1e-9)
        assert np.allclose(x.get(), ok["x"], rtol=1e-9, atol=1e-13)
        if k == 6:              # a second solve on the same solver object starts from the previous x (solvers.c:753-754)
            res2 = S.solve(A, b, x)
            ok2 = oracle.pcg(rp, ci, v, bh, x0=ok["x"], tol=1e-30, precond=precond, maxit=k, rel_change=rel_change)
            assert res2["iterations"] == k and np.allclose(x.get(), ok2["x"], rtol=1e-8, atol=1e-12)
    b0, x0 = mk.Vector(n, fill=0.0), mk.Vector(n, fill=3.0)
    S = mk.PCG(max_iter=10, tol=1e-9, precond=precond, rel_change=rel_change)
    S.setup(A, b0, x0)
    assert S.solve(A, b0, x0)["iterations"] == 0 and np.all(x0.get() == 0.0)


def test_spmv_vsweep_device_generated_matches_oracle_generator():
    """mk_matrix_generate_stencil_scaled and the oracle's restatement of it give the same bits; 192 x 160 x 72 is large
    enough for the selector to take the vsweep kernel by itself"""
    dims = (192, 160, 72)
    A = mk.Matrix.stencil(7, *dims, scale_seed=5)
    rp, ci, v = oracle.stencil_csr(7, *dims, scale_seed=5)
    grp, gci, gv = A.to_csr()
    assert np.array_equal(grp, rp) and np.array_equal(gci, ci) and np.array_equal(gv.view(np.uint64), v.view(np.uint64))
    assert A.info().kernel == mk.SPMV_VSWEEP
    n = rp.size - 1
    _vsweep_checks(A, rp, ci, v, n, np.random.default_rng(1), jacobi=False)
    # rows [lo, hi) of the same operator generated as a block carry the same values (global row in the hash)
    lo, hi = 1000, 50000
    B = mk.Matrix.stencil(7, *dims, row_begin=lo, row_end=hi, scale_seed=5)
    brp, bci, bv = B.to_csr()
    assert np.array_equal(bci, ci[rp[lo]:rp[hi]]) and np.array_equal(bv.view(np.uint64), v[rp[lo]:rp[hi]].view(np.uint64))


@pytest.mark.parametrize("n,hb", [(7, 3), (511, 2), (513, 3), (1025, 1), (100001, 3), (70000, 2), (262921, 3)])
def test_spmv_vsweep_ragged_banded(n, hb):
    """banded matrices with random values (the contiguous sweep mode): odd row counts, fewer rows than a chunk, short rows
    at both ends"""
    rp, ci, v = oracle.banded_csr(n, hb)
    rng = np.random.default_rng(n + hb)
    rows = np.repeat(np.arange(n), np.diff(rp))
    v = np.where(ci == rows, v + rng.uniform(0, 1, v.size), v * rng.uniform(0.2, 1.0, v.size))   # stays diagonally dominant
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    if A.info().value_patterns > 0 or A.info().index_patterns == 0:
        assert n < 40          # a handful of rows: one (offsets, values) table entry per row, the pattern kernel serves it
        return
    _vsweep_checks(A, rp, ci, v, n, rng)


@pytest.mark.parametrize("seed,P,K,nbase", [(1, 640, 9, 8), (2, 1024, 5, 7), (3, 96, 40, 6), (4, 2048, 12, 5), (5, 512, 21, 3),
                                            (6, 640, 9, 12), (7, 96, 40, 20), (8, 512, 21, 28), (9, 1024, 5, 9), (10, 2048, 7, 17)])
def test_spmv_vsweep_random_patterns(seed, P, K, nbase):
    """not a grid: a random base pattern with offsets a*P + b (unsymmetric, unsorted), every row a random sub-list of it,
    every entry its own value; the kernel must reproduce the serial reference bit for bit"""
    rng = np.random.default_rng(seed + 50)
    n = P * K
    bmax = min(P // 2 - 1, 300)
    offs = []
    while len(offs) < nbase:
        o = int(rng.integers(-1, 2)) * P + int(rng.integers(-bmax, bmax + 1))
        if o not in offs:
            offs.append(o)
    if 0 not in offs:
        offs[int(rng.integers(0, nbase))] = 0
    offs = np.array(offs)
    masks = [np.ones(nbase, bool)] + [rng.random(nbase) < 0.7 for _ in range(19)]
    for m in masks:
        m[offs == 0] = True
    which = rng.integers(0, len(masks), n)
    which[rng.random(n) < 0.6] = 0
    rows_c = []
    for i in range(n):
        keep = masks[which[i]] & (i + offs >= 0) & (i + offs < n)
        rows_c.append(i + offs[keep])
    lens = np.array([c.size for c in rows_c])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate(rows_c).astype(np.int32)
    v = rng.normal(size=ci.size)
    rows = np.repeat(np.arange(n), lens)
    v[ci == rows] = 4.0 + nbase + rng.uniform(0, 1, int((ci == rows).sum()))
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    _vsweep_checks(A, rp, ci, v, n, rng)


def test_spmv_vsweep_eligibility():
    rp, ci, v = oracle.stencil_csr(27, 800, 4, 4, scale_seed=2)      # 27 entries per row on 800-wide planes: stage + window exceed the LDS
    n = rp.size - 1
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    assert A.info().kernel == mk.SPMV_STREAM
    with pytest.raises(mk.MkError, match="at most 28 entries"):
        A.set_spmv_kernel(mk.SPMV_VSWEEP)
    rp, ci, v = oracle.stencil_csr(7, 16, 16, 16)                      # constant coefficients: the table holds the values
    B = mk.Matrix.from_csr(rp.size - 1, rp.size - 1, rp, ci, v)
    with pytest.raises(mk.MkError):
        B.set_spmv_kernel(mk.SPMV_VSWEEP)
    rp, ci, v = oracle.stencil_csr(7, 16, 16, 16, scale_seed=1)
    Cm = mk.Matrix.from_csr(rp.size - 1, rp.size - 1, rp, ci, v)
    with pytest.raises(mk.MkError):
        Cm.set_spmv_kernel(mk.SPMV_SWEEP)                              # no shared values: not the constant-coefficient sweep


def test_spmv_sweep_kernel_eligibility():
    rp, ci, v = oracle.stencil_csr(7, 31, 29, 3)           # plane of 899 rows: odd stride, a lane's two rows would straddle planes
    n = rp.size - 1
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    assert A.info().kernel == mk.SPMV_PATTERN
    with pytest.raises(mk.MkError):
        A.set_spmv_kernel(mk.SPMV_SWEEP)
    B = mk.Matrix.banded(1 << 22, hb=3)                      # contiguous mode, enough chunks to be the default
    assert B.info().kernel == mk.SPMV_SWEEP
    rpb, cib, vb = B.to_csr()
    xh = np.random.default_rng(5).normal(size=1 << 22)
    x, y = mk.Vector(1 << 22, values=xh), mk.Vector(1 << 22)
    B.spmv(x, y)
    assert np.array_equal(y.get(), oracle.csr_spmv(rpb, cib, vb, xh))


def test_spmv_pattern_kernel_wide_addressing():
    """matrices of 2^29 rows or more use plain 64-bit addressing in the pattern kernel; that build is
    forced here on a small operator (MK_PATTERN_WIDE is read once per process, hence the child)"""
    import subprocess, sys
    code = (
        "import numpy as np, iterative_solvers_amd as mk\n"
        "from oracle import stencil_csr, csr_spmv\n"
        "rp, ci, v = stencil_csr(7, 40, 24, 12)\n"
        "n = rp.size - 1\n"
        "A = mk.Matrix.from_csr(n, n, rp, ci, v)\n"
        "assert A.info().kernel == mk.SPMV_PATTERN\n"
        "xh = np.random.default_rng(1).normal(size=n)\n"
        "x, y = mk.Vector(n, values=xh), mk.Vector(n)\n"
        "A.spmv(x, y)\n"
        "assert np.array_equal(y.get(), csr_spmv(rp, ci, v, xh))\n"
        "yx, xx = A.spmv_dot(x, y)\n"
        "assert np.array_equal(y.get(), csr_spmv(rp, ci, v, xh))\n"
        "print('wide ok')\n")
    env = dict(os.environ, MK_PATTERN_WIDE="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=110)
    assert o.returncode == 0 and "wide ok" in o.stdout, o.stderr[-2000:]


@pytest.mark.parametrize("no_value_compression", ["0", "1"])
def test_nontemporal_instantiations_match_oracle(no_value_compression):
    """above 10 M rows the single-use streams are nontemporal: other instantiations of the sweep, stream and vector
    kernels (the ones bench.py runs).  MK_NT=1 forces them on small operators; the switch is read once per
    process, hence the child.  With value compression off the same solves go through the stream kernel + K2/K3."""
    import subprocess, sys
    code = (
        "import numpy as np, iterative_solvers_amd as mk, oracle\n"
        "for kind, dims in ((7, (64, 64, 40)), (27, (40, 40, 6)), (7, (640, 4, 6))):\n"
        "    rp, ci, v = oracle.stencil_csr(kind, *dims)\n"
        "    n = rp.size - 1\n"
        "    A = mk.Matrix.from_csr(n, n, rp, ci, v)\n"
        "    if A.info().value_patterns: A.set_spmv_kernel(mk.SPMV_SWEEP)\n"
        "    xh = np.random.default_rng(1).normal(size=n)\n"
        "    x, y = mk.Vector(n, values=xh), mk.Vector(n)\n"
        "    A.spmv(x, y)\n"
        "    ref = oracle.csr_spmv(rp, ci, v, xh)\n"
        "    mag = oracle.csr_spmv(rp, ci, np.abs(v), np.abs(xh))\n"
        "    same = (lambda got: np.array_equal(got, ref)) if A.info().value_patterns else (lambda got: (np.abs(got - ref) / mag).max() <= 1e-12)\n"
        "    assert same(y.get())\n"
        "    yx, xx = A.spmv_dot(x, y)\n"
        "    assert same(y.get()) and abs(yx - ref @ xh) <= 1e-12 * np.abs(ref * xh).sum()\n"
        "    for precond in (0, 1):\n"
        "        o = oracle.pcg(rp, ci, v, np.ones(n), tol=1e-9, precond=precond, maxit=1000, want_norms=True, rel_change=1)\n"
        "        b, x0 = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)\n"
        "        S = mk.PCG(max_iter=1000, tol=1e-9, two_norm=1, rel_change=1, precond=precond, logging=1, check_every=5)\n"
        "        S.setup(A, b, x0)\n"
        "        r = S.solve(A, b, x0)\n"
        "        assert r['converged'] == 1 and abs(r['iterations'] - o['iterations']) <= 2, (r, o['iterations'])\n"
        "        assert np.linalg.norm(x0.get() - o['x']) <= 1e-6 * np.linalg.norm(o['x'])\n"
        "        h = S.residual_history(2000)\n"
        "        m = min(h.size, o['rel_norms'].size, 40)\n"
        "        assert np.allclose(h[:m], o['rel_norms'][:m], rtol=1e-8)\n"
        "    J = mk.Jacobi(max_iter=30, tol=1e-30, check_every=7)\n"
        "    xj = mk.Vector(n, fill=0.0)\n"
        "    b = mk.Vector(n, fill=1.0)\n"
        "    J.setup(A, b, xj)\n"
        "    J.solve(A, b, xj)\n"
        "    oj = oracle.jacobi(rp, ci, v, np.ones(n), maxit=30, tol=1e-30)\n"
        "    assert np.allclose(xj.get(), oj['x'], rtol=1e-12, atol=1e-14)\n"
        "    for cls, fn in ((mk.BiCGSTAB, oracle.bicgstab), (mk.CGNR, oracle.cgnr)):\n"
        "        b, x0 = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)\n"
        "        S = cls(max_iter=4, tol=1e-30)\n"
        "        S.setup(A, b, x0)\n"
        "        S.solve(A, b, x0)\n"
        "        o = fn(rp, ci, v, np.ones(n), tol=1e-30, maxit=4)\n"
        "        assert np.linalg.norm(x0.get() - o['x']) <= 1e-9 * np.linalg.norm(o['x'])\n"
        "r, c, v = oracle.rmat_coo(13, 1 << 18, seed=3)\n"
        "n = 1 << 13\n"
        "rp, ci, val = oracle.coo_to_csr(n, r, c, v)\n"
        "A = mk.Matrix.from_csr(n, n, rp, ci, val)\n"
        "assert A.info().kernel == mk.SPMV_SPLIT\n"
        "xh = np.random.default_rng(1).normal(size=n)\n"
        "x, y = mk.Vector(n, values=xh), mk.Vector(n)\n"
        "A.spmv(x, y)\n"
        "ref = oracle.csr_spmv(rp, ci, val, xh)\n"
        "mag = oracle.csr_spmv(rp, ci, np.abs(val), np.abs(xh))\n"
        "assert (np.abs(y.get() - ref) / np.maximum(mag, 1e-300)).max() <= 1e-12\n"
        "print('nt ok')\n")
    env = dict(os.environ, MK_NT="1", MK_NO_VALUE_COMPRESSION=no_value_compression,
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=110)
    assert o.returncode == 0 and "nt ok" in o.stdout, (o.stdout[-500:], o.stderr[-2000:])


def test_spmv_index_compression_is_lossless_and_optional():
    """random short rows have ~n distinct patterns: the table would not fit, so the plain stream
    kernel runs; a block-structured matrix with a few hundred patterns is compressed; both must
    agree bit for bit with the uncompressed product (same pairs, same order)"""
    rng = np.random.default_rng(4)
    n = 6000
    lens = rng.integers(1, 9, n)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = rng.integers(0, n, rp[-1]).astype(np.int32)
    v = rng.normal(size=rp[-1])
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    assert A.info().kernel == mk.SPMV_STREAM and A.info().index_patterns == 0
    _check_spmv(A, rp, ci, v, n, n, rng)
    # 300 patterns: row i uses offsets of pattern i % 300 (wrapped into range)
    pats = [np.sort(rng.choice(np.arange(-40, 41), size=int(rng.integers(1, 9)), replace=False)) for _ in range(300)]
    rows_c, lens2 = [], []
    for i in range(n):
        c = i + pats[i % 300]
        c = c[(c >= 0) & (c < n)]
        rows_c.append(c)
        lens2.append(c.size)
    rp2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int32)
    ci2 = np.concatenate(rows_c).astype(np.int32)
    v2 = rng.normal(size=rp2[-1])
    B = mk.Matrix.from_csr(n, n, rp2, ci2, v2)
    assert 300 <= B.info().index_patterns < 1200
    xh = rng.normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    B.spmv(x, y)
    assert np.array_equal(y.get(), oracle.csr_spmv(rp2, ci2, v2, xh))    # one lane per row: reference order
    _check_spmv(B, rp2, ci2, v2, n, n, rng, alpha=0.7, beta=-1.3)


def test_spmv_stream_ragged_rows():
    # rows of very different lengths incl. empty ones, chunk spans not 16-byte aligned
    rng = np.random.default_rng(9)
    n = 3001
    lens = rng.integers(0, 12, n)
    lens[rng.random(n) < 0.2] = 0
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = rng.integers(0, n, rp[-1]).astype(np.int32)
    v = rng.normal(size=rp[-1])
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    for lanes_kernel in (mk.SPMV_STREAM, mk.SPMV_VECTOR):
        A.set_spmv_kernel(lanes_kernel)
        _check_spmv(A, rp, ci, v, n, n, rng, alpha=1.5, beta=0.25)

def _hex3_csr(nx, ny, nz):
    """3 unknowns per node of an nx x ny x nz grid, 27 node neighbours x 3x3 blocks: up to 81 entries per row"""
    nn = nx * ny * nz
    idx = np.arange(nn)
    ix, iy, iz = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    rr, cc = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = (ix + dx >= 0) & (ix + dx < nx) & (iy + dy >= 0) & (iy + dy < ny) & (iz + dz >= 0) & (iz + dz < nz)
                rr.append(idx[ok])
                cc.append(idx[ok] + dx + dy * nx + dz * nx * ny)
    nrow, ncol = np.concatenate(rr), np.concatenate(cc)
    rows = (nrow[:, None, None] * 3 + np.arange(3)[None, :, None] + np.zeros((1, 1, 3), np.int64)).reshape(-1)
    cols = (ncol[:, None, None] * 3 + np.arange(3)[None, None, :] + np.zeros((1, 3, 1), np.int64)).reshape(-1)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=3 * nn))]).astype(np.int32)
    return 3 * nn, rp, cols.astype(np.int32), rows


def test_spmv_stream_long_rows():
    """rows of 24..95 entries (several lanes per row, more than 8 entries per lane: the batched tail of the
    row pass), with and without an index-pattern table, and one Jacobi solve through the same kernel"""
    rng = np.random.default_rng(21)
    # a) block-structured: 81 entries per interior row, patterns shared -> column indices from the table
    n, rp, ci, rows = _hex3_csr(11, 9, 7)
    v = rng.normal(size=ci.size)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    i = A.info()
    assert i.kernel == mk.SPMV_STREAM and i.max_row_len == 81 and 81 <= i.index_patterns <= 2048
    fullest = np.add.reduceat(np.diff(rp), np.arange(0, n, 32)).max()
    assert i.stream_lanes == 8 and i.stream_chunk == (fullest + 3) // 4 * 4   # lanes from the mean row length, chunk = 32 rows
    _check_spmv(A, rp, ci, v, n, n, rng)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=-0.3, beta=2.0)
    # b) irregular: 60..95 random columns per row, no two rows alike -> col[] is read
    n2 = 5003
    lens = rng.integers(60, 96, n2)
    lens[::17] = 95
    rp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows2 = np.repeat(np.arange(n2), lens)
    ci2 = rng.integers(0, n2, rp2[-1])
    ci2 = ci2[np.lexsort((ci2, rows2))].astype(np.int32)
    v2 = rng.normal(size=rp2[-1])
    B = mk.Matrix.from_csr(n2, n2, rp2, ci2, v2)
    assert B.info().kernel == mk.SPMV_STREAM and B.info().index_patterns == 0
    lens32 = np.add.reduceat(lens, np.arange(0, n2, 32))
    assert B.info().stream_lanes == 8 and B.info().stream_chunk == (lens32.max() + 3) // 4 * 4   # the fullest run of 32 rows, not 32 x 95
    _check_spmv(B, rp2, ci2, v2, n2, n2, rng, alpha=1.25, beta=-0.5)
    yx, xx = None, None
    xh = rng.normal(size=n2)
    x, y = mk.Vector(n2, values=xh), mk.Vector(n2)
    yx, xx = B.spmv_dot(x, y)
    ref = oracle.csr_spmv(rp2, ci2, v2, xh)
    assert np.isclose(yx, ref @ xh, rtol=1e-11, atol=1e-9) and np.isclose(xx, xh @ xh, rtol=1e-12)
    # b2) node blocks: three rows per node with one (random) column list, no two nodes alike -> the list is staged once
    nodes = 1700
    nb = [np.unique(np.concatenate([[i], rng.integers(max(0, i - 40), min(nodes, i + 41), int(rng.integers(8, 30)))])) for i in range(nodes)]
    nb[5] = nb[4].copy()                # two neighbouring nodes with the same list: six rows share it
    nb[7] = np.array([7])               # a node coupled to itself only
    lists = [(np.repeat(q * 3, 3) + np.tile(np.arange(3), q.size)).astype(np.int32) for q in nb]
    ci3 = np.concatenate([np.tile(l, 3) for l in lists]).astype(np.int32)
    lens3 = np.repeat([l.size for l in lists], 3)
    # rows 3 i + 1 of every 11th node drop their last entry: same length class broken, list not shared with its neighbours
    n3 = 3 * nodes
    rp3 = np.concatenate([[0], np.cumsum(lens3)]).astype(np.int32)
    keep = np.ones(ci3.size, bool)
    for i in range(0, nodes, 11):
        if lists[i].size > 3:
            keep[rp3[3 * i + 2] - 1] = False      # last entry of row 3 i + 1
            lens3[3 * i + 1] -= 1
    ci3 = ci3[keep]
    rp3 = np.concatenate([[0], np.cumsum(lens3)]).astype(np.int32)
    v3 = rng.normal(size=ci3.size)
    C3 = mk.Matrix.from_csr(n3, n3, rp3, ci3, v3)
    i3 = C3.info()
    assert i3.kernel == mk.SPMV_STREAM and i3.index_patterns == 0
    same = np.zeros(n3, bool)
    for r in range(1, n3):
        a, b_ = ci3[rp3[r - 1]:rp3[r]], ci3[rp3[r]:rp3[r + 1]]
        same[r] = a.size == b_.size and b_.size > 0 and np.array_equal(a, b_)
    assert i3.shared_columns == int(lens3[same].sum()) and i3.shared_columns > 0.55 * ci3.size
    _check_spmv(C3, rp3, ci3, v3, n3, n3, rng)
    _check_spmv(C3, rp3, ci3, v3, n3, n3, rng, alpha=0.5, beta=1.5)
    xh3 = rng.normal(size=n3)
    x3, y3 = mk.Vector(n3, values=xh3), mk.Vector(n3)
    C3.spmv(x3, y3)
    shared_result = y3.get()
    yx3, xx3 = C3.spmv_dot(x3, y3)
    assert np.array_equal(y3.get(), shared_result) and np.isclose(yx3, shared_result @ xh3, rtol=1e-11, atol=1e-9)
    rp_b, ci_b, v_b = C3.to_csr()
    assert np.array_equal(rp_b, rp3) and np.array_equal(ci_b, ci3) and _same_bits(v_b, v3)     # the CSR arrays stay as given
    # c) Jacobi on a): the sweep picks the diagonal out of the row as it goes by, also from the tail batch
    vd = v.copy()
    absum = np.bincount(rows, weights=np.abs(v), minlength=n)
    diag = (ci == rows)
    vd[diag] = 1.5 * absum[rows[diag]]
    o = oracle.jacobi(rp, ci, vd, np.ones(n), tol=1e-10, maxit=300, check_every=1)
    D = mk.Matrix.from_csr(n, n, rp, ci, vd)
    b, xs = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.Jacobi(max_iter=300, tol=1e-10, check_every=1)
    S.setup(D, b, xs)
    res = S.solve(D, b, xs)
    assert res["converged"] == 1 == o["converged"] and res["iterations"] == o["iterations"]
    assert np.allclose(xs.get(), o["x"], rtol=1e-11, atol=0)



@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_spmv_on_golden_matrices(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n, nc = int(g["nrows"]), int(g["ncols"])
    A = mk.Matrix.from_coo(n, nc, g["coo_rows"], g["coo_cols"], g["coo_vals"])
    _check_spmv(A, g["csr_rowptr"], g["csr_col"], g["csr_val"], n, nc, np.random.default_rng(1))


@pytest.mark.parametrize("seed,n,p_same,p_empty", [(1, 5003, 0.6, 0.1), (2, 777, 0.9, 0.0), (3, 4096, 0.45, 0.3), (4, 64, 1.0, 0.0),
                                                   (5, 9000, 1.0, 0.0)])   # (4): few rows, served by a pattern table instead; (5): every row the first row's list
def test_spmv_shared_column_lists_mixtures(seed, n, p_same, p_empty):
    """rows that repeat the list of the row before them, mixed with empty rows and fresh lists (also: every row the same
    list, runs that cross chunk borders, an empty row between two equal lists): product, fused dots and a Jacobi sweep
    through the compact column array against the oracle on the CSR as given"""
    rng = np.random.default_rng(seed)
    lists, prev = [], None
    for r in range(n):
        u = rng.random()
        if prev is not None and prev.size and u < p_same:
            cur = prev.copy()
        elif u > 1.0 - p_empty and r > 0:
            cur = np.zeros(0, np.int64)
        else:
            cur = np.unique(np.concatenate([[r], rng.integers(0, n, int(rng.integers(20, 60)))]))
        lists.append(cur)
        prev = cur
    lens = np.array([l.size for l in lists])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate(lists).astype(np.int32)
    v = rng.normal(size=ci.size)
    rows = np.repeat(np.arange(n), lens)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    i = A.info()
    same = np.array([r > 0 and lists[r].size > 0 and lists[r].size == lists[r - 1].size and np.array_equal(lists[r], lists[r - 1]) for r in range(n)])
    if i.kernel == mk.SPMV_STREAM and i.index_patterns == 0 and lens[same].sum() >= 0.4 * ci.size and ci.size / n >= 12:
        assert i.shared_columns == int(lens[same].sum())
    _check_spmv(A, rp, ci, v, n, n, rng)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=-1.5, beta=0.75)
    xh = rng.normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    yx, xx = A.spmv_dot(x, y)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    assert np.allclose(y.get(), ref, rtol=1e-11, atol=1e-11) and np.isclose(yx, ref @ xh, rtol=1e-10, atol=1e-9)
    # one Jacobi sweep (diagonal made dominant where a row has one)
    vd = v.copy()
    absum = np.bincount(rows, weights=np.abs(v), minlength=n)
    dmask = ci == rows
    vd[dmask] = 2.0 * absum[rows[dmask]] + 1.0
    if np.bincount(rows[dmask], minlength=n).min() == 1:          # every row has its diagonal: no empty rows in this case
        o = oracle.jacobi(rp, ci, vd, np.ones(n), tol=0.0, maxit=3, check_every=1)
        D = mk.Matrix.from_csr(n, n, rp, ci, vd)
        b, xs = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
        S = mk.Jacobi(max_iter=3, tol=0.0, check_every=1)
        S.setup(D, b, xs)
        S.solve(D, b, xs)
        assert np.allclose(xs.get(), o["x"], rtol=1e-12, atol=0)


def _grouped_matrix(rng, nodes, run_lens, list_len, window):
    """rows in runs that share one column list each (run lengths drawn from run_lens), lists of about list_len columns in
    blocks of three within +-window nodes, every entry its own value"""
    lists, lens = [], []
    node = 0
    while node < nodes:
        k = int(rng.choice(run_lens))
        m = max(1, int(rng.integers(max(1, list_len // 2), list_len * 3 // 2 + 1)) // 3)
        nb = np.unique(np.concatenate([[node], rng.integers(max(0, node - window), min(nodes, node + window + 1), m)]))
        lst = (np.repeat(nb * 3, 3) + np.tile(np.arange(3), nb.size)).astype(np.int64)
        for _ in range(k):
            lists.append(lst)
        node += 1
    n = len(lists)
    ncols = 3 * nodes
    lens = np.array([l.size for l in lists])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate(lists).astype(np.int32)
    v = rng.normal(size=ci.size)
    return n, ncols, rp, ci, v


@pytest.mark.parametrize("seed,nodes,run_lens,list_len,window", [
    (1, 3000, [3], 78, 300),                      # three unknowns per node: the shape of the authors' finite-element inputs
    (2, 2500, [1, 2, 3, 4, 5, 9], 40, 200),       # runs longer than a group (cut every four rows) among single rows
    (3, 1200, [3, 6], 200, 900),                  # lists beyond eight entries per lane (the batches of four)
    (4, 4000, [2, 3], 12, 50),                    # short lists, many groups per chunk
    (5, 700, [64], 30, 100),                      # long runs: most groups are continuations of the run before them
])
def test_spmv_row_group_kernel(seed, nodes, run_lens, list_len, window):
    """matrices whose consecutive rows repeat one column list are multiplied group by group (group.hip):
    x gathered once per list, one multiply-add per row of the group.  Product (alpha / beta forms), product + dots and the
    untouched CSR arrays against the oracle on the CSR as given, 1e-12 per row."""
    rng = np.random.default_rng(seed)
    n, ncols, rp, ci, v = _grouped_matrix(rng, nodes, run_lens, list_len, window)
    A = mk.Matrix.from_csr(n, ncols, rp, ci, v)
    i = A.info()
    assert i.kernel == mk.SPMV_STREAM and i.index_patterns == 0 and i.shared_columns > 0
    assert i.row_groups > 0 and i.group_lanes >= 4, (i.row_groups, i.group_lanes)
    # groups: every run of k rows gives ceil(k / 4) of them
    same = np.zeros(n, bool)
    for r in range(1, n):
        same[r] = rp[r + 1] - rp[r] == rp[r] - rp[r - 1] and np.array_equal(ci[rp[r - 1]:rp[r]], ci[rp[r]:rp[r + 1]])
    starts = np.flatnonzero(~same)
    runs = np.diff(np.concatenate([starts, [n]]))
    assert i.row_groups == int(np.sum((runs + 3) // 4))
    _check_spmv(A, rp, ci, v, n, ncols, rng)
    _check_spmv(A, rp, ci, v, n, ncols, rng, alpha=-0.75, beta=1.25)
    _check_spmv(A, rp, ci, v, n, ncols, rng, alpha=2.0, beta=-1.0)
    rp_b, ci_b, v_b = A.to_csr()
    assert np.array_equal(rp_b, rp) and np.array_equal(ci_b, ci) and _same_bits(v_b, v)
    if n == ncols:
        xh = rng.normal(size=n)
        x, y = mk.Vector(n, values=xh), mk.Vector(n, fill=5.0)
        yx, xx = A.spmv_dot(x, y)
        ref = oracle.csr_spmv(rp, ci, v, xh)
        mag = oracle.csr_spmv(rp, ci, np.abs(v), np.abs(xh))
        assert (np.abs(y.get() - ref) / np.maximum(mag, 1e-300)).max() <= SPMV_RTOL
        assert abs(yx - ref @ xh) <= 1e-12 * np.abs(ref * xh).sum() and abs(xx - xh @ xh) <= 1e-12 * (xh @ xh)


def test_pcg_on_a_three_unknowns_per_node_matrix_through_the_row_group_kernel():
    """CG (solver id 50 and 51) on a symmetric positive definite matrix with 3 x 3 node blocks: K1 runs on the row-group kernel;
    iteration count, history and solution against orc_pcg"""
    rng = np.random.default_rng(11)
    nodes = 2000
    nb = [np.unique(np.concatenate([[i], rng.integers(max(0, i - 60), min(nodes, i + 61), 9)])) for i in range(nodes)]
    adj = np.zeros((0, 2), np.int64)
    pairs = np.concatenate([np.stack([np.full(q.size, i), q], 1) for i, q in enumerate(nb)])
    pairs = np.unique(np.concatenate([pairs, pairs[:, ::-1]]), axis=0)                 # symmetric node graph, sorted by (row, col)
    r3 = (pairs[:, 0, None, None] * 3 + np.arange(3)[None, :, None] + np.zeros((1, 1, 3), np.int64)).reshape(-1)
    c3 = (pairs[:, 1, None, None] * 3 + np.arange(3)[None, None, :] + np.zeros((1, 3, 1), np.int64)).reshape(-1)
    order = np.lexsort((c3, r3))
    r3, c3 = r3[order], c3[order]
    n = 3 * nodes
    w = rng.uniform(-1.0, 1.0, r3.size)
    lo, hi = np.minimum(r3, c3), np.maximum(r3, c3)
    key = lo * n + hi
    _, inv = np.unique(key, return_inverse=True)
    w = rng.uniform(-1.0, 1.0, inv.max() + 1)[inv]                                     # symmetric values
    absum = np.bincount(r3, weights=np.abs(w), minlength=n)
    w[r3 == c3] = 1.2 * absum[r3[r3 == c3]] + 1.0                                      # strictly dominant diagonal: SPD
    rp = np.concatenate([[0], np.cumsum(np.bincount(r3, minlength=n))]).astype(np.int32)
    ci = c3.astype(np.int32)
    A = mk.Matrix.from_csr(n, n, rp, ci, w)
    assert A.info().row_groups > 0
    bh = rng.normal(size=n)
    for precond in (0, 1):
        o = oracle.pcg(rp, ci, w, bh, tol=1e-10, precond=precond, maxit=500, want_norms=True)
        b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
        S = mk.PCG(max_iter=500, tol=1e-10, two_norm=1, rel_change=1, precond=precond, logging=1)
        S.setup(A, b, x)
        res = S.solve(A, b, x)
        assert res["converged"] == 1 == o["converged"] and abs(res["iterations"] - o["iterations"]) <= CG_ITER_TOL
        assert np.linalg.norm(x.get() - o["x"]) <= 1e-8 * np.linalg.norm(o["x"])
        hist = S.residual_history(1000)
        m = min(hist.size, o["rel_norms"].size, 30)
        assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)


def _banded_matrix(rng, n, mean, hw, empty=()):
    """rows of about `mean` entries, columns in runs of three anywhere within +-hw of the diagonal (tools/fem_probe.py's `rand`
    shape), every entry its own value; rows in `empty` ranges have no entry"""
    lens = rng.integers(mean * 2 // 3, mean * 4 // 3 + 1, n)
    for a, b in empty:
        lens[a:b] = 0
    rp = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(rp[-1])
    rows = np.repeat(np.arange(n), lens)
    base = rows + rng.integers(-hw, hw + 1, nnz)
    ci = np.clip(base - base % 3 + (np.arange(nnz) % 3), 0, n - 1)
    ci = ci[np.lexsort((ci, rows))].astype(np.int32)
    return rp.astype(np.int32), ci, rng.normal(size=nnz)


class _env:
    """environment switches the library reads per matrix (band plan), set around one construction"""
    def __init__(self, **kw):
        self.kw = kw
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update({k: str(v) for k, v in self.kw.items()})
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("seed,n,mean,hw,empty", [
    (1, 30000, 45, 3000, ()),                          # tools/fem_probe.py `rand 2000000 45 3000` in small: panels of 1024 rows
    (2, 20011, 20, 700, ()),                           # odd row count: the last panel and its last chunk are partial
    (3, 25000, 80, 400, ((5000, 5700), (24990, 25000))),   # 16 lanes per row; a panel without entries, empty rows at the end
    (4, 9000, 30, 150, ((0, 3),)),                     # fewer rows than two of the largest panels; empty rows in front
    (5, 12000, 100, 2000, ()),                         # rows beyond eight entries per lane (the batches of four of pass 2)
])
def test_spmv_band_kernel(seed, n, mean, hw, empty):
    """VERDICT r2 item 3 (x-window staging in LDS for banded rows, band.hip): rows that share neither offsets nor column lists
    but stay within a bounded distance of the diagonal are multiplied panel by panel with the panel's window of x in LDS.
    Product (alpha / beta forms), product + dots and the untouched CSR arrays against the oracle, 1e-12 per row; the same
    matrix without the plan (MK_NO_BAND) takes the stream kernel proper and agrees too."""
    rng = np.random.default_rng(seed)
    rp, ci, v = _banded_matrix(rng, n, mean, hw, empty)
    with _env(MK_BAND_MIN_ROWS=1000):
        A = mk.Matrix.from_csr(n, n, rp, ci, v)
        i = A.info()
    assert i.kernel == mk.SPMV_STREAM and i.index_patterns == 0 and i.row_groups == 0
    assert 0 < i.band_window <= 2 * hw + 4096 + 3 and i.band_strips == 0, (i.band_window, i.band_strips)
    _check_spmv(A, rp, ci, v, n, n, rng)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=-0.75, beta=1.25)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=2.0, beta=-1.0)
    rp_b, ci_b, v_b = A.to_csr()
    assert np.array_equal(rp_b, rp) and np.array_equal(ci_b, ci) and _same_bits(v_b, v)
    xh = rng.normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n, fill=5.0)
    yx, xx = A.spmv_dot(x, y)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    mag = oracle.csr_spmv(rp, ci, np.abs(v), np.abs(xh))
    assert (np.abs(y.get() - ref) / np.maximum(mag, 1e-300)).max() <= SPMV_RTOL
    assert abs(yx - ref @ xh) <= 1e-12 * np.abs(ref * xh).sum() and abs(xx - xh @ xh) <= 1e-12 * (xh @ xh)
    with _env(MK_BAND_MIN_ROWS=1000, MK_NO_BAND=1):
        B = mk.Matrix.from_csr(n, n, rp, ci, v)
        assert B.info().band_window == 0 and B.info().kernel == mk.SPMV_STREAM
    y2 = mk.Vector(n, fill=-1.0)
    B.spmv(x, y2)
    assert (np.abs(y2.get() - y.get()) / np.maximum(mag, 1e-300)).max() <= SPMV_RTOL
    # the other kernels still serve the matrix on request
    for k in (mk.SPMV_VECTOR, mk.SPMV_SPLIT):
        A.set_spmv_kernel(k)
        _check_spmv(A, rp, ci, v, n, n, rng)


@pytest.mark.parametrize("seed,n,mean,hw,empty", [
    (1, 60000, 80, 12000, ()),                         # tools/fem_probe.py `rand 1500000 80 20000` in small: 5 strips
    (2, 50001, 64, 8000, ((100, 400), (49990, 50001))),   # 3 strips; empty rows; odd row count
    (3, 40000, 130, 30000, ()),                        # 12 strips, the most there are
])
def test_spmv_band_kernel_diagonal_strips(seed, n, mean, hw, empty):
    """rows that reach further than one LDS window: the matrix is cut into diagonal strips (entries grouped by column - row),
    each served by the band kernel, y accumulated strip by strip; the dots ride in the last strip's launch.  Against the
    oracle on the CSR as given, 1e-12 per row (the row's sum is formed strip by strip: another order, still a fixed one)."""
    rng = np.random.default_rng(seed)
    rp, ci, v = _banded_matrix(rng, n, mean, hw, empty)
    with _env(MK_BAND_MIN_ROWS=1000, MK_BAND_STRIP_MIN_ROWS=1000):   # (strips are for matrices of 750 000 rows and more by default)
        A = mk.Matrix.from_csr(n, n, rp, ci, v)
        i = A.info()
    assert i.kernel == mk.SPMV_STREAM and i.band_strips >= 2 and 0 < i.band_window < 8000, (i.band_strips, i.band_window)
    assert -(-(2 * hw) // 7000) <= i.band_strips <= -(-(2 * hw + 3) // 5400), i.band_strips   # (the fewest strips whose windows fit beside the stage)
    _check_spmv(A, rp, ci, v, n, n, rng)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=-0.75, beta=1.25)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=2.0, beta=-1.0)
    rp_b, ci_b, v_b = A.to_csr()
    assert np.array_equal(rp_b, rp) and np.array_equal(ci_b, ci) and _same_bits(v_b, v)
    xh = rng.normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n, fill=5.0)
    yx, xx = A.spmv_dot(x, y)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    mag = oracle.csr_spmv(rp, ci, np.abs(v), np.abs(xh))
    assert (np.abs(y.get() - ref) / np.maximum(mag, 1e-300)).max() <= SPMV_RTOL
    assert abs(yx - ref @ xh) <= 1e-12 * np.abs(ref * xh).sum() and abs(xx - xh @ xh) <= 1e-12 * (xh @ xh)
    y1 = y.get().copy()
    A.spmv_dot(x, y)
    assert _same_bits(y.get(), y1)                     # the same bits from run to run
    with _env(MK_BAND_MIN_ROWS=1000):     # below the strips' size threshold (or with MK_NO_BAND_STRIPS=1): one window with one block per CU where that fits, else the stream kernel
        B = mk.Matrix.from_csr(n, n, rp, ci, v)
        ib = B.info()
        assert ib.band_strips == 0 and (ib.band_window == 0 or ib.band_window > 2 * hw)
    y2 = mk.Vector(n, fill=-1.0)
    B.spmv(x, y2)
    assert (np.abs(y2.get() - y1) / np.maximum(mag, 1e-300)).max() <= SPMV_RTOL


def test_spmv_band_kernel_serves_rare_fat_rows():
    """a few fat rows (constraint rows) among regular ones fail the stream kernel's test on the longest row; the band kernel packs
    its chunks by entries and serves the matrix instead of the split kernel; with more than one fat row in a thousand the split
    kernel keeps it"""
    rng = np.random.default_rng(31)
    n = 40000
    lens = rng.integers(30, 61, n)
    fat = rng.choice(n, 30, replace=False)
    lens[fat] = rng.integers(600, 1400, 30)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(n), lens)
    ci = np.clip(rows + rng.integers(-1500, 1501, rows.size), 0, n - 1)
    ci = ci[np.lexsort((ci, rows))].astype(np.int32)
    v = rng.normal(size=ci.size)
    with _env(MK_BAND_MIN_ROWS=1000):
        A = mk.Matrix.from_csr(n, n, rp, ci, v)
        i = A.info()
    assert i.kernel == mk.SPMV_STREAM and i.band_window > 0 and i.max_row_len >= 600, (i.kernel, i.band_window, i.max_row_len)
    _check_spmv(A, rp, ci, v, n, n, rng)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=-1.5, beta=0.5)
    xh = rng.normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n, fill=5.0)
    yx, xx = A.spmv_dot(x, y)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    mag = oracle.csr_spmv(rp, ci, np.abs(v), np.abs(xh))
    assert (np.abs(y.get() - ref) / np.maximum(mag, 1e-300)).max() <= SPMV_RTOL
    assert abs(yx - ref @ xh) <= 1e-12 * np.abs(ref * xh).sum()
    # a Jacobi sweep on it takes the stream kernel behind the band plan (the diagonal is not stored in every row: values only)
    lens[rng.choice(n, 400, replace=False)] = 700          # one fat row in a hundred: the split kernel's
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(n), lens)
    ci = np.clip(rows + rng.integers(-1500, 1501, rows.size), 0, n - 1)
    ci = ci[np.lexsort((ci, rows))].astype(np.int32)
    v = rng.normal(size=ci.size)
    with _env(MK_BAND_MIN_ROWS=1000):
        B = mk.Matrix.from_csr(n, n, rp, ci, v)
        assert B.info().kernel == mk.SPMV_SPLIT and B.info().band_window == 0
    _check_spmv(B, rp, ci, v, n, n, rng)


def test_spmv_band_kernel_one_block_per_cu_window():
    """a window beyond half a CU's LDS on rows too short for strips (fewer than six entries per row and strip): one window, one
    block per CU"""
    rng = np.random.default_rng(21)
    n = 60000
    rp, ci, v = _banded_matrix(rng, n, 17, 6000)   # (about 16.5 entries per row: below 3 strips x 6)
    with _env(MK_BAND_MIN_ROWS=1000):
        A = mk.Matrix.from_csr(n, n, rp, ci, v)
        i = A.info()
    assert i.band_strips == 0 and 12000 < i.band_window < 14000, (i.band_strips, i.band_window)
    _check_spmv(A, rp, ci, v, n, n, rng)
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=0.5, beta=2.0)


def test_spmv_band_kernel_declines_wide_rows_and_small_matrices():
    """no plan when the window would not fit the LDS, when rows are shorter than asked for, or below the size threshold"""
    rng = np.random.default_rng(7)
    rp, ci, v = _banded_matrix(rng, 40000, 30, 18000)
    with _env(MK_BAND_MIN_ROWS=1000):
        A = mk.Matrix.from_csr(40000, 40000, rp, ci, v)
        assert A.info().band_window == 0
    _check_spmv(A, rp, ci, v, 40000, 40000, rng)
    rp, ci, v = _banded_matrix(rng, 30000, 9, 300)
    with _env(MK_BAND_MIN_ROWS=1000, MK_BAND_MIN_LEN=16):
        A = mk.Matrix.from_csr(30000, 30000, rp, ci, v)
        assert A.info().band_window == 0          # mean row length below the threshold asked for
    with _env(MK_BAND_MIN_ROWS=1000):
        A = mk.Matrix.from_csr(30000, 30000, rp, ci, v)
        assert A.info().band_window > 0           # (default: from four entries per row on)
    _check_spmv(A, rp, ci, v, 30000, 30000, rng)
    rp, ci, v = _banded_matrix(rng, 30000, 30, 300)
    A = mk.Matrix.from_csr(30000, 30000, rp, ci, v)
    assert A.info().band_window == 0              # default threshold: 100 000 rows


def test_pcg_through_the_band_kernel():
    """CG (ids 50 and 51) on a symmetric positive definite banded matrix: K1 runs on the band kernel; iteration count, history
    and solution against orc_pcg"""
    rng = np.random.default_rng(12)
    n = 24000
    r = np.repeat(np.arange(n), 12)
    off = rng.integers(-900, 901, r.size)
    c = np.where((r + off < 0) | (r + off >= n), r - off, r + off)   # (reflected, not clipped: clipping piles hundreds of entries into the border rows)
    pairs = np.unique(np.concatenate([np.stack([r, c], 1), np.stack([c, r], 1), np.stack([np.arange(n)] * 2, 1)]), axis=0)
    r, c = pairs[:, 0], pairs[:, 1]
    key = np.minimum(r, c) * n + np.maximum(r, c)
    _, inv = np.unique(key, return_inverse=True)
    w = rng.uniform(-1.0, 1.0, inv.max() + 1)[inv]
    absum = np.bincount(r, weights=np.abs(w), minlength=n)
    w[r == c] = 1.2 * absum[r[r == c]] + 1.0
    rp = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))]).astype(np.int32)
    ci = c.astype(np.int32)
    with _env(MK_BAND_MIN_ROWS=1000):
        A = mk.Matrix.from_csr(n, n, rp, ci, w)
        assert A.info().band_window > 0
    bh = rng.normal(size=n)
    for precond in (0, 1):
        o = oracle.pcg(rp, ci, w, bh, tol=1e-10, precond=precond, maxit=500, want_norms=True)
        b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
        S = mk.PCG(max_iter=500, tol=1e-10, two_norm=1, rel_change=1, precond=precond, logging=1)
        S.setup(A, b, x)
        res = S.solve(A, b, x)
        assert res["converged"] == 1 == o["converged"] and abs(res["iterations"] - o["iterations"]) <= CG_ITER_TOL
        assert np.linalg.norm(x.get() - o["x"]) <= 1e-8 * np.linalg.norm(o["x"])
        hist = S.residual_history(1000)
        m = min(hist.size, o["rel_norms"].size, 30)
        assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)


@pytest.mark.parametrize("dof,nx,ny", [(2, 37, 23), (4, 20, 31)])
def test_spmv_pattern_shapes_short_rows(dof, nx, ny):
    """several unknowns per node of a 2-D 5-point grid, every entry its own value: the index patterns of a node's rows differ
    by a shift only, so the stream kernel stages shapes + (shape, shift) per pattern; one lane per row, hence the
    reference's summation order: bit-identical to the oracle, also with empty rows in between"""
    rng = np.random.default_rng(dof)
    nn = nx * ny
    lists = []
    for node in range(nn):
        ix, iy = node % nx, node // nx
        nb = [node + d for d, ok in ((-nx, iy > 0), (-1, ix > 0), (0, True), (1, ix < nx - 1), (nx, iy < ny - 1)) if ok]
        cols = np.array([dof * j + e for j in nb for e in range(dof)], np.int64)
        for d in range(dof):
            lists.append(cols if not (node % 97 == 5 and d == 1) else np.zeros(0, np.int64))     # a few empty rows
    n = dof * nn
    lens = np.array([l.size for l in lists])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate(lists).astype(np.int32)
    v = rng.normal(size=ci.size)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    i = A.info()
    assert i.kernel == mk.SPMV_STREAM and i.stream_lanes <= 2 and dof * 9 <= i.index_patterns <= dof * 9 + 1
    xh = rng.normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    A.spmv(x, y)
    if i.stream_lanes == 1:
        assert np.array_equal(y.get(), oracle.csr_spmv(rp, ci, v, xh))
    _check_spmv(A, rp, ci, v, n, n, rng, alpha=0.25, beta=-2.0)
    yx, xx = A.spmv_dot(x, y)
    assert np.isclose(yx, oracle.csr_spmv(rp, ci, v, xh) @ xh, rtol=1e-11, atol=1e-9)


def test_selector_long_row_among_short_ones():
    """short rows with a few long ones: the chunk capacity would fit the stream kernel, the rows of more than 32 entries per
    lane send the matrix to the split kernel; without them it stays on the stream kernel (both against the oracle)"""
    rng = np.random.default_rng(31)
    n = 20000
    for long_len, expect in ((400, mk.SPMV_SPLIT), (0, mk.SPMV_STREAM)):
        lens = rng.integers(3, 10, n)
        if long_len:
            lens[777::5000] = long_len
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rows = np.repeat(np.arange(n), lens)
        ci = rng.integers(0, n, rp[-1])
        ci = ci[np.lexsort((ci, rows))].astype(np.int32)
        v = rng.normal(size=rp[-1])
        A = mk.Matrix.from_csr(n, n, rp, ci, v)
        i = A.info()
        assert i.kernel == expect and i.stream_lanes == 1 and i.stream_chunk <= 4088
        _check_spmv(A, rp, ci, v, n, n, rng, alpha=2.0, beta=-0.5)
        xh = rng.normal(size=n)
        x, y = mk.Vector(n, values=xh), mk.Vector(n)
        yx, xx = A.spmv_dot(x, y)          # split kernel: the long rows are summed by a wave each, also for the fused dots
        ref = oracle.csr_spmv(rp, ci, v, xh)
        assert np.allclose(y.get(), ref, rtol=1e-11, atol=1e-11) and np.isclose(yx, ref @ xh, rtol=1e-10, atol=1e-9) and np.isclose(xx, xh @ xh, rtol=1e-12)
        assert (yx, xx) == A.spmv_dot(x, y)                                  # and the same bits every time
        if long_len:                       # the stream kernel still serves it when asked to
            A.set_spmv_kernel(mk.SPMV_STREAM)
            _check_spmv(A, rp, ci, v, n, n, rng)


def test_spmv_skewed_rows_rmat():
    scale, ne = 14, 1 << 18
    r, c, v = oracle.rmat_coo(scale, ne)
    n = 1 << scale
    rp, ci, val = oracle.coo_to_csr(n, r, c, v)
    A = mk.Matrix.from_coo(n, n, r, c, v)
    info = A.info()
    assert info.max_row_len == np.diff(rp).max() and info.nnz == ne
    assert sum(info.hist) == n and info.hist[0] == np.sum(np.diff(rp) == 0)
    _check_spmv(A, rp, ci, val, n, n, np.random.default_rng(5))


@pytest.mark.parametrize("case", ["one_huge_row", "many_long_rows", "long_then_empty_tail", "no_entries", "dense_block"])
def test_spmv_split_kernel_row_splitting(case):
    """rows that cross one or many 2048-entry chunk borders, empty rows around them"""
    rng = np.random.default_rng(len(case))
    if case == "one_huge_row":
        n, lens = 50, np.zeros(50, np.int64)
        lens[7] = 10000; lens[8] = 3; lens[49] = 1
    elif case == "many_long_rows":
        n = 400
        lens = rng.integers(0, 30, n)
        lens[rng.integers(0, n, 12)] = rng.integers(1500, 9000, 12)
    elif case == "long_then_empty_tail":
        n, lens = 3000, np.zeros(3000, np.int64)
        lens[0] = 2048; lens[1] = 2048 * 3; lens[2] = 1; lens[10] = 4095
    elif case == "no_entries":
        n, lens = 100, np.zeros(100, np.int64)
    else:
        n, lens = 64, np.full(64, 512)
    ncols = 5000
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = rng.integers(0, ncols, rp[-1]).astype(np.int32)
    v = rng.normal(size=rp[-1])
    A = mk.Matrix.from_csr(n, ncols, rp, ci, v)
    A.set_spmv_kernel(mk.SPMV_SPLIT)
    _check_spmv(A, rp, ci, v, n, ncols, rng)
    _check_spmv(A, rp, ci, v, n, ncols, rng, alpha=-2.0, beta=0.5)
    if case == "one_huge_row":
        assert A.info().long_rows == 1


def test_spmv_split_kernel_large_matrix_stream():
    """above 200 MB of (col, val) the split kernel reads the matrix nontemporally (other instantiation): 20 M entries"""
    scale, ne = 18, 20_000_000
    d = mk.DeviceCOO.rmat(scale, ne, seed=9)
    A = mk.Matrix.from_device_coo(d)
    assert A.info().kernel == mk.SPMV_SPLIT and A.info().nnz == ne
    rp, ci, val = A.to_csr()
    n = 1 << scale
    _check_spmv(A, rp, ci, val, n, n, np.random.default_rng(2))
    _check_spmv(A, rp, ci, val, n, n, np.random.default_rng(3), alpha=0.5, beta=-1.5)
    xh = np.random.default_rng(4).normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    ytx, xtx = A.spmv_dot(x, y)
    ref = oracle.csr_spmv(rp, ci, val, xh)
    mag = oracle.csr_spmv(rp, ci, np.abs(val), np.abs(xh))
    assert (np.abs(y.get() - ref) / np.maximum(mag, 1e-300)).max() <= SPMV_RTOL
    assert abs(ytx - ref @ xh) <= 1e-12 * (mag @ np.abs(xh))


def test_spmv_split_kernel_column_relabelling_is_bit_identical():
    """power-law matrices whose x outgrows the caches get their columns renumbered hot-first (x is gathered into
    that order once per product): same pairs in the same order, so the bits must not change.  The size threshold
    is lowered in a child process (read once); the parent computes the same products without relabelling."""
    import hashlib, subprocess, sys
    code = (
        "import hashlib, numpy as np, iterative_solvers_amd as mk, oracle\n"
        "r, c, v = oracle.rmat_coo(14, 1 << 18, seed=5)\n"
        "n = 1 << 14\n"
        "A = mk.Matrix.from_coo(n, n, r, c, v)\n"
        "assert A.info().kernel == mk.SPMV_SPLIT\n"
        "xh = np.random.default_rng(6).normal(size=n)\n"
        "yh = np.random.default_rng(7).normal(size=n)\n"
        "x, y = mk.Vector(n, values=xh), mk.Vector(n, values=yh)\n"
        "h = hashlib.sha1()\n"
        "A.spmv(x, y, -0.5, 2.0); h.update(y.get().tobytes())\n"
        "A.spmv(x, y); h.update(y.get().tobytes())\n"
        "yx, xx = A.spmv_dot(x, y); h.update(y.get().tobytes()); h.update(np.array([yx, xx]).tobytes())\n"
        "rp, ci, val = A.to_csr()\n"
        "orp, oci, oval = oracle.coo_to_csr(n, r, c, v)\n"
        "assert np.array_equal(ci, oci) and np.array_equal(rp, orp)\n"   # get_csr still returns the original numbering
        "print('HASH', h.hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag, extra in (("plain", {"MK_NO_COLUMN_RELABEL": "1"}), ("relabelled", {"MK_RELABEL_MIN_COLS": "1000"})):
        o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=root, **extra), capture_output=True, text=True, timeout=110)
        assert o.returncode == 0 and "HASH" in o.stdout, (tag, o.stdout[-300:], o.stderr[-2000:])
        out[tag] = o.stdout.split("HASH")[1].split()[0]
    assert out["plain"] == out["relabelled"]


def test_spmv_split_fused_dot_square():
    scale, ne = 13, 1 << 18
    r, c, v = oracle.rmat_coo(scale, ne, seed=3)
    n = 1 << scale
    rp, ci, val = oracle.coo_to_csr(n, r, c, v)
    A = mk.Matrix.from_csr(n, n, rp, ci, val)
    assert A.info().kernel == mk.SPMV_SPLIT                   # skewed rows -> nnz-balanced kernel
    xh = np.random.default_rng(1).normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    ytx, xtx = A.spmv_dot(x, y)
    ref = oracle.csr_spmv(rp, ci, val, xh)
    mag = oracle.csr_spmv(rp, ci, np.abs(val), np.abs(xh))
    assert (np.abs(y.get() - ref) / np.maximum(mag, 1e-300)).max() <= SPMV_RTOL
    assert abs(ytx - ref @ xh) <= 1e-12 * (mag @ np.abs(xh)) and abs(xtx - xh @ xh) <= 1e-13 * (xh @ xh)


def test_spmv_dot_fused():
    rp, ci, v = oracle.stencil_csr(7, 20, 21, 22)
    n = rp.size - 1
    rng = np.random.default_rng(11)
    xh = rng.normal(size=n)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    ytx, xtx = A.spmv_dot(x, y)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    assert np.allclose(y.get(), ref, rtol=1e-13, atol=1e-13)
    assert abs(ytx - ref @ xh) <= 1e-12 * np.abs(ref * xh).sum()
    assert abs(xtx - xh @ xh) <= 1e-13 * (xh @ xh)


def test_spmv_shape_errors():
    rp, ci, v = oracle.stencil_csr(5, 4, 4)
    A = mk.Matrix.from_csr(16, 16, rp, ci, v)
    with pytest.raises(mk.MkError):
        A.spmv(mk.Vector(15), mk.Vector(16))
    with pytest.raises(mk.MkError):
        mk.Matrix.from_csr(16, 16, rp, np.where(ci == 15, 16, ci), v)   # column out of range


# ------------------------------------------------------------------ PCG / Jacobi
@pytest.mark.parametrize("precond", [0, 1])
@pytest.mark.parametrize("tol", [1e-6, 1e-10])
def test_pcg_poisson2d_matches_oracle(precond, tol):
    # config 1 shape, scaled down: 2D 5-point Poisson, b = 1, x0 = 0 (solvers.c:404-426)
    rp, ci, v = oracle.stencil_csr(5, 96, 96)
    n = rp.size - 1
    o = oracle.pcg(rp, ci, v, np.ones(n), tol=tol, precond=precond, maxit=1000, want_norms=True)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=1000, tol=tol, two_norm=1, rel_change=1, precond=precond, logging=1, check_every=7)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    assert res["converged"] == 1 and o["converged"] == 1
    assert abs(res["iterations"] - o["iterations"]) <= CG_ITER_TOL
    assert res["rel_res_norm"] < tol
    xs = x.get()
    true_res = np.linalg.norm(np.ones(n) - oracle.csr_spmv(rp, ci, v, xs)) / np.sqrt(n)
    assert true_res < tol * 1.5
    assert np.linalg.norm(xs - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
    hist = S.residual_history(2000)
    assert hist.size == res["iterations"] + 1
    m = min(hist.size, o["rel_norms"].size, 40)
    assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)


def test_pcg_3d_jacobi_precond_random_diag():
    # config 2 shape scaled down; a varying diagonal makes the preconditioner matter
    rp, ci, v = oracle.stencil_csr(7, 24, 24, 24)
    n = rp.size - 1
    rng = np.random.default_rng(2)
    scale = rng.uniform(1.0, 50.0, n)
    rows = np.repeat(np.arange(n), np.diff(rp))
    v = v * np.sqrt(scale[rows] * scale[ci])         # D^1/2 A D^1/2 stays SPD
    bh = rng.normal(size=n)
    for precond in (0, 1):
        o = oracle.pcg(rp, ci, v, bh, tol=1e-8, precond=precond, maxit=2000)
        A = mk.Matrix.from_csr(n, n, rp, ci, v)
        b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
        S = mk.PCG(max_iter=2000, tol=1e-8, precond=precond)
        S.setup(A, b, x)
        res = S.solve(A, b, x)
        assert res["converged"] == 1
        assert abs(res["iterations"] - o["iterations"]) <= CG_ITER_TOL, (res, o["iterations"])
        assert np.linalg.norm(x.get() - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])


_HALFX_SNIPPET = r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import iterative_solvers_amd as mk
import oracle
for kind, dims, precond in ((7, (32, 16, 20), 0), (27, (16, 18, 9), 1), (5, (96, 96, 1), 1)):
    rp, ci, v = oracle.stencil_csr(kind, *dims)
    n = rp.size - 1
    bh = np.random.default_rng(5).normal(size=n)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    A.set_spmv_kernel(mk.SPMV_SWEEP)
    o = oracle.pcg(rp, ci, v, bh, tol=1e-9, precond=precond, maxit=1000, want_norms=True)
    b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=1000, tol=1e-9, precond=precond, logging=1, check_every=5)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    assert res["converged"] == 1 and abs(res["iterations"] - o["iterations"]) <= 2, (res, o["iterations"])
    assert np.linalg.norm(x.get() - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
    hist = S.residual_history(2000)
    m = min(hist.size, o["rel_norms"].size, 40)
    assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)
    for k in (1, 2, 3, 6):
        b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
        S = mk.PCG(max_iter=k, tol=1e-30, precond=precond)
        S.setup(A, b, x)
        assert S.solve(A, b, x)["iterations"] == k
        ok = oracle.pcg(rp, ci, v, bh, tol=1e-30, precond=precond, maxit=k)
        assert np.allclose(x.get(), ok["x"], rtol=1e-9, atol=1e-13), (kind, k)
print("halfx ok")
"""


@pytest.mark.parametrize("no_halved", ["0", "1"])
def test_pcg_fused_sweep_step_halved_solution_update(no_halved):
    """the fused CG step updates x every second step with both steps' terms (sweep MODE 7 / 8; MK_NO_HALVED_X=1: in every step).
    Both orders against orc_pcg -- counts, history, solutions after exactly k steps for odd and even k (a solve that ends on an
    even step applies the pending update after the loop); the switch is read once per process, hence the child processes"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MK_NO_HALVED_X=no_halved)
    out = subprocess.run([sys.executable, "-c", _HALFX_SNIPPET, root], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "halfx ok" in out.stdout, out.stderr[-3000:]


@pytest.mark.parametrize("kind,dims", [(5, (96, 96, 1)), (7, (32, 16, 20)), (27, (16, 18, 9)), (7, (64, 32, 9)), (7, (640, 4, 6))])
@pytest.mark.parametrize("precond", [0, 1])
@pytest.mark.parametrize("rel_change", [0, 1])
def test_pcg_fused_sweep_step_matches_oracle(kind, dims, precond, rel_change):
    """matrices served by the sweep kernel take the fused CG step (A p recomputed instead of stored, x updated
    one kernel later, stop flags handed from kernel to kernel): same iteration counts, history and solution"""
    rp, ci, v = oracle.stencil_csr(kind, *dims)
    n = rp.size - 1
    tol = 1e-9
    o = oracle.pcg(rp, ci, v, np.ones(n), tol=tol, precond=precond, maxit=1000, want_norms=True, rel_change=rel_change)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    A.set_spmv_kernel(mk.SPMV_SWEEP)
    for check_every in (1, 7):
        b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
        S = mk.PCG(max_iter=1000, tol=tol, two_norm=1, rel_change=rel_change, precond=precond, logging=1, check_every=check_every)
        S.setup(A, b, x)
        res = S.solve(A, b, x)
        assert res["converged"] == 1 and o["converged"] == 1
        assert abs(res["iterations"] - o["iterations"]) <= CG_ITER_TOL, (res, o["iterations"])
        assert res["rel_res_norm"] < tol
        xs = x.get()
        assert np.linalg.norm(np.ones(n) - oracle.csr_spmv(rp, ci, v, xs)) / np.sqrt(n) < tol * 1.5
        assert np.linalg.norm(xs - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
        hist = S.residual_history(2000)
        assert hist.size == res["iterations"] + 1
        m = min(hist.size, o["rel_norms"].size, 40)
        assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)
    # exactly k iterations: x is complete although its update rides one kernel behind
    for k in (1, 2, 5):
        b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
        S = mk.PCG(max_iter=k, tol=1e-30, rel_change=rel_change, precond=precond)
        S.setup(A, b, x)
        res = S.solve(A, b, x)
        ok = oracle.pcg(rp, ci, v, np.ones(n), tol=1e-30, precond=precond, maxit=k, rel_change=rel_change)
        assert res["iterations"] == k and res["converged"] == 0
        assert np.isclose(res["rel_res_norm"], ok["rel_res_norm"], rtol=1e-9)
        assert np.allclose(x.get(), ok["x"], rtol=1e-9, atol=1e-13)
    # zero right-hand side and a restart from the previous solution
    b0, x0 = mk.Vector(n, fill=0.0), mk.Vector(n, fill=3.0)
    S = mk.PCG(max_iter=10, tol=1e-9, precond=precond, rel_change=rel_change)
    S.setup(A, b0, x0)
    assert S.solve(A, b0, x0)["iterations"] == 0 and np.all(x0.get() == 0.0)


def test_pcg_maxit_zero_rhs_and_restart():
    rp, ci, v = oracle.stencil_csr(5, 40, 40)
    n = rp.size - 1
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=5, tol=1e-12)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    o = oracle.pcg(rp, ci, v, np.ones(n), tol=1e-12, maxit=5)
    assert res["iterations"] == 5 and res["converged"] == 0
    assert np.isclose(res["rel_res_norm"], o["rel_res_norm"], rtol=1e-10)
    assert np.allclose(x.get(), o["x"], rtol=1e-10)
    # niter > 1 in the reference re-solves from the previous solution (solvers.c:753-754)
    res2 = S.solve(A, b, x)
    o2 = oracle.pcg(rp, ci, v, np.ones(n), x0=o["x"], tol=1e-12, maxit=5)
    assert np.isclose(res2["rel_res_norm"], o2["rel_res_norm"], rtol=1e-9)
    # zero right-hand side: hypre returns x = b with zero iterations
    b0 = mk.Vector(n, fill=0.0)
    res0 = S.solve(A, b0, x)
    assert res0["iterations"] == 0 and np.all(x.get() == 0.0)
    # NaN / Inf in b or x0 is an error (hypre_PCGSolve's ieee_check), not a zero right-hand side
    for bad in (np.nan, np.inf):
        bh = np.ones(n)
        bh[n // 2] = bad
        with pytest.raises(mk.MkError, match="NaN or Inf"):
            S.solve(A, mk.Vector(n, values=bh), mk.Vector(n, fill=0.0))
    with pytest.raises(mk.MkError, match="NaN or Inf"):
        S.solve(A, b, mk.Vector(n, values=np.where(np.arange(n) == 3, np.nan, 0.0)))
    J = mk.Jacobi(max_iter=5)
    J.setup(A, b, x)
    with pytest.raises(mk.MkError, match="NaN or Inf"):
        J.solve(A, mk.Vector(n, values=np.where(np.arange(n) == 3, np.nan, 1.0)), mk.Vector(n, fill=0.0))


def test_pcg_tol_zero_runs_exactly_maxit():
    rp, ci, v = oracle.stencil_csr(7, 12, 12, 12)
    n = rp.size - 1
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=20, tol=0.0, check_every=4)
    S.setup(A, b, x)
    assert S.solve(A, b, x)["iterations"] == 20


# ------------------------------------------------------------------ BiCGSTAB (solver id 6)
def _convection_diffusion(kind, dims, c=0.4):
    rp, ci, v = oracle.stencil_csr(kind, *dims)
    n = rp.size - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    v = v.copy()
    v[ci == rows + 1] += c          # upwind-ish first-order term: nonsymmetric
    v[ci == rows - 1] -= c
    return rp, ci, v, n


@pytest.mark.parametrize("kind,dims,kernel", [(5, (64, 48, 1), None), (7, (20, 16, 12), None), (7, (32, 16, 12), "sweep"),
                                              (7, (20, 16, 12), "vector")])
def test_bicgstab_matches_oracle(kind, dims, kernel):
    rp, ci, v, n = _convection_diffusion(kind, dims)
    rng = np.random.default_rng(11)
    bh = rng.normal(size=n)
    tol = 1e-9
    o = oracle.bicgstab(rp, ci, v, bh, tol=tol, maxit=500, want_norms=True)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    if kernel == "sweep":
        A.set_spmv_kernel(mk.SPMV_SWEEP)
    elif kernel == "vector":
        A.set_spmv_kernel(mk.SPMV_VECTOR)
    for check_every in (1, 16):
        b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
        S = mk.BiCGSTAB(max_iter=500, tol=tol, logging=1, check_every=check_every)
        S.setup(A, b, x)
        res = S.solve(A, b, x)
        assert res["converged"] == 1 == o["converged"]
        # irregular convergence magnifies the other summation order of the dots: 5 % of the count, at least 3
        assert abs(res["iterations"] - o["iterations"]) <= max(3, 0.05 * o["iterations"]), (res, o["iterations"])
        xs = x.get()
        true_rel = np.linalg.norm(bh - oracle.csr_spmv(rp, ci, v, xs)) / np.linalg.norm(bh)
        assert true_rel <= tol * 1.0001 and abs(res["rel_res_norm"] - true_rel) <= 1e-3 * tol
        assert np.linalg.norm(xs - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
        hist = S.residual_history(1000)
        assert hist.size == res["iterations"] + 1
        m = min(hist.size, o["norms"].size, 12)      # BiCGSTAB histories drift apart sooner than CG's
        assert np.allclose(hist[:m], o["norms"][:m], rtol=1e-6)


def test_bicgstab_maxit_restart_and_zero_residual():
    rp, ci, v, n = _convection_diffusion(5, (40, 40, 1))
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.BiCGSTAB(max_iter=4, tol=1e-14)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    o = oracle.bicgstab(rp, ci, v, np.ones(n), tol=1e-14, maxit=4)
    assert res["iterations"] == 4 and res["converged"] == 0
    assert np.isclose(res["rel_res_norm"], o["rel_res_norm"], rtol=1e-8)
    assert np.allclose(x.get(), o["x"], rtol=1e-9, atol=1e-13)
    res2 = S.solve(A, b, x)          # niter > 1 in the reference re-solves from the previous x (solvers.c:541-542)
    o2 = oracle.bicgstab(rp, ci, v, np.ones(n), x0=o["x"], tol=1e-14, maxit=4)
    assert np.isclose(res2["rel_res_norm"], o2["rel_res_norm"], rtol=1e-7)
    # the exact solution as start: zero residual, zero iterations
    xs = np.linalg.solve(_dense(rp, ci, v, n), np.ones(n)) if n <= 2000 else None
    b0, x0 = mk.Vector(n, fill=0.0), mk.Vector(n, fill=0.0)
    r0 = S.solve(A, b0, x0)
    assert r0["iterations"] == 0 and np.all(x0.get() == 0.0)


def _dense(rp, ci, v, n):
    M = np.zeros((n, n))
    rows = np.repeat(np.arange(n), np.diff(rp))
    np.add.at(M, (rows, ci), v)
    return M


# ------------------------------------------------------------------ CGNR (solver id 40)
@pytest.mark.parametrize("kind,dims,kernel", [(5, (40, 32, 1), None), (7, (14, 12, 10), None), (7, (32, 16, 8), "sweep")])
def test_cgnr_matches_oracle(kind, dims, kernel):
    rp, ci, v, n = _convection_diffusion(kind, dims)
    bh = np.random.default_rng(12).normal(size=n)
    tol = 1e-8
    o = oracle.cgnr(rp, ci, v, bh, tol=tol, maxit=5000, want_norms=True)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    if kernel == "sweep":
        A.set_spmv_kernel(mk.SPMV_SWEEP)
    for check_every in (1, 16):
        b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
        S = mk.CGNR(max_iter=5000, tol=tol, logging=1, check_every=check_every)
        S.setup(A, b, x)
        res = S.solve(A, b, x)
        assert res["converged"] == 1 == o["converged"]
        assert abs(res["iterations"] - o["iterations"]) <= max(CG_ITER_TOL, 0.02 * o["iterations"]), (res, o["iterations"])
        xs = x.get()
        true_rel = np.linalg.norm(bh - oracle.csr_spmv(rp, ci, v, xs)) / np.linalg.norm(bh)
        assert true_rel < tol and abs(res["rel_res_norm"] - true_rel) <= 1e-3 * tol
        assert np.linalg.norm(xs - o["x"]) <= 1e-5 * np.linalg.norm(o["x"])
        hist = S.residual_history(6000)
        assert hist.size == res["iterations"] + 1
        m = min(hist.size, o["rel_norms"].size, 30)
        assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)


def test_cgnr_transpose_maxit_and_zero_rhs():
    # a matrix that is far from symmetric: the transpose built at setup must be the real one
    rng = np.random.default_rng(13)
    n = 3000
    lens = rng.integers(1, 7, n)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = rng.integers(0, n, rp[-1]).astype(np.int32)
    v = rng.normal(size=rp[-1])
    rows = np.repeat(np.arange(n), lens)
    # make it well conditioned: strong diagonal appended to every row
    rp2 = rp + np.arange(n + 1, dtype=np.int32)
    ci2 = np.insert(ci, rp[1:], np.arange(n)).astype(np.int32)
    v2 = np.insert(v, rp[1:], 8.0 + rng.uniform(0, 1, n))
    A = mk.Matrix.from_csr(n, n, rp2, ci2, v2)
    bh = rng.normal(size=n)
    b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
    S = mk.CGNR(max_iter=6, tol=1e-14)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    o = oracle.cgnr(rp2, ci2, v2, bh, tol=1e-14, maxit=6)
    assert res["iterations"] == 6 and res["converged"] == 0
    assert np.isclose(res["rel_res_norm"], o["rel_res_norm"], rtol=1e-9)
    assert np.allclose(x.get(), o["x"], rtol=1e-9, atol=1e-13)
    b0, x0 = mk.Vector(n, fill=0.0), mk.Vector(n, fill=2.0)
    r0 = S.solve(A, b0, x0)
    assert r0["iterations"] == 0 and np.all(x0.get() == 0.0)


# ------------------------------------------------------------------ GMRES(k) (solver id 12)
@pytest.mark.parametrize("kind,dims,k_dim,kernel", [(5, (48, 40, 1), 30, None), (7, (16, 14, 12), 30, None),
                                                    (7, (16, 14, 12), 5, None), (7, (32, 16, 10), 30, "sweep")])
def test_gmres_matches_oracle(kind, dims, k_dim, kernel):
    rp, ci, v, n = _convection_diffusion(kind, dims)
    bh = np.random.default_rng(14).normal(size=n)
    tol = 1e-9
    o = oracle.gmres(rp, ci, v, bh, tol=tol, maxit=2000, k_dim=k_dim, want_norms=True)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    if kernel == "sweep":
        A.set_spmv_kernel(mk.SPMV_SWEEP)
    b, x = mk.Vector(n, values=bh), mk.Vector(n, fill=0.0)
    S = mk.GMRES(max_iter=2000, tol=tol, k_dim=k_dim, logging=1)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    assert res["converged"] == 1 == o["converged"]
    assert abs(res["iterations"] - o["iterations"]) <= max(CG_ITER_TOL, 0.02 * o["iterations"]), (res, o["iterations"])
    xs = x.get()
    true_rel = np.linalg.norm(bh - oracle.csr_spmv(rp, ci, v, xs)) / np.linalg.norm(bh)
    assert true_rel <= tol * 1.0001 and abs(res["rel_res_norm"] - true_rel) <= 1e-3 * tol
    assert np.linalg.norm(xs - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
    hist = S.residual_history(4000)
    assert hist.size == res["iterations"] + 1
    m = min(hist.size, o["norms"].size, 2 * k_dim + 5)      # across at least one restart
    assert np.allclose(hist[:m], o["norms"][:m], rtol=1e-7)


def test_gmres_maxit_restart_and_zero_residual():
    rp, ci, v, n = _convection_diffusion(5, (40, 40, 1))
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.GMRES(max_iter=7, tol=1e-14, k_dim=3)           # 2 full cycles + 1 iteration
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    o = oracle.gmres(rp, ci, v, np.ones(n), tol=1e-14, maxit=7, k_dim=3)
    assert res["iterations"] == 7 and res["converged"] == 0
    assert np.isclose(res["rel_res_norm"], o["rel_res_norm"], rtol=1e-9)
    assert np.allclose(x.get(), o["x"], rtol=1e-9, atol=1e-13)
    res2 = S.solve(A, b, x)                                 # niter > 1: continue from the previous x
    o2 = oracle.gmres(rp, ci, v, np.ones(n), x0=o["x"], tol=1e-14, maxit=7, k_dim=3)
    assert np.isclose(res2["rel_res_norm"], o2["rel_res_norm"], rtol=1e-8)
    b0, x0 = mk.Vector(n, fill=0.0), mk.Vector(n, fill=0.0)
    r0 = S.solve(A, b0, x0)                                 # zero residual at the start: nothing to do
    assert r0["iterations"] == 0 and np.all(x0.get() == 0.0)
    with pytest.raises(mk.MkError):
        mk.GMRES(k_dim=100)


@pytest.mark.parametrize("check_every", [1, 5])
def test_jacobi_banded_matches_oracle(check_every):
    # config 5 shape scaled down
    n = 20000
    rp, ci, v = oracle.banded_csr(n, 3)
    o = oracle.jacobi(rp, ci, v, np.ones(n), tol=1e-9, maxit=400, check_every=check_every)
    A = mk.Matrix.banded(n, 3)
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.Jacobi(max_iter=400, tol=1e-9, check_every=check_every)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    assert res["converged"] == 1 == o["converged"]
    assert res["iterations"] == o["iterations"]
    assert np.isclose(res["rel_res_norm"], o["rel_res_norm"], rtol=1e-9)
    assert np.allclose(x.get(), o["x"], rtol=1e-12, atol=0)


def test_jacobi_hits_maxit():
    rp, ci, v = oracle.stencil_csr(5, 30, 30)   # weakly dominant: converges slowly
    n = rp.size - 1
    o = oracle.jacobi(rp, ci, v, np.ones(n), tol=1e-12, maxit=37, check_every=10)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.Jacobi(max_iter=37, tol=1e-12, check_every=10)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    assert res["converged"] == 0 and res["iterations"] == 37 == o["iterations"]
    assert np.allclose(x.get(), o["x"], rtol=1e-12)


# ------------------------------------------------------------------ full-size properties
def test_full_size_config2_properties():
    """BASELINE config 2 at full size (7-pt 256^3): the oracle would take minutes, so check
    size-independent properties: A*1 = boundary deficit (row sums), symmetry <Ax,y> = <x,Ay>,
    linearity, and that CG reduces the true residual."""
    nx = 256
    A = mk.Matrix.stencil(7, nx, nx, nx)
    n = nx ** 3
    assert A.nnz == 7 * n - 6 * nx * nx
    ones, y = mk.Vector(n, fill=1.0), mk.Vector(n)
    ytx, xtx = A.spmv_dot(ones, y)
    # sum of all row sums = 6n - (number of neighbour links) = 6 nx^2 missing links
    assert ytx == 6.0 * nx * nx and xtx == float(n)
    yh = y.get()
    assert yh.min() == 0.0 and yh.max() == 3.0 and yh[0] == 3.0 and yh[n // 2 + nx // 2 * nx + nx // 2] == 0.0
    rng = np.random.default_rng(0)
    u, w = rng.normal(size=n), rng.normal(size=n)
    U, W, AU, AW = mk.Vector(n, values=u), mk.Vector(n, values=w), mk.Vector(n), mk.Vector(n)
    A.spmv(U, AU)
    A.spmv(W, AW)
    au, aw = AU.get(), AW.get()
    assert abs(au @ w - u @ aw) <= 1e-11 * (np.abs(au) @ np.abs(w))
    A.spmv(U, AW, alpha=2.0, beta=-1.0)          # AW <- 2 A u - A w
    assert np.allclose(AW.get(), 2.0 * au - aw, rtol=1e-13, atol=1e-12)
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=50, tol=0.0, precond=1)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    assert res["iterations"] == 50
    A.spmv(x, y)
    true_rel = np.linalg.norm(1.0 - y.get()) / np.sqrt(n)
    assert np.isclose(true_rel, res["rel_res_norm"], rtol=1e-6)


def test_full_size_config3_conversion_is_bit_exact_and_product_matches():
    """BASELINE config 3 at full size (R-MAT scale 24, 2^28 unsorted entries, duplicates kept): the oracle's counting sort
    takes a few seconds at this size, so the device conversion is compared with it bit for bit (row pointers, columns in
    the reference's stable order, value bits), then one product on the 2^24 x 2^24 matrix against the oracle's."""
    scale, ne = 24, 1 << 28
    d = mk.DeviceCOO.rmat(scale, ne, seed=42)
    nr, nc, r, c, v = d.download()
    assert nr == 1 << scale and r.size == ne
    A = mk.Matrix.from_device_coo(d)
    del d
    rp, ci, val = A.to_csr()
    orp, oci, ov = oracle.coo_to_csr(nr, r, c, v)
    del r, c, v
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and _same_bits(val, ov)
    del rp, ci, val
    info = A.info()
    assert info.kernel == mk.SPMV_SPLIT and info.max_row_len > 100000
    xh = np.random.default_rng(3).normal(size=nc)
    x, y = mk.Vector(nc, values=xh), mk.Vector(nr)
    A.spmv(x, y)
    got = y.get()
    ref = oracle.csr_spmv(orp, oci, ov, xh)
    mag = oracle.csr_spmv(orp, oci, np.abs(ov), np.abs(xh))
    assert (np.abs(got - ref) / np.maximum(mag, 1e-300)).max() <= SPMV_RTOL
    assert np.count_nonzero(got) == np.count_nonzero(np.diff(orp))      # empty rows give exact zeros, others do not cancel
    mk.memory_trim()


def test_full_size_config4_rank_share_row_sums():
    """BASELINE config 4, the share of rank 0 of 8 at full size (27-point 512^3, first 64 planes, global column numbers):
    A * 1 = 26 - (neighbours present), summed over the share = 27 rows - (sum over rows of the 3x3x3 box sizes)"""
    nx, planes = 512, 64
    rows = nx * nx * planes
    A = mk.Matrix.stencil(27, nx, nx, nx, row_begin=0, row_end=rows)
    assert A.nrows == rows and A.ncols == nx ** 3
    box = (3 * nx - 2) * (3 * nx - 2) * (3 * planes - 1)      # rank 0 owns the z = 0 boundary only
    assert A.nnz == box
    ones, y = mk.Vector(nx ** 3, fill=1.0), mk.Vector(rows)
    A.spmv(ones, y)
    yh = y.get()
    assert yh.sum() == 27.0 * rows - box
    assert yh.min() == 0.0 and yh[0] == 26.0 - 7.0 and yh[nx * nx + nx + 1] == 0.0
    del A, ones, y
    mk.memory_trim()


def test_full_size_config5_banded_jacobi_properties():
    """BASELINE config 5 at full size (banded, half bandwidth 3, 2^26 rows, plain Jacobi): rows further than 3 k from both
    ends see the same arithmetic whatever n is, so after k sweeps the ends and the interior must equal those of a small
    oracle run, and the residual norm follows from the small run's per-row residuals."""
    n, hb, k, m = 1 << 26, 3, 20, 4096
    rp, ci, v = oracle.banded_csr(m, hb)
    o = oracle.jacobi(rp, ci, v, np.ones(m), tol=0.0, maxit=k, check_every=10)
    rp2, ci2, v2 = oracle.banded_csr(2 * m, hb)
    o2 = oracle.jacobi(rp2, ci2, v2, np.ones(2 * m), tol=0.0, maxit=k, check_every=10)
    xs = o["x"]
    A = mk.Matrix.banded(n, hb)
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.Jacobi(max_iter=k, tol=0.0, check_every=10)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    assert res["iterations"] == k == o["iterations"] and res["converged"] == 0
    xf = x.get()
    edge = 1024                                   # > hb * k: the ends' influence has not reached row `edge`
    assert np.allclose(xf[:edge], xs[:edge], rtol=1e-12, atol=0) and np.allclose(xf[-edge:], xs[-edge:], rtol=1e-12, atol=0)
    mid = xf[edge:n - edge]
    assert mid.min() == mid.max() and np.isclose(mid[0], xs[m // 2], rtol=1e-12, atol=0)
    # ||r||^2 = (ends' part) + rows * rho^2 with the same ends' part and interior rho at every n: two small runs give both
    q1, q2 = o["rel_res_norm"] ** 2 * m, o2["rel_res_norm"] ** 2 * 2 * m
    rho2 = (q2 - q1) / m
    expect = np.sqrt((q1 - m * rho2 + n * rho2) / n)
    assert np.isclose(res["rel_res_norm"], expect, rtol=1e-9)
    del A, b, x
    mk.memory_trim()


@pytest.mark.parametrize("seed", [None, 3])
def test_full_size_target_512_cubed_cg_properties(seed):
    """the target line's operator at full size (7-point 512^3; seed 3: the variable-coefficient form D A D the driver-run
    roofline is measured on): row sums / symmetry, and CG's recurrence residual equals the true residual b - A x"""
    nx = 512
    n = nx ** 3
    A = mk.Matrix.stencil(7, nx, nx, nx, scale_seed=seed)
    assert A.nnz == 7 * n - 6 * nx * nx
    assert A.info().kernel == (mk.SPMV_SWEEP if seed is None else mk.SPMV_VSWEEP)
    ones, y = mk.Vector(n, fill=1.0), mk.Vector(n)
    if seed is None:
        ytx, xtx = A.spmv_dot(ones, y)
        assert ytx == 6.0 * nx * nx and xtx == float(n)
    rng = np.random.default_rng(1)
    u = rng.normal(size=n)
    U, AU = mk.Vector(n, values=u), mk.Vector(n)
    A.spmv(U, AU)
    au = AU.get()
    A.spmv(ones, y)
    a1 = y.get()
    assert abs(au.sum() - u @ a1) <= 1e-11 * (np.abs(u) @ np.abs(a1) + np.abs(au).sum())      # <A u, 1> = <u, A 1>
    del U, AU, au, a1, u
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=30, tol=0.0, precond=0)
    S.setup(A, b, x)
    res = S.solve(A, b, x)
    assert res["iterations"] == 30
    A.spmv(x, y)
    true_rel = np.linalg.norm(1.0 - y.get()) / np.sqrt(n)
    assert np.isclose(true_rel, res["rel_res_norm"], rtol=1e-6)
    del A, b, x, y, ones, S
    mk.memory_trim()


def test_allocator_keeps_and_reuses_large_blocks():
    """freed device arrays of 1 MB and more wait in the library's free list and serve the next request of their size;
    mk_memory_trim() returns them to the driver (include/mk_krylov.h mk_memory_info / mk_memory_trim)"""
    if os.environ.get("MK_NO_ALLOC_CACHE") or os.environ.get("MK_ALLOC_CACHE_MB"):
        pytest.skip("the free list is switched off or resized by the environment")
    mk.init(0)
    mk.memory_trim()
    n = 3 * (1 << 20)                                   # 24 MB of doubles
    live0, idle0 = mk.memory_info()
    assert idle0 == 0
    v = mk.Vector(n, fill=1.5)
    live1, idle1 = mk.memory_info()
    assert live1 - live0 >= 8 * n and idle1 == 0
    del v
    live2, idle2 = mk.memory_info()
    assert live2 == live0 and idle2 >= 8 * n            # kept, not freed
    w = mk.Vector(n, fill=2.5)                          # same size: the kept block is handed out again
    live3, idle3 = mk.memory_info()
    assert idle3 == 0 and live3 == live1
    assert np.all(w.get() == 2.5)
    small = mk.Vector(1000, fill=1.0)                   # below 1 MB: never kept
    del small
    assert mk.memory_info()[1] == 0
    del w
    assert mk.memory_info()[1] >= 8 * n
    mk.memory_trim()
    assert mk.memory_info() == (live0, 0)


_RELABEL_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import oracle
import iterative_solvers_amd as mk
mk.init(0)
for scale, ne in ((16, 1 << 20), (18, 1 << 21)):
    r, c, v = oracle.rmat_coo(scale, ne, seed=scale)
    n = 1 << scale
    rp, ci, vv = oracle.coo_to_csr(n, r, c, v)
    A = mk.Matrix.from_device_coo(mk.DeviceCOO.from_host(n, n, r, c, v))
    A.set_spmv_kernel(mk.SPMV_SPLIT)
    xh = np.random.default_rng(scale).normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    A.spmv(x, y)               # first product: builds the column relabelling, with MK_RELABEL_CHECK=1 it verifies its degrees
    ref = oracle.csr_spmv(rp, ci, vv, xh)
    scale_y = oracle.csr_spmv(rp, ci, np.abs(vv), np.abs(xh))
    assert np.all(np.abs(y.get() - ref) <= 1e-12 * np.maximum(scale_y, 1e-300))
print("relabel ok")
"""


def test_column_degrees_per_xcd_match_device_scope_count():
    """the column relabelling counts column degrees with workgroup-scope atomics into one copy per XCD (picked by the
    hardware's XCC id); MK_RELABEL_CHECK=1 makes the library recount with device-scope atomics and fail on any difference.
    Child process: MK_RELABEL_MIN_COLS is read once per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MK_RELABEL_CHECK="1", MK_RELABEL_MIN_COLS="1000")
    out = subprocess.run([sys.executable, "-c", _RELABEL_SNIPPET, root], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "relabel ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
