"""Pins the oracle (oracle/oracle.c) to the reference: every fixture under tests/golden/ was
produced by the reference's own sparse_matrix.c (see tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, GOLDEN_CASES


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_sort_coo_row_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sr, sc, sv = oracle.sort_coo_row(int(g["nrows"]), g["coo_rows"], g["coo_cols"], g["coo_vals"])
    assert np.array_equal(sr, g["sorted_rows"])
    assert np.array_equal(sc, g["sorted_cols"])
    assert np.array_equal(sv.view(np.uint64), g["sorted_vals"].view(np.uint64))


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_coo_to_csr_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rp, ci, v = oracle.coo_to_csr(int(g["nrows"]), g["coo_rows"], g["coo_cols"], g["coo_vals"])
    assert np.array_equal(rp, g["csr_rowptr"])
    assert np.array_equal(ci, g["csr_col"])
    assert np.array_equal(v.view(np.uint64), g["csr_val"].view(np.uint64))


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (no /root/reference)")
def test_oracle_vs_live_reference_random():
    rng = np.random.default_rng(7)
    for nrows, nnz in [(1, 5), (17, 0), (100, 1000), (5000, 20000), (300000, 100000)]:
        rows = rng.integers(0, nrows, nnz).astype(np.int32)
        cols = rng.integers(0, nrows, nnz).astype(np.int32)
        vals = rng.normal(size=nnz)
        a = oracle.sort_coo_row(nrows, rows, cols, vals)
        b = oracle.ref_sort_coo_row(nrows, nrows, rows, cols, vals)
        for u, w in zip(a, b):
            assert np.array_equal(u, w)
        a = oracle.coo_to_csr(nrows, rows, cols, vals)
        b = oracle.ref_coo_to_csr(nrows, nrows, rows, cols, vals)
        for u, w in zip(a, b):
            assert np.array_equal(u, w)


def test_spmv_known_answer():
    # the only SpMV input the reference holds: test/sblas_mv_rd.c:15-32
    k = np.load(os.path.join(GOLDEN, "sblas_mv_kat.npz"))
    y = oracle.csr_spmv(k["iaptr"], k["iaind"], k["aval"], k["x"])
    assert np.allclose(y, [3.5, 11.3, 66.9, 21.7, 77.0], rtol=1e-14)
    assert np.allclose(y, k["y"], rtol=1e-15)


def test_pcg_against_scipy():
    """PARITY UNPINNED (HYPRE absent): cross-check only.  scipy's cg stops on
    ||r|| <= rtol*||b||, HYPRE on ||r||^2/||b||^2 < tol^2 (+ rel_change), so iteration counts
    may differ by the rel_change extra iterations; the solution must agree."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    rp, ci, v = oracle.stencil_csr(5, 32, 32)
    A = sp.csr_matrix((v, ci, rp))
    b = np.ones(A.shape[0])
    its = [0]
    xs, info = spla.cg(A, b, rtol=1e-8, atol=0.0, maxiter=1000, callback=lambda _: its.__setitem__(0, its[0] + 1))
    assert info == 0
    o = oracle.pcg(rp, ci, v, b, tol=1e-8, rel_change=0)
    assert o["converged"] == 1
    assert abs(o["iterations"] - its[0]) <= 2
    assert np.linalg.norm(o["x"] - xs) / np.linalg.norm(xs) < 1e-7
    assert np.linalg.norm(b - A @ o["x"]) / np.linalg.norm(b) < 1e-8 * 1.01
    o2 = oracle.pcg(rp, ci, v, b, tol=1e-8, rel_change=1)
    assert o2["iterations"] >= o["iterations"]
    oj = oracle.pcg(rp, ci, v, b, tol=1e-8, precond=1)
    assert oj["converged"] == 1
    assert np.linalg.norm(oj["x"] - xs) / np.linalg.norm(xs) < 1e-7


def test_jacobi_converges_banded():
    rp, ci, v = oracle.banded_csr(2000, 3)
    b = np.ones(2000)
    o = oracle.jacobi(rp, ci, v, b, tol=1e-10, maxit=500, check_every=5)
    import scipy.sparse as sp
    A = sp.csr_matrix((v, ci, rp))
    assert o["converged"] == 1 and o["iterations"] % 5 == 0
    assert np.linalg.norm(b - A @ o["x"]) / np.linalg.norm(b) < 1e-10


def test_partition_rows():
    rp, _, _ = oracle.stencil_csr(7, 8, 8, 8)
    for npart in (1, 2, 3, 8):
        rs = oracle.partition_rows(rp, npart)
        assert rs[0] == 0 and rs[-1] == 512 and np.all(np.diff(rs) >= 0)
        share = np.diff(rp[rs].astype(np.int64))
        assert share.sum() == rp[-1]
        assert share.max() - share.min() <= 2 * 7 * npart
    # empty rows and a part that would start past the end
    rp = np.array([0, 0, 0, 5, 5, 5, 6, 6], np.int64)
    rs = oracle.partition_rows(rp, 4)
    assert rs[0] == 0 and rs[-1] == 7 and np.all(np.diff(rs) >= 0)


def test_generators_shapes():
    rp, ci, v = oracle.stencil_csr(27, 5, 4, 3)
    assert rp[-1] == (3 * 5 - 2) * (3 * 4 - 2) * (3 * 3 - 2)
    rp, ci, v = oracle.stencil_csr(7, 16, 16, 16)
    assert rp[-1] == 7 * 4096 - 6 * 256
    r, c, vv = oracle.rmat_coo(10, 1 << 14)
    assert r.max() < 1024 and c.max() < 1024 and np.all(np.abs(vv) <= 1)
    assert np.bincount(r, minlength=1024)[0] > (1 << 14) * 0.76 ** 10 * 0.7
