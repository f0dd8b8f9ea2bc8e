"""Distributed path on ONE GPU: several logical ranks in one process ("local world"): same
partition / ghost map / diag-offd split / solver code as over RCCL, with the exchange done by
device copies.  Checked against the global oracle results."""
import os

import numpy as np
import pytest

import iterative_solvers_amd as mk
from iterative_solvers_amd import distributed as mkd
import oracle

pytestmark = pytest.mark.gpu


def _blocks(rp, ci, v, n, rs, comms):
    As = []
    for r, cm in enumerate(comms):
        lo, hi = int(rs[r]), int(rs[r + 1])
        blk = mk.Matrix.from_csr(hi - lo, n, rp[lo:hi + 1] - rp[lo], ci[rp[lo]:rp[hi]], v[rp[lo]:rp[hi]])
        As.append(mkd.DistMatrix(blk, n, rs, cm))
    return As


@pytest.mark.parametrize("world,cuts", [(2, None), (3, [0, 100, 1500, 1728]), (4, [0, 432, 432, 1000, 1728])])
def test_dist_spmv_stencil(world, cuts):
    rp, ci, v = oracle.stencil_csr(7, 12, 12, 12)
    n = rp.size - 1
    rng = np.random.default_rng(world)
    v = v * rng.uniform(0.5, 1.5, v.size)
    rs = np.array(cuts, np.int64) if cuts else mkd.slab_row_start(12, 12, 12, world)
    comms = mkd.Comm.local_world(world)
    As = _blocks(rp, ci, v, n, rs, comms)
    xh = rng.normal(size=n)
    xs = [A.vector(values=xh[rs[r]:rs[r + 1]]) for r, A in enumerate(As)]
    ys = [A.vector(fill=-7.0) for A in As]
    mkd.dist_spmv(As, xs, ys)
    got = np.concatenate([y.get() for y in ys])
    ref = oracle.csr_spmv(rp, ci, v, xh)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    i0 = As[0].info()
    assert i0["n_local"] == rs[1] and i0["nnz_diag"] + i0["nnz_offd"] == rp[rs[1]]
    if cuts is None and world == 2:
        assert i0["n_ghost"] == 144 and i0["n_offd_rows"] == 144 and i0["n_neighbors"] == 1


def test_dist_spmv_all_to_all_rmat():
    scale, ne, world = 11, 1 << 15, 4
    r, c, v = oracle.rmat_coo(scale, ne, seed=7)
    n = 1 << scale
    rp, ci, val = oracle.coo_to_csr(n, r, c, v)
    rs = mk.partition_rows(rp, world)                 # nnz-balanced blocks (solvers.c:214-231)
    comms = mkd.Comm.local_world(world)
    As = _blocks(rp, ci, val, n, rs, comms)
    assert max(A.info()["n_neighbors"] for A in As) == world - 1
    xh = np.random.default_rng(3).normal(size=n)
    xs = [A.vector(values=xh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    ys = [A.vector() for A in As]
    mkd.dist_spmv(As, xs, ys)
    got = np.concatenate([y.get() for y in ys])
    ref = oracle.csr_spmv(rp, ci, val, xh)
    mag = oracle.csr_spmv(rp, ci, np.abs(val), np.abs(xh))
    assert (np.abs(got - ref) / np.maximum(mag, 1e-300)).max() <= 1e-12


@pytest.mark.parametrize("world", [2, 3])
def test_dist_node_block_matrix_spmv_and_pcg(world):
    """a finite-element-like matrix (three unknowns per node, random neighbour nodes, symmetric, dominant diagonal) cut into
    nnz-balanced row blocks: the diag blocks take the stream kernel with shared column lists, the cuts fall inside nodes;
    product and Jacobi-preconditioned CG against the global oracle"""
    rng = np.random.default_rng(40 + world)
    nodes = 1500
    pairs = set()
    for i in range(nodes):
        pairs.add((i, i))
        for j in rng.integers(max(0, i - 60), min(nodes, i + 61), int(rng.integers(5, 14))):
            pairs.add((i, int(j))); pairs.add((int(j), i))
    pi, pj = np.array(sorted(pairs)).T
    # node pair -> 3x3 block, symmetric values
    w = {}
    for a, b in zip(pi, pj):
        if (b, a) in w:
            w[(a, b)] = w[(b, a)].T
        else:
            blk = -rng.uniform(0.1, 1.0, (3, 3))
            w[(a, b)] = (blk + blk.T) / 2 if a == b else blk
    rows, cols, vals = [], [], []
    for a, b in zip(pi, pj):
        for d in range(3):
            for e in range(3):
                rows.append(3 * a + d); cols.append(3 * b + e); vals.append(w[(a, b)][d, e])
    rows, cols, vals = np.array(rows), np.array(cols), np.array(vals)
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order].astype(np.int32), vals[order]
    n = 3 * nodes
    absum = np.bincount(rows, weights=np.abs(vals), minlength=n)
    vals[rows == cols] = 1.2 * absum[rows[rows == cols]]
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    rs = mk.partition_rows(rp, world)
    comms = mkd.Comm.local_world(world)
    As = _blocks(rp, cols, vals, n, rs, comms)
    xh = rng.normal(size=n)
    xs = [A.vector(values=xh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    ys = [A.vector() for A in As]
    mkd.dist_spmv(As, xs, ys)
    got = np.concatenate([y.get() for y in ys])
    ref = oracle.csr_spmv(rp, cols, vals, xh)
    mag = oracle.csr_spmv(rp, cols, np.abs(vals), np.abs(xh))
    assert (np.abs(got - ref) / np.maximum(mag, 1e-300)).max() <= 1e-12
    o = oracle.pcg(rp, cols, vals, np.ones(n), tol=1e-9, precond=1, maxit=500, want_norms=True)
    bs = [A.vector(fill=1.0) for A in As]
    x0 = [A.vector(fill=0.0) for A in As]
    Ss = [mk.PCG(max_iter=500, tol=1e-9, precond=1, logging=1, check_every=7) for _ in As]
    res = mkd.dist_pcg_solve(Ss, As, bs, x0)
    assert res[0]["converged"] == 1 and abs(res[0]["iterations"] - o["iterations"]) <= 2
    xg = np.concatenate([x.get() for x in x0])
    assert np.linalg.norm(xg - o["x"]) <= 1e-7 * np.linalg.norm(o["x"])


def test_dist_banded_matrix_through_the_band_kernel():
    """a banded symmetric positive definite matrix without shared offsets or column lists over three row blocks: the diag
    blocks take the band kernel (sliding window of x in LDS, band.hip) with the distributed vector's owned part as x; product
    and CG against the global oracle"""
    rng = np.random.default_rng(77)
    n = 36000
    r = np.repeat(np.arange(n), 10)
    off = rng.integers(-700, 701, r.size)
    c = np.where((r + off < 0) | (r + off >= n), r - off, r + off)
    pairs = np.unique(np.concatenate([np.stack([r, c], 1), np.stack([c, r], 1), np.stack([np.arange(n)] * 2, 1)]), axis=0)
    r, c = pairs[:, 0], pairs[:, 1]
    key = np.minimum(r, c) * n + np.maximum(r, c)
    _, inv = np.unique(key, return_inverse=True)
    w = rng.uniform(-1.0, 1.0, inv.max() + 1)[inv]
    absum = np.bincount(r, weights=np.abs(w), minlength=n)
    w[r == c] = 1.2 * absum[r[r == c]] + 1.0
    rp = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))]).astype(np.int32)
    ci = c.astype(np.int32)
    world = 3
    rs = mk.partition_rows(rp, world)
    old = os.environ.get("MK_BAND_MIN_ROWS")
    os.environ["MK_BAND_MIN_ROWS"] = "1000"
    try:
        As = _blocks(rp, ci, w, n, rs, mkd.Comm.local_world(world))
        xh = rng.normal(size=n)
        xs = [A.vector(values=xh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
        ys = [A.vector() for A in As]
        mkd.dist_spmv(As, xs, ys)
        assert all(A.diag_info().band_window > 0 for A in As), [A.diag_info().band_window for A in As]
    finally:
        if old is None:
            del os.environ["MK_BAND_MIN_ROWS"]
        else:
            os.environ["MK_BAND_MIN_ROWS"] = old
    got = np.concatenate([y.get() for y in ys])
    ref = oracle.csr_spmv(rp, ci, w, xh)
    mag = oracle.csr_spmv(rp, ci, np.abs(w), np.abs(xh))
    assert (np.abs(got - ref) / np.maximum(mag, 1e-300)).max() <= 1e-12
    bh = rng.normal(size=n)
    o = oracle.pcg(rp, ci, w, bh, tol=1e-10, precond=1, maxit=500, want_norms=True)
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    x0 = [A.vector(fill=0.0) for A in As]
    Ss = [mk.PCG(max_iter=500, tol=1e-10, precond=1, logging=1) for _ in As]
    res = mkd.dist_pcg_solve(Ss, As, bs, x0)
    assert res[0]["converged"] == 1 and abs(res[0]["iterations"] - o["iterations"]) <= 2
    xg = np.concatenate([x.get() for x in x0])
    assert np.linalg.norm(xg - o["x"]) <= 1e-8 * np.linalg.norm(o["x"])


@pytest.mark.parametrize("world,precond", [(1, 0), (2, 1), (4, 0), (8, 1)])
def test_dist_pcg_matches_oracle(world, precond):
    rp, ci, v = oracle.stencil_csr(7, 16, 16, 24)
    n = rp.size - 1
    rng = np.random.default_rng(5)
    d = rng.uniform(1.0, 20.0, n)
    rows = np.repeat(np.arange(n), np.diff(rp))
    v = v * np.sqrt(d[rows] * d[ci])
    bh = rng.normal(size=n)
    o = oracle.pcg(rp, ci, v, bh, tol=1e-9, precond=precond, maxit=3000, want_norms=True)
    rs = mkd.slab_row_start(16, 16, 24, world)
    comms = mkd.Comm.local_world(world) if world > 1 else [None]
    As = _blocks(rp, ci, v, n, rs, comms)
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.PCG(max_iter=3000, tol=1e-9, precond=precond, logging=1, check_every=9) for _ in As]
    res = mkd.dist_pcg_solve(Ss, As, bs, xs)
    assert all(q == res[0] for q in res)
    assert res[0]["converged"] == 1 and abs(res[0]["iterations"] - o["iterations"]) <= 2
    xg = np.concatenate([x.get() for x in xs])
    assert np.linalg.norm(xg - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
    hist = Ss[0].residual_history(4000)
    m = min(hist.size, o["rel_norms"].size, 30)
    assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)


@pytest.mark.parametrize("world,precond", [(1, 0), (2, 1), (3, 0), (4, 1)])
def test_dist_pcg_fused_sweep_step(world, precond):
    """constant-coefficient grid: the diag blocks run on the sweep kernel and the solver takes the fused step
    (A p recomputed, off-rank part applied from a compact buffer, x updated inside the residual sweep)"""
    dims = (32, 16, 24)
    rp, ci, v = oracle.stencil_csr(7, *dims)
    n = rp.size - 1
    bh = np.random.default_rng(8).normal(size=n)
    o = oracle.pcg(rp, ci, v, bh, tol=1e-9, precond=precond, maxit=2000, want_norms=True)
    rs = mkd.slab_row_start(*dims, world)
    comms = mkd.Comm.local_world(world) if world > 1 else [None]
    As = _blocks(rp, ci, v, n, rs, comms)
    for A in As:
        A.set_spmv_kernel(mk.SPMV_SWEEP)          # small blocks would default to the pattern kernel
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.PCG(max_iter=2000, tol=1e-9, precond=precond, logging=1, check_every=9) for _ in As]
    res = mkd.dist_pcg_solve(Ss, As, bs, xs)
    assert all(q == res[0] for q in res)
    assert res[0]["converged"] == 1 and abs(res[0]["iterations"] - o["iterations"]) <= 2
    xg = np.concatenate([x.get() for x in xs])
    assert np.linalg.norm(xg - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
    assert np.linalg.norm(bh - oracle.csr_spmv(rp, ci, v, xg)) <= 1.5e-9 * np.linalg.norm(bh)
    hist = Ss[0].residual_history(4000)
    m = min(hist.size, o["rel_norms"].size, 30)
    assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)
    # exactly k iterations
    xs2 = [A.vector(fill=0.0) for A in As]
    S2 = [mk.PCG(max_iter=3, tol=1e-30, precond=precond) for _ in As]
    r2 = mkd.dist_pcg_solve(S2, As, bs, xs2)
    o2 = oracle.pcg(rp, ci, v, bh, tol=1e-30, precond=precond, maxit=3)
    assert r2[0]["iterations"] == 3
    assert np.allclose(np.concatenate([x.get() for x in xs2]), o2["x"], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("kind,dims,world,precond", [(7, (32, 16, 24), 1, 0), (7, (32, 16, 24), 2, 1), (7, (32, 16, 24), 3, 0), (7, (32, 16, 24), 4, 1),
                                                     (27, (16, 18, 24), 2, 0), (27, (16, 18, 24), 8, 1), (7, (64, 8, 33), 5, 0)])
def test_dist_pcg_single_reduction_step(kind, dims, world, precond):
    """mk_pcg_set_single_reduction: ONE all-reduce per iteration (alpha by the Chronopoulos-Gear identity, both halves of the
    step in sweep MODE 6, r exchanged instead of p).  Same iterates as the two-reduction loop of hypre_PCGSolve in exact
    arithmetic: iteration counts within 2, history to 1e-8, exact-k solutions to 1e-9 against orc_pcg; uneven slabs and an
    odd number of planes included."""
    rp, ci, v = oracle.stencil_csr(kind, *dims)
    n = rp.size - 1
    bh = np.random.default_rng(11 + world).normal(size=n)
    o = oracle.pcg(rp, ci, v, bh, tol=1e-9, precond=precond, maxit=2000, want_norms=True)
    rs = mkd.slab_row_start(*dims, world)
    comms = mkd.Comm.local_world(world) if world > 1 else [None]
    As = _blocks(rp, ci, v, n, rs, comms)
    for A in As:
        A.set_spmv_kernel(mk.SPMV_SWEEP)
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    for check_every in (9, 1):
        xs = [A.vector(fill=0.0) for A in As]
        Ss = [mk.PCG(max_iter=2000, tol=1e-9, precond=precond, logging=1, check_every=check_every) for _ in As]
        for S in Ss:
            S.set_single_reduction(True)
        res = mkd.dist_pcg_solve(Ss, As, bs, xs)
        assert all(S.single_reduction_used() for S in Ss)
        assert all(q == res[0] for q in res)
        assert res[0]["converged"] == 1 and abs(res[0]["iterations"] - o["iterations"]) <= 2
        xg = np.concatenate([x.get() for x in xs])
        assert np.linalg.norm(xg - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
        assert np.linalg.norm(bh - oracle.csr_spmv(rp, ci, v, xg)) <= 1.5e-9 * np.linalg.norm(bh)
        hist = Ss[0].residual_history(4000)
        assert hist.size == res[0]["iterations"] + 1
        m = min(hist.size, o["rel_norms"].size)
        assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)
        assert abs(res[0]["rel_res_norm"] - hist[-1]) <= 1e-14
    # exactly k iterations
    for k in (1, 4):
        xs2 = [A.vector(fill=0.0) for A in As]
        S2 = [mk.PCG(max_iter=k, tol=1e-30, precond=precond) for _ in As]
        for S in S2:
            S.set_single_reduction(True)
        r2 = mkd.dist_pcg_solve(S2, As, bs, xs2)
        o2 = oracle.pcg(rp, ci, v, bh, tol=1e-30, precond=precond, maxit=k)
        assert r2[0]["iterations"] == k and r2[0]["converged"] == 0
        assert np.allclose(np.concatenate([x.get() for x in xs2]), o2["x"], rtol=1e-9, atol=1e-13)
    # a nonzero initial guess and a second solve with the same objects (niter > 1 of solvers.c:753-754 re-solves from x)
    x0 = np.random.default_rng(3).normal(size=n)
    xs3 = [A.vector(values=x0[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    S3 = [mk.PCG(max_iter=2000, tol=1e-8, precond=precond) for _ in As]
    for S in S3:
        S.set_single_reduction(True)
    r3 = mkd.dist_pcg_solve(S3, As, bs, xs3)
    o3 = oracle.pcg(rp, ci, v, bh, tol=1e-8, precond=precond, maxit=2000, x0=x0)
    assert r3[0]["converged"] == 1 and abs(r3[0]["iterations"] - o3["iterations"]) <= 2
    r4 = mkd.dist_pcg_solve(S3, As, bs, xs3)
    assert r4[0]["converged"] == 1 and r4[0]["iterations"] <= 3


def test_dist_pcg_single_reduction_falls_back():
    """variable coefficients (diag blocks not on the sweep kernel): the request is declined, the default step runs"""
    dims = (32, 16, 24)
    rp, ci, v = oracle.stencil_csr(7, *dims, scale_seed=4)
    n = rp.size - 1
    bh = np.random.default_rng(2).normal(size=n)
    o = oracle.pcg(rp, ci, v, bh, tol=1e-9, precond=0, maxit=2000)
    rs = mkd.slab_row_start(*dims, 2)
    As = _blocks(rp, ci, v, n, rs, mkd.Comm.local_world(2))
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.PCG(max_iter=2000, tol=1e-9) for _ in As]
    for S in Ss:
        S.set_single_reduction(True)
    res = mkd.dist_pcg_solve(Ss, As, bs, xs)
    assert not any(S.single_reduction_used() for S in Ss)
    assert res[0]["converged"] == 1 and abs(res[0]["iterations"] - o["iterations"]) <= 2
    # constant coefficients on planes so wide that the sweep plan takes 2048-row chunks (1024 threads: the single-reduction
    # kernel would not fit the register budget): declined as well
    dims = (700, 4, 12)
    rp, ci, v = oracle.stencil_csr(7, *dims)
    n = rp.size - 1
    bh = np.random.default_rng(4).normal(size=n)
    o = oracle.pcg(rp, ci, v, bh, tol=1e-9, precond=0, maxit=2000)
    rs = mkd.slab_row_start(*dims, 2)
    As = _blocks(rp, ci, v, n, rs, mkd.Comm.local_world(2))
    for A in As:
        A.set_spmv_kernel(mk.SPMV_SWEEP)
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.PCG(max_iter=2000, tol=1e-9) for _ in As]
    for S in Ss:
        S.set_single_reduction(True)
    res = mkd.dist_pcg_solve(Ss, As, bs, xs)
    assert not any(S.single_reduction_used() for S in Ss)
    assert res[0]["converged"] == 1 and abs(res[0]["iterations"] - o["iterations"]) <= 2


@pytest.mark.parametrize("world,precond,sweep", [(2, 0, False), (2, 1, True), (4, 0, True), (4, 1, False), (8, 0, True), (8, 1, True)])
def test_dist_27pt_rowblock_cg(world, precond, sweep):
    """BASELINE configs[3] scaled down: 27-point stencil, z-slab row blocks (solvers.c:214-281), CG.  The off-rank rows of
    a slab face hold 9 entries each (two lanes per row in the off-rank kernel); diag blocks on the default kernel and on
    the sweep kernel (fused step)."""
    dims = (16, 18, 24)
    rp, ci, v = oracle.stencil_csr(27, *dims)
    n = rp.size - 1
    rng = np.random.default_rng(27 + world)
    rs = mkd.slab_row_start(*dims, world)
    comms = mkd.Comm.local_world(world)
    As = _blocks(rp, ci, v, n, rs, comms)
    if sweep:
        for A in As:
            A.set_spmv_kernel(mk.SPMV_SWEEP)
    i1 = As[1].info()
    plane = dims[0] * dims[1]
    assert i1["n_ghost"] == (2 if world > 2 else 1) * plane and i1["n_offd_rows"] == i1["n_ghost"]
    assert i1["nnz_offd"] == i1["n_ghost"] // plane * (3 * dims[0] - 2) * (3 * dims[1] - 2)     # 9 per interior face row
    # product
    xh = rng.normal(size=n)
    xs = [A.vector(values=xh[rs[r]:rs[r + 1]]) for r, A in enumerate(As)]
    ys = [A.vector(fill=-3.0) for A in As]
    mkd.dist_spmv(As, xs, ys)
    got = np.concatenate([y.get() for y in ys])
    ref = oracle.csr_spmv(rp, ci, v, xh)
    mag = oracle.csr_spmv(rp, ci, np.abs(v), np.abs(xh))
    assert (np.abs(got - ref) / mag).max() <= 1e-12
    # solve
    bh = rng.normal(size=n)
    o = oracle.pcg(rp, ci, v, bh, tol=1e-9, precond=precond, maxit=2000, want_norms=True)
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.PCG(max_iter=2000, tol=1e-9, precond=precond, logging=1, check_every=7) for _ in As]
    res = mkd.dist_pcg_solve(Ss, As, bs, xs)
    assert all(q == res[0] for q in res)
    assert res[0]["converged"] == 1 and abs(res[0]["iterations"] - o["iterations"]) <= 2
    xg = np.concatenate([x.get() for x in xs])
    assert np.linalg.norm(xg - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
    hist = Ss[0].residual_history(4000)
    m = min(hist.size, o["rel_norms"].size, 30)
    assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)
    # exactly k iterations
    xs2 = [A.vector(fill=0.0) for A in As]
    S2 = [mk.PCG(max_iter=4, tol=1e-30, precond=precond) for _ in As]
    r2 = mkd.dist_pcg_solve(S2, As, bs, xs2)
    o2 = oracle.pcg(rp, ci, v, bh, tol=1e-30, precond=precond, maxit=4)
    assert r2[0]["iterations"] == 4
    assert np.allclose(np.concatenate([x.get() for x in xs2]), o2["x"], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("world,precond", [(2, 0), (3, 1)])
def test_dist_pcg_variable_coefficients_on_vsweep_diag_blocks(world, precond):
    """variable-coefficient grid operator over z-slabs: the diag blocks run MK_SPMV_VSWEEP (values streamed through LDS),
    the solver takes the unfused step (K1 + off-rank product, K2, K3)"""
    dims = (32, 16, 24)
    rp, ci, v = oracle.stencil_csr(7, *dims, scale_seed=4)
    n = rp.size - 1
    rng = np.random.default_rng(world)
    rs = mkd.slab_row_start(*dims, world)
    As = _blocks(rp, ci, v, n, rs, mkd.Comm.local_world(world))
    for A in As:
        A.set_spmv_kernel(mk.SPMV_VSWEEP)
        assert A.diag_info().kernel == mk.SPMV_VSWEEP
    xh = rng.normal(size=n)
    xs = [A.vector(values=xh[rs[r]:rs[r + 1]]) for r, A in enumerate(As)]
    ys = [A.vector(fill=-3.0) for A in As]
    mkd.dist_spmv(As, xs, ys)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    mag = oracle.csr_spmv(rp, ci, np.abs(v), np.abs(xh))
    assert (np.abs(np.concatenate([y.get() for y in ys]) - ref) / mag).max() <= 1e-12
    bh = rng.normal(size=n)
    o = oracle.pcg(rp, ci, v, bh, tol=1e-9, precond=precond, maxit=2000, want_norms=True)
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.PCG(max_iter=2000, tol=1e-9, precond=precond, logging=1, check_every=7) for _ in As]
    res = mkd.dist_pcg_solve(Ss, As, bs, xs)
    assert res[0]["converged"] == 1 and abs(res[0]["iterations"] - o["iterations"]) <= 2
    assert np.linalg.norm(np.concatenate([x.get() for x in xs]) - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
    hist = Ss[0].residual_history(4000)
    m = min(hist.size, o["rel_norms"].size, 30)
    assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)


def test_dist_pcg_rejects_nan_rhs():
    rp, ci, v = oracle.stencil_csr(5, 12, 12)
    n = rp.size - 1
    rs = mkd.even_row_start(n, 2)
    As = _blocks(rp, ci, v, n, rs, mkd.Comm.local_world(2))
    bh = np.ones(n)
    bh[5] = np.nan
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    xs = [A.vector(fill=0.0) for A in As]
    with pytest.raises(mk.MkError, match="NaN or Inf"):
        mkd.dist_pcg_solve([mk.PCG(max_iter=5) for _ in As], As, bs, xs)


def test_dist_pcg_maxit_and_empty_rank():
    rp, ci, v = oracle.stencil_csr(5, 20, 20)
    n = rp.size - 1
    rs = np.array([0, 150, 150, 400], np.int64)       # the middle rank owns nothing
    comms = mkd.Comm.local_world(3)
    As = _blocks(rp, ci, v, n, rs, comms)
    bs = [A.vector(fill=1.0) for A in As]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.PCG(max_iter=7, tol=1e-14) for _ in As]
    res = mkd.dist_pcg_solve(Ss, As, bs, xs)
    o = oracle.pcg(rp, ci, v, np.ones(n), tol=1e-14, maxit=7)
    assert res[0]["iterations"] == 7 and res[0]["converged"] == 0
    assert np.isclose(res[0]["rel_res_norm"], o["rel_res_norm"], rtol=1e-9)
    assert np.allclose(np.concatenate([x.get() for x in xs]), o["x"], rtol=1e-9)


@pytest.mark.parametrize("world", [1, 2, 5])
def test_dist_jacobi_banded_matches_oracle(world):
    # BASELINE config 5 shape scaled down: banded, half bandwidth 3, diagonally dominant
    n = 30000
    rp, ci, v = oracle.banded_csr(n, 3)
    o = oracle.jacobi(rp, ci, v, np.ones(n), tol=1e-9, maxit=400, check_every=10)
    rs = mkd.even_row_start(n, world)
    comms = mkd.Comm.local_world(world) if world > 1 else [None]
    As = _blocks(rp, ci, v, n, rs, comms)
    bs = [A.vector(fill=1.0) for A in As]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.Jacobi(max_iter=400, tol=1e-9, check_every=10) for _ in As]
    res = mkd.dist_jacobi_solve(Ss, As, bs, xs)
    assert res[0]["converged"] == 1 == o["converged"] and res[0]["iterations"] == o["iterations"]
    assert np.isclose(res[0]["rel_res_norm"], o["rel_res_norm"], rtol=1e-9)
    assert np.allclose(np.concatenate([x.get() for x in xs]), o["x"], rtol=1e-12, atol=0)
    if world > 1:
        assert As[0].info()["n_ghost"] == 3 and As[1].info()["n_offd_rows"] in (3, 6)


_SPLIT_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import oracle
import iterative_solvers_amd as mk
from iterative_solvers_amd import distributed as mkd
mk.init(0)
def blocks(rp, ci, v, n, rs, comms):
    out = []
    for r in range(len(rs) - 1):
        lo, hi = int(rs[r]), int(rs[r + 1])
        blk = mk.Matrix.from_csr(hi - lo, n, (rp[lo:hi + 1] - rp[lo]).astype(np.int32), ci[rp[lo]:rp[hi]], v[rp[lo]:rp[hi]])
        out.append(mkd.DistMatrix(blk, n, rs, comms[r]))
    return out
cases = [("banded", 60000, np.array([0, 20000, 41000, 60000], np.int64), 10, 1e-9),      # uneven blocks
         ("banded", 40001, np.array([0, 10000, 20000, 30000, 40001], np.int64), 1, 1e-9),
         ("lap7", (12, 10, 60), None, 5, 1e-9),
         # ADVICE r2: planes of 1408 chunks of 512 rows.  A middle rank's interior launch + two boundary launches would need
         # 3 x 1408 = 4224 slots of the 4096-slot slab: set-up must size the interior launch around the boundary launches
         # (or keep the one-launch sweep), never write past the slab or fail in the loop
         ("lap7", (256, 2816, 30), None, 5, 1e-2)]
for kind, shape, rs, every, tol in cases:
    if kind == "banded":
        n = shape
        rp, ci, v = oracle.banded_csr(n, 3)
    else:
        rp, ci, v = oracle.stencil_csr(7, *shape)
        n = rp.size - 1
        v = v.copy(); v[ci == np.repeat(np.arange(n), np.diff(rp))] += 1.0      # strictly dominant
        rs = mkd.slab_row_start(*shape, 3)
    o = oracle.jacobi(rp, ci, v, np.ones(n), tol=tol, maxit=300, check_every=every)
    comms = mkd.Comm.local_world(len(rs) - 1)
    As = blocks(rp, ci, v, n, rs, comms)
    for a in As:
        a.set_spmv_kernel(mk.SPMV_SWEEP)          # small blocks would default to the pattern kernel
    assert all(a.diag_info().kernel == mk.SPMV_SWEEP for a in As), [a.diag_info().kernel for a in As]
    bs = [a.vector(fill=1.0) for a in As]
    xs = [a.vector(fill=0.0) for a in As]
    Ss = [mk.Jacobi(max_iter=300, tol=tol, check_every=every) for _ in As]
    res = mkd.dist_jacobi_solve(Ss, As, bs, xs)
    assert res[0]["converged"] == o["converged"] and res[0]["iterations"] == o["iterations"], (kind, res[0], o["iterations"])
    assert np.isclose(res[0]["rel_res_norm"], o["rel_res_norm"], rtol=1e-9)
    assert np.allclose(np.concatenate([x.get() for x in xs]), o["x"], rtol=1e-12, atol=0), kind
    del As, bs, xs, comms
print("split ok")
"""


@pytest.mark.parametrize("split", ["1", "0"])
def test_dist_jacobi_split_sweep_matches_oracle(split):
    """row-block Jacobi with the sweep cut into an interior launch (beside the halo and the off-rank product) and boundary
    launches: the default over RCCL, forced here in a local world through MK_JACOBI_SPLIT (read once per process, hence the
    child process); '0' runs the same cases through the one-launch order"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MK_JACOBI_SPLIT=split)
    out = subprocess.run([sys.executable, "-c", _SPLIT_SNIPPET, root], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "split ok" in out.stdout, out.stderr[-3000:]


def test_comm_may_be_destroyed_before_its_matrices():
    """a host language's finalisers run in any order: destroying the communicators first must leave the matrices usable
    for their own destruction (the last matrix tears the communicator down)"""
    rp, ci, v = oracle.stencil_csr(5, 12, 12)
    n = rp.size - 1
    rs = mkd.even_row_start(n, 2)
    comms = mkd.Comm.local_world(2)
    As = _blocks(rp, ci, v, n, rs, comms)
    L = mk.load()
    for c in comms:                       # communicators first (the wrappers must not destroy them a second time)
        L.mk_comm_destroy(c.h)
        c.h = None
    del comms
    infos = [A.info() for A in As]        # still readable
    assert infos[0]["n_ghost"] > 0
    del As                                # the last matrix releases the world


@pytest.mark.parametrize("kind,nx", [(7, 512), (27, 256)])
def test_full_size_rowblock_cg_matches_single_matrix(kind, nx):
    """the 8-GPU target line at full size (7-point 512^3 as 8 slabs of 64 planes; 27-point at 256^3: 512^3 has more entries
    than one matrix's int32 row pointers hold) driven as a local world: after 30 iterations the residual history equals the
    single-matrix solve's (other summation order of the inner products: 1e-9), and so does x"""
    world, iters = 8, 30
    n = nx ** 3
    A = mk.Matrix.stencil(kind, nx, nx, nx)
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=iters, tol=0.0, precond=1, logging=1, check_every=50)
    S.setup(A, b, x)
    one = S.solve(A, b, x)
    h1 = S.residual_history(iters + 1)
    x1 = x.get()
    del A, b, x, S
    rs = mkd.slab_row_start(nx, nx, nx, world)
    comms = mkd.Comm.local_world(world)
    As = [mkd.DistMatrix(mk.Matrix.stencil(kind, nx, nx, nx, row_begin=int(rs[r]), row_end=int(rs[r + 1])), n, rs, comms[r]) for r in range(world)]
    assert all(a.info()["n_ghost"] == (2 if 0 < r < world - 1 else 1) * nx * nx for r, a in enumerate(As))
    bs = [a.vector(fill=1.0) for a in As]
    xs = [a.vector(fill=0.0) for a in As]
    Ss = [mk.PCG(max_iter=iters, tol=0.0, precond=1, logging=1, check_every=50) for _ in As]
    res = mkd.dist_pcg_solve(Ss, As, bs, xs)
    assert all(q == res[0] for q in res) and res[0]["iterations"] == iters == one["iterations"]
    assert np.isclose(res[0]["rel_res_norm"], one["rel_res_norm"], rtol=1e-9)
    h8 = Ss[0].residual_history(iters + 1)
    assert h8.size == h1.size == iters + 1 and np.allclose(h8, h1, rtol=1e-9)
    x8 = np.concatenate([v.get() for v in xs])
    assert np.abs(x8 - x1).max() <= 1e-9 * np.abs(x1).max()
    del As, bs, xs, Ss, comms
    mk.memory_trim()


def test_dist_errors():
    rp, ci, v = oracle.stencil_csr(5, 8, 8)
    comms = mkd.Comm.local_world(2)
    blk = mk.Matrix.from_csr(32, 64, rp[:33], ci[:rp[32]], v[:rp[32]])
    with pytest.raises(mk.MkError):
        mkd.DistMatrix(blk, 64, np.array([0, 30, 64], np.int64), comms[0])     # block size != row range
    A0 = mkd.DistMatrix(mk.Matrix.from_csr(32, 64, rp[:33], ci[:rp[32]], v[:rp[32]]), 64, np.array([0, 32, 64], np.int64), comms[0])
    with pytest.raises(mk.MkError):
        mkd.dist_spmv([A0], [A0.vector()], [A0.vector()])                      # rank 1 never created its block


def test_rccl_code_path_single_rank():
    """RCCL itself refuses two ranks on one GPU, so the real communicator is exercised with one
    rank: dlopen + symbol binding, comm init, grouped send/recv, all-gather, all-reduce, and the
    solver running its RCCL branch (communication stream, events, all-reduced scalars)."""
    comm = mkd.Comm.single_rccl()
    comm.self_test()
    rp, ci, v = oracle.stencil_csr(7, 14, 14, 14)
    n = rp.size - 1
    A = mkd.DistMatrix(mk.Matrix.from_csr(n, n, rp, ci, v), n, np.array([0, n], np.int64), comm)
    b, x = A.vector(fill=1.0), A.vector(fill=0.0)
    S = mk.PCG(max_iter=500, tol=1e-8, precond=1)
    res = mkd.dist_pcg_solve([S], [A], [b], [x])[0]
    o = oracle.pcg(rp, ci, v, np.ones(n), tol=1e-8, precond=1, maxit=500)
    assert res["converged"] == 1 and abs(res["iterations"] - o["iterations"]) <= 2
    assert np.linalg.norm(x.get() - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
    y = A.vector()
    mkd.dist_spmv([A], [x], [y])
    assert np.allclose(y.get(), oracle.csr_spmv(rp, ci, v, x.get()), rtol=1e-12, atol=1e-12)
