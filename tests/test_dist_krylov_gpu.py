"""Row-block BiCGSTAB (solver id 6), GMRES(k) (id 12) and CGNR (id 40) in a local world (several logical ranks on one GPU):
the same loops as on one matrix with every product behind a halo exchange and every inner product all-reduced, checked
against the global oracle (oracle/oracle.c orc_bicgstab / orc_gmres / orc_cgnr) and against the one-matrix solvers."""
import numpy as np
import pytest

import iterative_solvers_amd as mk
from iterative_solvers_amd import distributed as mkd
import oracle

pytestmark = pytest.mark.gpu


def _convection_diffusion(kind, dims, c=0.4):
    rp, ci, v = oracle.stencil_csr(kind, *dims)
    n = rp.size - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    v = v.copy()
    v[ci == rows + 1] += c
    v[ci == rows - 1] -= c
    return rp, ci, v, n


def _blocks(rp, ci, v, n, rs, comms):
    As = []
    for r, cm in enumerate(comms):
        lo, hi = int(rs[r]), int(rs[r + 1])
        blk = mk.Matrix.from_csr(hi - lo, n, rp[lo:hi + 1] - rp[lo], ci[rp[lo]:rp[hi]], v[rp[lo]:rp[hi]])
        As.append(mkd.DistMatrix(blk, n, rs, cm))
    return As


def _world(world):
    return mkd.Comm.local_world(world) if world > 1 else [None]


def _true_rel(rp, ci, v, bh, xs):
    return np.linalg.norm(bh - oracle.csr_spmv(rp, ci, v, xs)) / np.linalg.norm(bh)


@pytest.mark.parametrize("world,kind,dims,cuts", [(1, 7, (20, 16, 12), None), (2, 7, (20, 16, 12), None), (3, 5, (64, 48, 1), None),
                                                  (4, 7, (20, 16, 12), [0, 900, 900, 2500, 3840])])
def test_dist_bicgstab_matches_oracle(world, kind, dims, cuts):
    rp, ci, v, n = _convection_diffusion(kind, dims)
    bh = np.random.default_rng(11).normal(size=n)
    tol = 1e-9
    o = oracle.bicgstab(rp, ci, v, bh, tol=tol, maxit=500, want_norms=True)
    rs = np.array(cuts, np.int64) if cuts else mkd.even_row_start(n, world)
    As = _blocks(rp, ci, v, n, rs, _world(world))
    for check_every in (1, 16):
        bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
        xs = [A.vector(fill=0.0) for A in As]
        Ss = [mk.BiCGSTAB(max_iter=500, tol=tol, logging=1, check_every=check_every) for _ in As]
        res = mkd.dist_bicgstab_solve(Ss, As, bs, xs)
        assert all(q == res[0] for q in res)
        assert res[0]["converged"] == 1 == o["converged"]
        assert abs(res[0]["iterations"] - o["iterations"]) <= max(3, 0.05 * o["iterations"]), (res[0], o["iterations"])
        xg = np.concatenate([x.get() for x in xs])
        true_rel = _true_rel(rp, ci, v, bh, xg)
        assert true_rel <= tol * 1.0001 and abs(res[0]["rel_res_norm"] - true_rel) <= 1e-3 * tol
        assert np.linalg.norm(xg - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
        hist = Ss[-1].residual_history(1000)
        assert hist.size == res[0]["iterations"] + 1
        m = min(hist.size, o["norms"].size, 12)
        assert np.allclose(hist[:m], o["norms"][:m], rtol=1e-6)


def test_dist_bicgstab_maxit_continue_and_zero_rhs():
    rp, ci, v, n = _convection_diffusion(5, (40, 40, 1))
    world = 3
    rs = mkd.even_row_start(n, world)
    As = _blocks(rp, ci, v, n, rs, _world(world))
    bs = [A.vector(fill=1.0) for A in As]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.BiCGSTAB(max_iter=4, tol=1e-14) for _ in As]
    res = mkd.dist_bicgstab_solve(Ss, As, bs, xs)
    o = oracle.bicgstab(rp, ci, v, np.ones(n), tol=1e-14, maxit=4)
    assert res[0]["iterations"] == 4 and res[0]["converged"] == 0
    assert np.isclose(res[0]["rel_res_norm"], o["rel_res_norm"], rtol=1e-8)
    assert np.allclose(np.concatenate([x.get() for x in xs]), o["x"], rtol=1e-9, atol=1e-13)
    res2 = mkd.dist_bicgstab_solve(Ss, As, bs, xs)       # niter > 1 re-solves from the previous x (solvers.c:541-542)
    o2 = oracle.bicgstab(rp, ci, v, np.ones(n), x0=o["x"], tol=1e-14, maxit=4)
    assert np.isclose(res2[0]["rel_res_norm"], o2["rel_res_norm"], rtol=1e-7)
    b0 = [A.vector(fill=0.0) for A in As]
    x0 = [A.vector(fill=0.0) for A in As]
    r0 = mkd.dist_bicgstab_solve(Ss, As, b0, x0)
    assert r0[0]["iterations"] == 0 and all(np.all(x.get() == 0.0) for x in x0)


@pytest.mark.parametrize("world,kind,dims,k_dim", [(1, 7, (16, 14, 12), 30), (2, 7, (16, 14, 12), 30), (3, 5, (48, 40, 1), 30),
                                                   (4, 7, (16, 14, 12), 5)])
def test_dist_gmres_matches_oracle(world, kind, dims, k_dim):
    rp, ci, v, n = _convection_diffusion(kind, dims)
    bh = np.random.default_rng(14).normal(size=n)
    tol = 1e-9
    o = oracle.gmres(rp, ci, v, bh, tol=tol, maxit=2000, k_dim=k_dim, want_norms=True)
    rs = mk.partition_rows(rp, world)
    As = _blocks(rp, ci, v, n, rs, _world(world))
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.GMRES(max_iter=2000, tol=tol, k_dim=k_dim, logging=1) for _ in As]
    res = mkd.dist_gmres_solve(Ss, As, bs, xs)
    assert all(q == res[0] for q in res)
    assert res[0]["converged"] == 1 == o["converged"]
    assert abs(res[0]["iterations"] - o["iterations"]) <= max(2, 0.02 * o["iterations"]), (res[0], o["iterations"])
    xg = np.concatenate([x.get() for x in xs])
    true_rel = _true_rel(rp, ci, v, bh, xg)
    assert true_rel <= tol * 1.0001 and abs(res[0]["rel_res_norm"] - true_rel) <= 1e-3 * tol
    assert np.linalg.norm(xg - o["x"]) <= 1e-6 * np.linalg.norm(o["x"])
    hist = Ss[-1].residual_history(4000)
    assert hist.size == res[0]["iterations"] + 1
    m = min(hist.size, o["norms"].size, 2 * k_dim + 5)
    assert np.allclose(hist[:m], o["norms"][:m], rtol=1e-7)


def test_dist_gmres_maxit_restart_empty_rank_and_zero_rhs():
    rp, ci, v, n = _convection_diffusion(5, (40, 40, 1))
    rs = np.array([0, 500, 500, n], np.int64)            # rank 1 owns nothing
    As = _blocks(rp, ci, v, n, rs, _world(3))
    bs = [A.vector(fill=1.0) for A in As]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.GMRES(max_iter=7, tol=1e-14, k_dim=3) for _ in As]       # 2 full cycles + 1 iteration
    res = mkd.dist_gmres_solve(Ss, As, bs, xs)
    o = oracle.gmres(rp, ci, v, np.ones(n), tol=1e-14, maxit=7, k_dim=3)
    assert res[0]["iterations"] == 7 and res[0]["converged"] == 0
    assert np.isclose(res[0]["rel_res_norm"], o["rel_res_norm"], rtol=1e-9)
    assert np.allclose(np.concatenate([x.get() for x in xs]), o["x"], rtol=1e-9, atol=1e-13)
    b0 = [A.vector(fill=0.0) for A in As]
    x0 = [A.vector(fill=0.0) for A in As]
    r0 = mkd.dist_gmres_solve(Ss, As, b0, x0)
    assert r0[0]["iterations"] == 0 and all(np.all(x.get() == 0.0) for x in x0)


def test_dist_krylov_rejects_mismatched_settings_and_foreign_vectors():
    rp, ci, v, n = _convection_diffusion(5, (16, 16, 1))
    rs = mkd.even_row_start(n, 2)
    As = _blocks(rp, ci, v, n, rs, _world(2))
    bs = [A.vector(fill=1.0) for A in As]
    xs = [A.vector(fill=0.0) for A in As]
    with pytest.raises(mk.MkError, match="disagree"):
        mkd.dist_bicgstab_solve([mk.BiCGSTAB(max_iter=5), mk.BiCGSTAB(max_iter=6)], As, bs, xs)
    with pytest.raises(mk.MkError, match="disagree"):
        mkd.dist_gmres_solve([mk.GMRES(k_dim=5), mk.GMRES(k_dim=6)], As, bs, xs)
    plain = [mk.Vector(int(rs[k + 1] - rs[k]), fill=0.0) for k in range(2)]    # no room for the ghosts
    with pytest.raises(mk.MkError, match="mk_dist_vector_create"):
        mkd.dist_gmres_solve([mk.GMRES(), mk.GMRES()], As, bs, plain)
    with pytest.raises(mk.MkError, match="all of them"):
        mkd.dist_bicgstab_solve([mk.BiCGSTAB()], As[:1], bs[:1], xs[:1])


@pytest.mark.parametrize("world,kind,dims,cuts", [(1, 7, (14, 12, 10), None), (2, 7, (14, 12, 10), None), (3, 5, (40, 32, 1), None),
                                                  (4, 7, (14, 12, 10), [0, 500, 500, 1100, 1680])])
def test_dist_cgnr_matches_oracle(world, kind, dims, cuts):
    rp, ci, v, n = _convection_diffusion(kind, dims)
    bh = np.random.default_rng(12).normal(size=n)
    tol = 1e-8
    o = oracle.cgnr(rp, ci, v, bh, tol=tol, maxit=5000, want_norms=True)
    rs = np.array(cuts, np.int64) if cuts else mkd.even_row_start(n, world)
    As = _blocks(rp, ci, v, n, rs, _world(world))
    for check_every in (1, 16):
        bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
        xs = [A.vector(fill=0.0) for A in As]
        Ss = [mk.CGNR(max_iter=5000, tol=tol, logging=1, check_every=check_every) for _ in As]
        res = mkd.dist_cgnr_solve(Ss, As, bs, xs)
        assert all(q == res[0] for q in res)
        assert res[0]["converged"] == 1 == o["converged"]
        assert abs(res[0]["iterations"] - o["iterations"]) <= max(2, 0.02 * o["iterations"]), (res[0], o["iterations"])
        xg = np.concatenate([x.get() for x in xs])
        true_rel = _true_rel(rp, ci, v, bh, xg)
        assert true_rel < tol and abs(res[0]["rel_res_norm"] - true_rel) <= 1e-3 * tol
        assert np.linalg.norm(xg - o["x"]) <= 1e-5 * np.linalg.norm(o["x"])
        hist = Ss[-1].residual_history(6000)
        assert hist.size == res[0]["iterations"] + 1
        m = min(hist.size, o["rel_norms"].size, 30)
        assert np.allclose(hist[:m], o["rel_norms"][:m], rtol=1e-8)


def test_dist_cgnr_far_from_symmetric_all_to_all_and_zero_rhs():
    """random rows (every rank's rows touch columns of every other rank, nothing symmetric): the reverse exchange must carry
    the real transpose; a fixed number of steps against the oracle, then the zero right-hand side"""
    rng = np.random.default_rng(13)
    n, world = 3000, 4
    lens = rng.integers(1, 7, n)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(n, k, replace=False)) for k in lens]).astype(np.int32)
    v = rng.normal(size=rp[-1])
    rows = np.repeat(np.arange(n), lens)
    hit = ci == rows
    v[hit] = 8.0 + rng.uniform(0, 1, int(hit.sum()))
    # rows without a diagonal entry get one (kept sorted)
    need = np.setdiff1d(np.arange(n), rows[hit])
    r2 = np.concatenate([rows, need]); c2 = np.concatenate([ci, need]).astype(np.int32); v2 = np.concatenate([v, 8.0 + rng.uniform(0, 1, need.size)])
    order = np.lexsort((c2, r2))
    r2, c2, v2 = r2[order], c2[order], v2[order]
    rp2 = np.concatenate([[0], np.cumsum(np.bincount(r2, minlength=n))]).astype(np.int32)
    rs = mk.partition_rows(rp2, world)
    As = _blocks(rp2, c2, v2, n, rs, _world(world))
    assert max(A.info()["n_neighbors"] for A in As) == world - 1
    bh = rng.normal(size=n)
    bs = [A.vector(values=bh[rs[k]:rs[k + 1]]) for k, A in enumerate(As)]
    xs = [A.vector(fill=0.0) for A in As]
    Ss = [mk.CGNR(max_iter=6, tol=1e-14) for _ in As]
    res = mkd.dist_cgnr_solve(Ss, As, bs, xs)
    o = oracle.cgnr(rp2, c2, v2, bh, tol=1e-14, maxit=6)
    assert res[0]["iterations"] == 6 and res[0]["converged"] == 0
    assert np.isclose(res[0]["rel_res_norm"], o["rel_res_norm"], rtol=1e-9)
    assert np.allclose(np.concatenate([x.get() for x in xs]), o["x"], rtol=1e-9, atol=1e-13)
    b0 = [A.vector(fill=0.0) for A in As]
    x0 = [A.vector(fill=2.0) for A in As]
    r0 = mkd.dist_cgnr_solve(Ss, As, b0, x0)
    assert r0[0]["iterations"] == 0 and all(np.all(x.get() == 0.0) for x in x0)
