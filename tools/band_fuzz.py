#!/usr/bin/env python3
"""Randomised shapes through the band kernel (single window, one block per CU, diagonal strips) against the stream kernel of the
same library on the same matrix (MK_NO_BAND=1) and against scipy: rows of very different lengths, fat rows, empty stretches,
unsorted columns, duplicates, rectangular matrices.  usage: band_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk

def case(rng, k):
    n = int(rng.integers(20000, 400000))
    extra_cols = int(rng.choice([0, 0, 0, 1, 777]))
    mean = int(rng.choice([int(rng.integers(4, 16)), int(rng.integers(16, 110))]))
    hw = int(rng.choice([50, 400, 1500, 3000, 5000, 9000, 20000]))
    lens = rng.integers(max(1, mean // 3), mean * 5 // 3 + 1, n)
    if rng.random() < 0.5:                       # empty stretches
        for _ in range(int(rng.integers(1, 4))):
            a = int(rng.integers(0, n)); lens[a:a + int(rng.integers(1, 3000))] = 0
    if rng.random() < 0.5:                       # fat rows
        idx = rng.integers(0, n, int(rng.integers(1, 200)))
        lens[idx] = np.minimum(lens[idx] * int(rng.integers(2, 12)), 1500)
    rp = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(rp[-1])
    rows = np.repeat(np.arange(n), lens)
    ncols = n + extra_cols
    ci = np.clip(rows + rng.integers(-hw, hw + 1, nnz), 0, ncols - 1)
    if rng.random() < 0.6:
        ci = ci[np.lexsort((ci, rows))]          # sorted within rows (else as drawn: unsorted, with duplicates)
    ci = ci.astype(np.int32)
    v = rng.normal(size=nnz)
    env = {"MK_BAND_MIN_ROWS": "1000", "MK_BAND_STRIP_MIN_ROWS": "1000"}
    os.environ.update(env)
    A = mk.Matrix.from_csr(n, ncols, rp.astype(np.int32), ci, v)
    i = A.info()
    os.environ["MK_NO_BAND"] = "1"
    B = mk.Matrix.from_csr(n, ncols, rp.astype(np.int32), ci, v)
    ib = B.info()
    del os.environ["MK_NO_BAND"]
    xh = rng.normal(size=ncols)
    yh = rng.normal(size=n)
    x = mk.Vector(ncols, values=xh)
    M = sp.csr_matrix((v, ci, rp), shape=(n, ncols))
    mag = abs(M) @ np.abs(xh) + 1.5 * np.abs(yh) + 1e-300
    worst = 0.0
    for alpha, beta in ((1.0, 0.0), (-0.5, 1.5)):
        ya, yb = mk.Vector(n, values=yh), mk.Vector(n, values=yh)
        A.spmv(x, ya, alpha, beta)
        B.spmv(x, yb, alpha, beta)
        ref = alpha * (M @ xh) + beta * yh
        worst = max(worst, (np.abs(ya.get() - ref) / mag).max(), (np.abs(yb.get() - ref) / mag).max())
    if n == ncols:
        ya = mk.Vector(n, fill=3.0)
        yx, xx = A.spmv_dot(x, ya)
        ref = M @ xh
        worst = max(worst, (np.abs(ya.get() - ref) / mag).max(), abs(yx - ref @ xh) / np.abs(ref * xh).sum(), abs(xx - xh @ xh) / (xh @ xh))
    print(f"case {k}: rows {n} cols {ncols} mean {nnz / n:.1f} max {lens.max()} hw {hw}: kernel {i.kernel} window {i.band_window} strips {i.band_strips} "
          f"(without: kernel {ib.kernel}) worst {worst:.2e}", flush=True)
    assert worst <= 1e-12, worst

def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    mk.init(0)
    for k in range(cases):
        case(rng, k)
    print("ok")

if __name__ == "__main__":
    main()
