#!/usr/bin/env python3
"""Overhead of the row-block machinery for plain Jacobi (BASELINE config 5) on ONE GPU: banded matrix of n rows as a single
matrix against the same problem cut into `world` row blocks driven as a local world.
usage: dist_jacobi_probe.py log2_rows world [iters]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk
from iterative_solvers_amd import distributed as mkd

n = 1 << int(sys.argv[1])
world = int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
mk.init(0)


def timed(fn):
    fn(20)
    mk.synchronize()
    t0 = time.perf_counter()
    fn(iters)
    mk.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


A = mk.Matrix.banded(n, 3)
b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)


def single(it):
    x.fill(0.0)
    S = mk.Jacobi(max_iter=it, tol=0.0, check_every=10)
    S.setup(A, b, x)
    S.solve(A, b, x)


us1 = timed(single)
del A, b, x
rs = np.linspace(0, n, world + 1).astype(np.int64)
comms = mkd.Comm.local_world(world)
As = [mkd.DistMatrix(mk.Matrix.banded(n, 3, row_begin=int(rs[r]), row_end=int(rs[r + 1])), n, rs, comms[r]) for r in range(world)]
bs = [a.vector(fill=1.0) for a in As]
xs = [a.vector(fill=0.0) for a in As]


def dist(it):
    for v in xs:
        v.fill(0.0)
    Ss = [mk.Jacobi(max_iter=it, tol=0.0, check_every=10) for _ in As]
    mkd.dist_jacobi_solve(Ss, As, bs, xs)


us2 = timed(dist)
del As, bs, xs, comms
print(f"banded 2^{sys.argv[1]} rows: single matrix {us1:.1f} us/iteration; {world} local ranks {us2:.1f} us/iteration "
      f"({us2 / world:.1f} us per rank-iteration)", flush=True)
