#!/usr/bin/env python3
"""Overhead of the row-block machinery on ONE GPU: CG on nx*ny*nz as a single matrix against the same
problem cut into `world` z-slabs driven as a local world (device copies instead of RCCL).
usage: dist_probe.py nx ny nz world [iters]      MK_PROBE_SINGLE=1: also time the single-reduction step"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk
from iterative_solvers_amd import distributed as mkd

nx, ny, nz, world = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 100
KIND = int(os.environ.get("MK_PROBE_KIND", "7"))
mk.init(0)
n = nx * ny * nz


def timed(fn):
    fn(5)
    mk.synchronize()
    t0 = time.perf_counter()
    fn(iters)
    mk.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


try:
    A = mk.Matrix.stencil(KIND, nx, ny, nz)
except mk.MkError as e:   # e.g. 27-point 512^3: 3.6 G entries do not fit one matrix's int32 row pointers
    print("single matrix skipped:", e, flush=True)
    A = None
b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)


def single(it):
    x.fill(0.0)
    S = mk.PCG(max_iter=it, tol=0.0, logging=0, check_every=50)
    S.setup(A, b, x)
    S.solve(A, b, x)


ms1 = timed(single) if A is not None else float('nan')
del A, b, x
rs = mkd.slab_row_start(nx, ny, nz, world)
comms = mkd.Comm.local_world(world)
As = [mkd.DistMatrix(mk.Matrix.stencil(KIND, nx, ny, nz, row_begin=int(rs[r]), row_end=int(rs[r + 1])), n, rs, comms[r]) for r in range(world)]
bs = [a.vector(fill=1.0) for a in As]
xs = [a.vector(fill=0.0) for a in As]


def dist(it):
    for v in xs:
        v.fill(0.0)
    Ss = [mk.PCG(max_iter=it, tol=0.0, logging=0, check_every=50) for _ in As]
    mkd.dist_pcg_solve(Ss, As, bs, xs)


ms2 = timed(dist)
ms3 = None
if os.environ.get("MK_PROBE_SINGLE"):
    def dist1(it):
        for v in xs:
            v.fill(0.0)
        Ss = [mk.PCG(max_iter=it, tol=0.0, logging=0, check_every=50) for _ in As]
        for S in Ss:
            S.set_single_reduction(True)
        mkd.dist_pcg_solve(Ss, As, bs, xs)
        assert Ss[0].single_reduction_used()
    ms3 = timed(dist1)
del As, bs, xs, comms   # release device objects before the interpreter (and a profiler's exit hooks) tear down
print(f"{nx}x{ny}x{nz}: single matrix {ms1:.3f} ms/iteration; {world} local ranks {ms2:.3f} ms/iteration "
      f"({ms2 / world:.3f} ms per rank-iteration)" + (f"; single-reduction step {ms3:.3f} ms/iteration ({ms3 / world:.3f} per rank-iteration)" if ms3 else ""), flush=True)
