#!/usr/bin/env python3
"""Time CG iterations over several grid sizes (A/B of vector-kernel bandwidth against the
spacing of the work vectors).  usage: cg_probe.py nx,ny,nz [nx,ny,nz ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk

mk.init(0)
iters = int(os.environ.get("MK_PROBE_ITERS", "60"))
for arg in sys.argv[1:]:
    nx, ny, nz = (int(a) for a in arg.split(","))
    A = mk.Matrix.stencil(int(os.environ.get("MK_PROBE_KIND", "7")), nx, ny, nz)
    if os.environ.get("MK_PROBE_KERNEL"):
        A.set_spmv_kernel(int(os.environ["MK_PROBE_KERNEL"]))
    n = A.nrows
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=10, tol=1e-30)
    S.setup(A, b, x)
    S.solve(A, b, x)
    x.fill(0.0)
    S = mk.PCG(max_iter=iters, tol=1e-30)
    S.setup(A, b, x)
    mk.timer_start()
    r = S.solve(A, b, x)
    ms = mk.timer_stop() / r["iterations"]
    by = 12 * A.nnz + 92 * n
    print(f"{nx}x{ny}x{nz} kernel {A.info().kernel}: {ms:.4f} ms/iter  {by/ms/1e6:.0f} GB/s effective  ({ms*1e6/n:.4f} ns/row)", flush=True)
    del S, A, b, x
