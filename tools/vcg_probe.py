#!/usr/bin/env python3
"""CG on variable-coefficient grids (vsweep kernel): ms per iteration and per kernel slot (HIP events inside the solve).
usage: vcg_probe.py [precond] nx,ny,nz [...]   (MK_NO_FUSED_K3=1 for the three-kernel loop)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk

mk.init(0)
args = sys.argv[1:]
precond = int(args.pop(0)) if args and "," not in args[0] else 0
iters = int(os.environ.get("MK_PROBE_ITERS", "100"))
for arg in args:
    nx, ny, nz = (int(a) for a in arg.split(","))
    A = mk.Matrix.stencil(int(os.environ.get("MK_PROBE_KIND", "7")), nx, ny, nz, scale_seed=1)
    n = A.nrows
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=10, tol=0.0, precond=precond)
    S.setup(A, b, x)
    S.solve(A, b, x)
    ts = []
    for _ in range(3):
        x.fill(0.0)
        S = mk.PCG(max_iter=iters, tol=0.0, precond=precond, check_every=50)
        S.setup(A, b, x)
        mk.synchronize()
        mk.timer_start()
        r = S.solve(A, b, x)
        ts.append(mk.timer_stop() / r["iterations"])
    x.fill(0.0)
    S = mk.PCG(max_iter=34, tol=0.0, precond=precond, check_every=50)
    S.set_kernel_timing(True)
    S.setup(A, b, x)
    S.solve(A, b, x)
    kt = S.kernel_timing()
    ms = sorted(ts)[1]
    by = 12 * A.nnz + (108 if precond else 92) * n
    print(f"{nx}x{ny}x{nz} precond {precond} kernel {A.info().kernel}: {ms:.4f} ms/iter  {1e3/ms:.1f} it/s  frac {by/ms/1e6/8000:.3f} of 8 TB/s on 8(d) bytes;"
          f" slots {[round(k[0], 4) for k in kt]}", flush=True)
    del S, A, b, x
