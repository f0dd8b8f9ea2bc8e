#!/usr/bin/env python3
"""BiCGSTAB iterations per second on a generated stencil (measurement for SURVEY 8f N3).
usage: bicgstab_probe.py kind nx ny nz [iters]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk

kind, nx, ny, nz = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 100
mk.init(0)
A = mk.Matrix.stencil(kind, nx, ny, nz)
n = A.nrows
b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
for it in (5, iters):
    x.fill(0.0)
    S = mk.BiCGSTAB(max_iter=it, tol=0.0, logging=0, check_every=50)
    S.setup(A, b, x)
    mk.synchronize()
    t0 = time.perf_counter()
    r = S.solve(A, b, x)
    mk.synchronize()
    dt = time.perf_counter() - t0
i = A.info()
spmv_b = 18 * n if i.kernel in (4, 5) else 12 * A.nnz + 20 * n
print(f"kind={kind} {nx}x{ny}x{nz} kernel={i.kernel}: {r['iterations']} iterations, {dt / r['iterations'] * 1e3:.3f} ms/iteration, "
      f"{r['iterations'] / dt:.1f} it/s, {(2 * spmv_b + 168 * n) * r['iterations'] / dt / 1e9:.0f} GB/s on 2 SpMV + 168 n bytes")
