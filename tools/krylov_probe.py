#!/usr/bin/env python3
"""Iterations per second of the other Krylov methods on a generated stencil (measurement for SURVEY 8f N3).
usage: krylov_probe.py bicgstab|cgnr|gmres kind nx ny nz [iters]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk

which = sys.argv[1]
kind, nx, ny, nz = (int(a) for a in sys.argv[2:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 90
mk.init(0)
A = mk.Matrix.stencil(kind, nx, ny, nz)
n = A.nrows
b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
for it in (6, iters):
    x.fill(0.0)
    if which == "bicgstab":
        S = mk.BiCGSTAB(max_iter=it, tol=0.0, logging=0, check_every=50)
    elif which == "cgnr":
        S = mk.CGNR(max_iter=it, tol=0.0, logging=0, check_every=50)
    else:
        S = mk.GMRES(max_iter=it, tol=0.0, k_dim=30, logging=0)
    S.setup(A, b, x)
    mk.synchronize()
    t0 = time.perf_counter()
    r = S.solve(A, b, x)
    mk.synchronize()
    dt = time.perf_counter() - t0
i = A.info()
spmv_b = 18 * n if i.kernel in (4, 5) else 12 * A.nnz + 20 * n
# vector bytes per iteration: BiCGSTAB 160 n; CGNR 112 n; GMRES(30) cycle average of (32 i + 40) n, i = 1..30
vec_b = {"bicgstab": 160, "cgnr": 112, "gmres": 32 * 15.5 + 40}[which] * n
nspmv = 1 if which == "gmres" else 2
print(f"{which} kind={kind} {nx}x{ny}x{nz} kernel={i.kernel}: {r['iterations']} iterations, {dt / r['iterations'] * 1e3:.3f} ms/iteration, "
      f"{r['iterations'] / dt:.1f} it/s, {(nspmv * spmv_b + vec_b) * r['iterations'] / dt / 1e9:.0f} GB/s on the byte model", flush=True)
