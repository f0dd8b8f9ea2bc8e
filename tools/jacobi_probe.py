#!/usr/bin/env python3
"""Time plain Jacobi sweeps.  usage: jacobi_probe.py kind nx ny nz   (kind 0: banded, n = nx, half bandwidth = ny)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk

kind, nx, ny, nz = (int(a) for a in sys.argv[1:5])
mk.init(0)
A = mk.Matrix.banded(nx, hb=ny) if kind == 0 else mk.Matrix.stencil(kind, nx, ny, nz)
n = A.nrows
b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
for it in (20, 200):
    S = mk.Jacobi(max_iter=it, tol=0.0, check_every=50)
    x.fill(0.0)
    S.setup(A, b, x)
    mk.timer_start()
    S.solve(A, b, x)
    ms = mk.timer_stop()
print(f"kind={kind} {nx}x{ny}x{nz} kernel={A.info().kernel}: {ms / 200 * 1e3:.1f} us per Jacobi sweep, {26 * n / (ms / 200) / 1e6:.0f} GB/s on 26 B/row")
