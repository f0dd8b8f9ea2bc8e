#!/usr/bin/env python3
"""split kernel: plain product (mk_spmv) against the form with the fused dots (mk_spmv_dot, the CG step) on a short-row matrix
with a few long rows (fem_probe.py skew 4000000 10 50000 1500).  usage: split_modes_probe.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import iterative_solvers_amd as mk
import fem_probe
mk.init(0)
n, rp, ci = fem_probe.skew(4000000, 10, 50000, 1500)
v = np.random.default_rng(0).normal(size=ci.size)
A = mk.Matrix.from_csr(n, n, rp, ci, v)
x, y = mk.Vector(n, fill=1.0), mk.Vector(n)
print("kernel", A.info().kernel)
for name, fn in (("spmv", lambda: A.spmv(x, y)), ("spmv_dot", lambda: A.spmv_dot(x, y, want=False))):
    for _ in range(3): fn()
    mk.timer_start()
    for _ in range(20): fn()
    print(name, round(mk.timer_stop() / 20 * 1e3, 1), "us")
