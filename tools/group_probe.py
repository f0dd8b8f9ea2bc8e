#!/usr/bin/env python3
"""one FE-like matrix of tools/fem_probe.py (rand3 <nodes> <mean_nb> <halfwin>) through the default SpMV kernel, 10 products:
the target of rocprofv3 passes (tools/pmc_passes.sh with PROBE=tools/group_probe.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import iterative_solvers_amd as mk
import fem_probe

mk.init(0)
a = [int(v) for v in sys.argv[1:4]] or [500000, 26, 300]
n, rp, ci = fem_probe.rand3(*a)
v = np.random.default_rng(0).normal(size=ci.size)
A = mk.Matrix.from_csr(n, n, rp, ci, v)
i = A.info()
x, y = mk.Vector(n, fill=1.0), mk.Vector(n)
for _ in range(3):
    A.spmv(x, y)
mk.timer_start()
for _ in range(10):
    A.spmv(x, y)
ms = mk.timer_stop() / 10
print(f"rand3 {a}: kernel {i.kernel} groups {i.row_groups} lanes {i.group_lanes}: {ms*1e3:.0f} us, frac {(12*ci.size+20*n)/ms/1e6/8000:.3f}")
