#!/usr/bin/env python3
"""BASELINE config 3: R-MAT (scale, edges) generated in HBM -> COO->CSR on the device -> SpMV.
usage: rmat_probe.py [scale] [log2_edges]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ne = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else scale + 4)
mk.init(0)
t0 = time.perf_counter()
coo = mk.DeviceCOO.rmat(scale, ne, seed=42)
mk.synchronize()
t1 = time.perf_counter()
mk.timer_start()
A = mk.Matrix.from_device_coo(coo)
conv_ms = mk.timer_stop()
n, nnz = A.nrows, A.nnz
i = A.info()
cb = 32 * nnz + 12 * n
print(f"rmat scale={scale} edges={ne}: generate {1e3*(t1-t0):.1f} ms; COO->CSR {conv_ms:.2f} ms "
      f"= {cb/conv_ms/1e6:.0f} GB/s effective ({cb/conv_ms/1e6/8000:.3f} of 8 TB/s)")
print(f"rows: max {i.max_row_len} mean {i.mean_row_len:.1f} empty {i.hist[0]} hist {list(i.hist)} kernel {i.kernel} lanes {i.lanes_per_row} long {i.long_rows}")
x, y = mk.Vector(n, fill=1.0), mk.Vector(n)
for _ in range(2):
    A.spmv(x, y)
reps = 10
mk.timer_start()
for _ in range(reps):
    A.spmv(x, y)
ms = mk.timer_stop() / reps
sb = 12 * nnz + 20 * n + 4
print(f"SpMV {ms*1e3:.0f} us = {sb/ms/1e6:.0f} GB/s effective ({sb/ms/1e6/8000:.3f} of 8 TB/s)")
if os.environ.get("MK_PROBE_CHECK"):
    import numpy as np
    xh = np.random.default_rng(0).normal(size=n)
    x.set(xh)
    A.spmv(x, y)
    got = y.get()
    yx, xx = A.spmv_dot(x, y)
    A.set_spmv_kernel(mk.SPMV_VECTOR)
    A.spmv(x, y)
    ref = y.get()
    print("max |panel - vector| / max|ref| =", np.abs(got - ref).max() / np.abs(ref).max(), " dots rel err", abs(yx - ref @ xh) / np.abs(ref * xh).sum(), abs(xx - xh @ xh) / (xh @ xh))
