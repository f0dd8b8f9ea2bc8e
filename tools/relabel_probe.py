#!/usr/bin/env python3
"""Prototype measurement: SpMV on an R-MAT matrix before and after relabelling the columns by descending in-degree
(hot x entries packed into few cache lines).  Host-side relabelling with numpy; only the SpMV is timed.
usage: relabel_probe.py [scale] [edges_log2]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ne = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 28)
mk.init(0)


def rate(A, tag):
    n = A.nrows
    x, y = mk.Vector(A.ncols, fill=1.0), mk.Vector(n)
    for _ in range(3):
        A.spmv(x, y)
    mk.timer_start()
    for _ in range(10):
        A.spmv(x, y)
    ms = mk.timer_stop() / 10
    print(f"{tag}: SpMV {ms * 1e3:.0f} us, kernel {A.info().kernel}", flush=True)


coo = mk.DeviceCOO.rmat(scale, ne, seed=42)
A = mk.Matrix.from_device_coo(coo)
rate(A, "original columns")
t0 = time.time()
rp, ci, val = A.to_csr()
del A
n = rp.size - 1
deg = np.bincount(ci, minlength=n)
order = np.argsort(-deg, kind="stable")          # new position -> old column
perm = np.empty(n, np.int32)
perm[order] = np.arange(n, dtype=np.int32)       # old column -> new position
ci2 = perm[ci]
print(f"host relabel {time.time() - t0:.1f} s; top 2^18 columns hold {deg[order[:1 << 18]].sum() / ci.size:.3f} of the entries, "
      f"top 2^20 {deg[order[:1 << 20]].sum() / ci.size:.3f}", flush=True)
B = mk.Matrix.from_csr(n, n, rp, ci2, val)
rate(B, "columns by descending in-degree")
