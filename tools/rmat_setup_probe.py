#!/usr/bin/env python3
"""R-MAT (BASELINE configs[2]): time to the first product = conversion + analysis (row histogram, kernel choice, hot-first
column relabelling) + product, against the steady-state product.  usage: rmat_setup_probe.py [scale] [log2_edges]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ne = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else scale + 4)
mk.init(0)
for rep in range(3):
    coo = mk.DeviceCOO.rmat(scale, ne, seed=42 + rep)
    mk.synchronize(); t0 = time.perf_counter()
    A = mk.Matrix.from_device_coo(coo)
    mk.synchronize(); t1 = time.perf_counter()
    A.info()
    mk.synchronize(); t2 = time.perf_counter()
    x, y = mk.Vector(A.ncols, fill=1.0), mk.Vector(A.nrows)
    mk.synchronize(); t3 = time.perf_counter()
    A.spmv(x, y)
    mk.synchronize(); t4 = time.perf_counter()
    A.spmv(x, y)
    mk.synchronize(); t5 = time.perf_counter()
    print(f"convert {1e3*(t1-t0):.2f} ms, analyse {1e3*(t2-t1):.2f} ms, first product {1e3*(t4-t3):.2f} ms, second {1e3*(t5-t4):.2f} ms", flush=True)
    del A, x, y, coo
