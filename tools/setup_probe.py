#!/usr/bin/env python3
"""one-off costs: matrix generation, analysis (pattern tables, plans) and the first product.  usage: setup_probe.py kind nx ny nz [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk
kind, nx, ny, nz = (int(a) for a in sys.argv[1:5])
seed = int(sys.argv[5]) if len(sys.argv) > 5 else None
mk.init(0)
for rep in range(2):
    mk.synchronize(); t0 = time.perf_counter()
    A = mk.Matrix.stencil(kind, nx, ny, nz, scale_seed=seed)
    mk.synchronize(); t1 = time.perf_counter()
    info = A.info()          # runs the analysis
    mk.synchronize(); t2 = time.perf_counter()
    x, y = mk.Vector(A.ncols, fill=1.0), mk.Vector(A.nrows)
    A.spmv(x, y)
    mk.synchronize(); t3 = time.perf_counter()
    A.spmv(x, y)
    mk.synchronize(); t4 = time.perf_counter()
    print(f"kind={kind} {nx}x{ny}x{nz} seed={seed}: generate {1e3*(t1-t0):.1f} ms, analyse {1e3*(t2-t1):.1f} ms, first product {1e3*(t3-t2):.1f} ms, second {1e3*(t4-t3):.2f} ms, kernel {info.kernel}", flush=True)
    del A, x, y
