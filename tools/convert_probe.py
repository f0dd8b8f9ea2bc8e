#!/usr/bin/env python3
"""COO -> CSR on the device, timed on the second and third conversion of the same size (the first one pays for the
allocations).  usage: convert_probe.py [scale] [edges_log2]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ne = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 28)
mk.init(0)
for rep in range(3):
    coo = mk.DeviceCOO.rmat(scale, ne, seed=42 + rep)
    mk.synchronize()
    mk.timer_start()
    A = mk.Matrix.from_device_coo(coo)
    ms = mk.timer_stop()
    cb = 32 * A.nnz + 12 * A.nrows
    print(f"conversion {rep}: {ms:.2f} ms = {cb / ms / 1e6:.0f} GB/s on 32 nnz + 12 n", flush=True)
    del A, coo
