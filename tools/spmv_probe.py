#!/usr/bin/env python3
"""Launch the fused SpMV+dot kernel a few times on one workload (for rocprofv3 --pmc passes
and quick A/B timing).  usage: spmv_probe.py [kind nx ny nz] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk

kind, nx, ny, nz = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (7, 256, 256, 256)
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
kernel = int(os.environ.get("MK_PROBE_KERNEL", "0"))
mk.init(0)
A = mk.Matrix.banded(nx, hb=ny) if kind == 0 else mk.Matrix.stencil(kind, nx, ny, nz)   # kind 0: banded, n = nx, half bandwidth = ny
if kernel:
    A.set_spmv_kernel(kernel, int(os.environ.get("MK_PROBE_LANES", "0")))
n = A.nrows
x, y = mk.Vector(n, fill=1.0), mk.Vector(n)
for _ in range(3):
    A.spmv_dot(x, y, want=False)
mk.timer_start()
for _ in range(reps):
    A.spmv_dot(x, y, want=False)
ms = mk.timer_stop() / reps
b = 12 * A.nnz + 20 * n + 4
i = A.info()
print(f"kind={kind} {nx}x{ny}x{nz} kernel={i.kernel} patterns={i.index_patterns}/{i.value_patterns} spmv+dot {ms*1e3:.1f} us  "
      f"{b/ms/1e6:.1f} GB/s  ({b/ms/1e6/8000:.3f} of 8 TB/s)")
