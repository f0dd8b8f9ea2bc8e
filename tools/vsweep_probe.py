"""probe: the variable-coefficient sweep kernel against the stream kernel (bitwise) and its time per product"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import iterative_solvers_amd as mk

def time_spmv_dot(A, x, y, reps=20):
    for _ in range(3):
        A.spmv_dot(x, y, want=False)
    mk.timer_start()
    for _ in range(reps):
        A.spmv_dot(x, y, want=False)
    return mk.timer_stop() / reps

dims = [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]] or [(64, 64, 40), (256, 256, 256), (512, 512, 512)]
for (nx, ny, nz) in dims:
    kind = int(os.environ.get("MK_PROBE_KIND", "7")) if nz > 1 else 5
    A = mk.Matrix.stencil(kind, nx, ny, nz, scale_seed=1)
    n, nnz = A.nrows, A.nnz
    info = A.info()
    xh = np.random.default_rng(0).normal(size=n)
    x, y0, y1 = mk.Vector(n, values=xh), mk.Vector(n), mk.Vector(n)
    out = {"dims": (nx, ny, nz), "auto_kernel": info.kernel}
    try:
        A.set_spmv_kernel(mk.SPMV_VSWEEP)
    except mk.MkError as e:
        print(out, "vsweep not eligible:", e)
        continue
    d1 = A.spmv_dot(x, y1)
    t1 = time_spmv_dot(A, x, y1)
    A.set_spmv_kernel(mk.SPMV_STREAM)
    d0 = A.spmv_dot(x, y0)
    t0 = time_spmv_dot(A, x, y0)
    a, b = y0.get(), y1.get()
    csr = 12 * nnz + 20 * n + 4
    out.update(bitwise_equal=bool(np.array_equal(a.view(np.uint64), b.view(np.uint64))), maxdiff=float(np.abs(a - b).max()),
               dots=(d0, d1), stream_ms=round(t0, 4), vsweep_ms=round(t1, 4),
               stream_frac=round(csr / t0 / 1e6 / 8000, 3), vsweep_frac=round(csr / t1 / 1e6 / 8000, 3))
    print(out, flush=True)
    del A, x, y0, y1
