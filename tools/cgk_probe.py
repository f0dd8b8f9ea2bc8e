"""probe: per-kernel times of the CG loop (HIP events inside the solve) for one workload; env switches apply"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk
nx, ny, nz = (int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "512x512x512").split("x"))
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
precond = int(sys.argv[3]) if len(sys.argv) > 3 else 0
A = mk.Matrix.stencil(7, nx, ny, nz, scale_seed=seed if seed else None)
n = A.nrows
b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
for rep in range(2):
    S = mk.PCG(max_iter=34, tol=0.0, precond=precond, logging=0, check_every=50)
    S.set_kernel_timing(True)
    S.setup(A, b, x)
    x.fill(0.0)
    S.solve(A, b, x)
kt = S.kernel_timing()
# the plain product + dots as its own launches (bench.py's roofline.spmv_only), so that its counters are in the same passes
xs, ys = mk.Vector(n, fill=1.0), mk.Vector(n)
for _ in range(12):
    A.spmv_dot(xs, ys, want=False)
mk.synchronize()
print(f"{nx}x{ny}x{nz} seed={seed} kernel={A.info().kernel} ms:", [round(m, 4) for m, c in kt], "sum", round(sum(m for m, c in kt), 4), flush=True)
