import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import iterative_solvers_amd as mk
sys.path.insert(0, "/root/repo/tools")
from fem_probe import rand
mk.init(0)
for (n, mean, hw) in [(200000, 45, 3000), (150000, 20, 1000), (120001, 80, 500)]:
    n, rp, ci = rand(n, mean, hw)
    v = np.random.default_rng(0).normal(size=ci.size)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    i = A.info()
    xh = np.random.default_rng(1).normal(size=n)
    x, y = mk.Vector(n, values=xh), mk.Vector(n, fill=3.0)
    A.spmv(x, y)
    import scipy.sparse as sp
    M = sp.csr_matrix((v, ci, rp), shape=(n, n))
    ref = M @ xh
    mag = abs(M) @ abs(xh)
    err = (np.abs(y.get() - ref) / np.maximum(mag, 1e-300)).max()
    yx, xx = A.spmv_dot(x, y)
    err2 = (np.abs(y.get() - ref) / np.maximum(mag, 1e-300)).max()
    print(n, mean, hw, "kernel", i.kernel, "window", i.band_window, "err", err, err2, abs(yx - ref @ xh) / np.abs(ref * xh).sum(), abs(xx - xh @ xh) / (xh @ xh), flush=True)
