"""mk_spmv_dot_static (the entry point that allocates nothing: all scratch in a file-scope device array) against
mk_spmv_dot through the C ABI: y and both dot products bit for bit, on two calls in a row with the second input smaller
than the first (the array keeps the previous call's contents), for every kernel family."""
import os

import numpy as np
import pytest

import iterative_solvers_amd as mk
import oracle

pytestmark = pytest.mark.gpu


def _both(A, xh):
    n = A.nrows
    x = mk.Vector(n, values=xh)
    y0, y1 = mk.Vector(n, fill=-1.0), mk.Vector(n, fill=-2.0)
    d0 = A.spmv_dot(x, y0)
    d1 = A.spmv_dot_static(x, y1)
    return d0, y0.get(), d1, y1.get()


def _matrix(case, size):
    rng = np.random.default_rng(size)
    if case == "sweep":
        A = mk.Matrix.stencil(7, *((64, 64, 40) if size else (32, 16, 12)))
        A.set_spmv_kernel(mk.SPMV_SWEEP)
    elif case == "pattern":
        A = mk.Matrix.stencil(27, *((20, 18, 16) if size else (9, 8, 7)))
        A.set_spmv_kernel(mk.SPMV_PATTERN)
    elif case == "stream":
        A = mk.Matrix.stencil(7, *((48, 40, 36) if size else (14, 12, 10)), scale_seed=3)
        assert A.info().kernel == mk.SPMV_STREAM
    elif case == "vsweep":
        A = mk.Matrix.stencil(7, *((64, 48, 36) if size else (32, 16, 10)), scale_seed=7)
        A.set_spmv_kernel(mk.SPMV_VSWEEP)
    elif case in ("band", "strips"):
        # rows near the diagonal that share nothing: the band kernel (sliding window of x in LDS), in one window or over
        # diagonal strips (several launches, the dots in the last one)
        n, mean, hw = ((60000, 80, 12000) if size else (40000, 72, 9000)) if case == "strips" else ((40000, 30, 800) if size else (12000, 20, 300))
        lens = rng.integers(mean * 2 // 3, mean * 4 // 3 + 1, n)
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rows = np.repeat(np.arange(n), lens)
        ci = np.clip(rows + rng.integers(-hw, hw + 1, rows.size), 0, n - 1)
        ci = ci[np.lexsort((ci, rows))].astype(np.int32)
        old = {k: os.environ.get(k) for k in ("MK_BAND_MIN_ROWS", "MK_BAND_STRIP_MIN_ROWS")}
        os.environ.update({"MK_BAND_MIN_ROWS": "1000", "MK_BAND_STRIP_MIN_ROWS": "1000"})
        try:
            A = mk.Matrix.from_csr(n, n, rp, ci, rng.normal(size=ci.size))
            i = A.info()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert i.band_window > 0 and (i.band_strips >= 2) == (case == "strips"), (i.band_window, i.band_strips)
    elif case == "vector":
        A = mk.Matrix.stencil(5, *((300, 200) if size else (40, 30)), scale_seed=4)
        A.set_spmv_kernel(mk.SPMV_VECTOR, 8)
    else:   # split: skewed rows, entries cut into chunks, carries folded by the fix-up kernel
        scale, ne = (15, 1 << 19) if size else (11, 1 << 14)
        A = mk.Matrix.from_device_coo(mk.DeviceCOO.rmat(scale, ne, seed=5 + size))
        assert A.info().kernel == mk.SPMV_SPLIT
        A.spmv(mk.Vector(A.nrows, fill=1.0), mk.Vector(A.nrows))     # builds (allocates) the chunk table
    return A, rng.normal(size=A.nrows)


@pytest.mark.parametrize("case", ["sweep", "pattern", "stream", "vsweep", "vector", "split", "band", "strips"])
def test_spmv_dot_static_matches_spmv_dot(case):
    for size in (1, 0, 1):                 # large, then smaller, then large again: stale scratch would show
        A, xh = _matrix(case, size)
        d0, y0, d1, y1 = _both(A, xh)
        assert np.array_equal(y0.view(np.uint64), y1.view(np.uint64)), (case, size)
        assert np.float64(d0[0]).view(np.uint64) == np.float64(d1[0]).view(np.uint64) and d0[1] == d1[1], (case, size, d0, d1)
        # and the product is the oracle's (1e-12), the dots follow from it
        rp, ci, v = A.to_csr()
        ref = oracle.csr_spmv(rp, ci, v, xh)
        mag = oracle.csr_spmv(rp, ci, np.abs(v), np.abs(xh))
        assert (np.abs(y1 - ref) / np.maximum(mag, 1e-300)).max() <= 1e-12
        assert np.isclose(d1[0], ref @ xh, rtol=1e-10) and np.isclose(d1[1], xh @ xh, rtol=1e-12)


def test_spmv_dot_static_rejects_what_the_scratch_cannot_hold():
    # split-kernel matrix of more than 2^26 entries: its carry pairs do not fit the static array
    A = mk.Matrix.from_device_coo(mk.DeviceCOO.rmat(20, (1 << 26) + 4096, seed=2))
    assert A.info().kernel == mk.SPMV_SPLIT
    n = A.nrows
    x, y = mk.Vector(n, fill=1.0), mk.Vector(n)
    with pytest.raises(mk.MkError, match="chunk table"):
        A.spmv_dot_static(x, y)                      # nothing has built the chunk table yet
    A.spmv(x, y)
    with pytest.raises(mk.MkError, match="exceed the static scratch"):
        A.spmv_dot_static(x, y)
    assert A.spmv_dot(x, y)[1] == float(n)           # the allocating entry point still serves it
    with pytest.raises(mk.MkError, match="alias"):
        A.spmv_dot_static(x, x)


CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import iterative_solvers_amd as mk
import test_zz_no_alloc as T
mk.init(0)
for case in ["sweep", "pattern", "stream", "vsweep", "vector", "split"]:
    A, xh = T._matrix(case, 1)
    n = A.nrows
    x, y = mk.Vector(n, values=xh), mk.Vector(n)
    A.info()
    A.spmv_dot_static(x, y)            # first call of the process on this kernel (one-time attribute set-up, no memory)
    mk.synchronize()
    before = mk.memory_info() + mk.memory_calls()
    for _ in range(3):
        A.spmv_dot_static(x, y)
    mk.synchronize()
    after = mk.memory_info() + mk.memory_calls()
    assert before == after, (case, before, after)
print("no allocation seen")
"""


@pytest.mark.parametrize("no_cache", ["0", "1"])
def test_spmv_dot_static_allocates_nothing(no_cache):
    """the property the entry point exists for: the library's allocator sees no request and no release across the
    call -- every block it hands out or takes back goes through cached_alloc / cached_free, which mk_memory_calls counts --
    with the caching allocator on and off (the switch is read at mk_init, hence the child process)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MK_NO_ALLOC_CACHE=no_cache)
    out = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "no allocation seen" in out.stdout, out.stderr[-3000:]
