"""CPU-side checks of the C ABI: the library loads, exports every symbol the header declares,
fails loudly without a GPU, and its host-only partitioner matches the oracle."""
import os
import re

import numpy as np
import pytest

import iterative_solvers_amd as mk
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = mk.load()
    header = open(os.path.join(ROOT, "include", "mk_krylov.h")).read()
    declared = set(re.findall(r"\b(mk_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(mk.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_compute_fails_loudly_without_init_or_gpu():
    import torch
    L = mk.load()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc = L.mk_init(0)
    assert rc < 0 and b"device" in L.mk_last_error_string().lower()
    with pytest.raises(mk.MkError):
        mk.Vector(4)


def test_partition_matches_oracle():
    rng = np.random.default_rng(3)
    for n, npart in [(1, 1), (10, 3), (512, 8), (1000, 7), (5, 8)]:
        lens = rng.integers(0, 9, n)
        lens[rng.random(n) < 0.3] = 0
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        a = mk.partition_rows(rp, npart)
        b = oracle.partition_rows(rp, npart)
        assert np.array_equal(a, b)
        assert a[0] == 0 and a[-1] == n and np.all(np.diff(a) >= 0)


def test_partition_balanced_on_stencil():
    rp, _, _ = oracle.stencil_csr(7, 16, 16, 16)
    rs = mk.partition_rows(rp, 8)
    share = np.diff(rp[rs].astype(np.int64))
    assert share.sum() == rp[-1]
    assert share.max() <= rp[-1] / 8 + 2 * 7
