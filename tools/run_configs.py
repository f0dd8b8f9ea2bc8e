#!/usr/bin/env python3
"""Run the five BASELINE.json configurations (or the part of each that fits ONE MI355X) and
print one JSON object per config.  The 8-GPU configs run their per-GPU share here (row block 0
of 8, no neighbours) so that kernel rates are on record; the real multi-GPU numbers come from
bench.py --gpus N on an 8-GPU node.   usage: run_configs.py [c1 c2 c3 c4 c5]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iterative_solvers_amd as mk  # noqa: E402

want = sys.argv[1:] or ["c1", "c2", "c3", "c4", "c5"]
mk.init(0)


def spmv_rate(A, reps=20):
    n = A.nrows
    x, y = mk.Vector(A.ncols, fill=1.0), mk.Vector(n)
    for _ in range(3):
        A.spmv(x, y)
    mk.timer_start()
    for _ in range(reps):
        A.spmv(x, y)
    ms = mk.timer_stop() / reps
    b = 12 * A.nnz + 20 * n + 4
    return ms, b / ms / 1e6


def cg(A, precond, tol, maxit, check_every=16):
    n = A.nrows
    b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
    S = mk.PCG(max_iter=maxit, tol=tol, precond=precond, logging=0, check_every=check_every)
    S.setup(A, b, x)
    mk.synchronize()
    t0 = time.perf_counter()
    res = S.solve(A, b, x)
    mk.synchronize()
    res["seconds"] = time.perf_counter() - t0
    res["iters_per_s"] = res["iterations"] / res["seconds"]
    return res


if "c1" in want:   # 2D 5-pt Poisson 1024^2 from a synthetic .mtx through the mmio path, unpreconditioned CG
    H = C.CDLL(os.path.join(ROOT, "host", "libmkhost.so"))
    H.fast_load_from_mtx_file.argtypes = [C.c_char_p, C.POINTER(mk.COO)]
    path = "/tmp/poisson5_1024.mtx"
    t0 = time.perf_counter()
    assert H.write_stencil_mtx(path.encode(), 5, 1024, 1024, 1) == 0
    t1 = time.perf_counter()
    coo = mk.COO()
    assert H.fast_load_from_mtx_file(path.encode(), C.byref(coo)) == 0
    t2 = time.perf_counter()
    A = C.c_void_p()
    mk._ck(mk.load().mk_matrix_from_coo(C.byref(coo), C.byref(A)))
    mk.synchronize()
    t3 = time.perf_counter()
    A = mk.Matrix(A)
    out = {"config": "c1 2D 5-pt Poisson 1024^2 via .mtx/mmio, CG (solver 50)", "rows": A.nrows, "nnz": A.nnz,
           "write_mtx_s": round(t1 - t0, 2), "parse_mtx_s": round(t2 - t1, 2), "upload_convert_s": round(t3 - t2, 3)}
    ms, gbs = spmv_rate(A, 100)
    out["spmv_us"], out["spmv_csr_model_GBps"] = round(ms * 1e3, 1), round(gbs, 1)
    for tol in (1e-6, 1e-10):
        r = cg(A, 0, tol, 5000)
        out[f"cg_tol{tol:g}"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()}
    print(json.dumps(out), flush=True)
    os.remove(path)

if "c2" in want:   # 3D 7-pt 256^3, Jacobi-PCG
    A = mk.Matrix.stencil(7, 256, 256, 256)
    ms, gbs = spmv_rate(A)
    r = cg(A, 1, 1e-6, 3000)
    print(json.dumps({"config": "c2 3D 7-pt 256^3 Jacobi-PCG (solver 51)", "rows": A.nrows, "nnz": A.nnz,
                      "spmv_us": round(ms * 1e3, 1), "spmv_csr_model_GBps": round(gbs, 1), "patterns": A.info().index_patterns,
                      "cg_tol1e-6": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()}}), flush=True)
    del A

if "c3" in want:   # R-MAT 2^24 rows, 2^28 entries: COO->CSR on the device + SpMV
    coo = mk.DeviceCOO.rmat(24, 1 << 28, seed=42)
    mk.synchronize()
    mk.timer_start()
    A = mk.Matrix.from_device_coo(coo)
    conv_ms = mk.timer_stop()
    i = A.info()
    ms, gbs = spmv_rate(A, 10)
    cb = 32 * A.nnz + 12 * A.nrows
    print(json.dumps({"config": "c3 R-MAT scale 24, 2^28 entries: COO->CSR + SpMV", "rows": A.nrows, "nnz": A.nnz,
                      "convert_ms": round(conv_ms, 2), "convert_GBps": round(cb / conv_ms / 1e6, 1),
                      "spmv_us": round(ms * 1e3, 1), "spmv_csr_model_GBps": round(gbs, 1), "kernel": i.kernel, "max_row": i.max_row_len,
                      "empty_rows": i.hist[0], "row_length_hist_log2": list(i.hist), "long_rows": i.long_rows}), flush=True)
    del A, coo

if "c4" in want:   # 27-pt 512^3 on 8 GPUs: one GPU's share (64 planes)
    n = 512 ** 3
    blk = mk.Matrix.stencil(27, 512, 512, 512, row_begin=0, row_end=n // 8)
    ms, gbs = spmv_rate(blk, 10)
    print(json.dumps({"config": "c4 27-pt 512^3: rank 0 of 8 (64 planes), local SpMV with global columns", "rows": blk.nrows,
                      "nnz": blk.nnz, "spmv_us": round(ms * 1e3, 1), "spmv_csr_model_GBps": round(gbs, 1)}), flush=True)
    del blk

if "c5" in want:   # banded Jacobi 2^26 rows on 8 GPUs: one GPU's share, and the whole problem on one GPU
    for nn, label in ((1 << 23, "rank share (2^23 rows)"), (1 << 26, "whole problem on one GPU (2^26 rows)")):
        A = mk.Matrix.banded(nn, 3)
        b, x = mk.Vector(nn, fill=1.0), mk.Vector(nn, fill=0.0)
        S = mk.Jacobi(max_iter=200, tol=0.0, check_every=10)
        S.setup(A, b, x)
        S.solve(A, b, x)
        mk.synchronize()
        t0 = time.perf_counter()
        r = S.solve(A, b, x)
        mk.synchronize()
        dt = time.perf_counter() - t0
        bytes_it = 12 * A.nnz + 28 * nn
        print(json.dumps({"config": "c5 banded hb=3 plain Jacobi (solver 70): " + label, "rows": nn, "nnz": A.nnz,
                          "iterations": r["iterations"], "us_per_iteration": round(dt / r["iterations"] * 1e6, 1),
                          "csr_model_GBps": round(bytes_it * r["iterations"] / dt / 1e9, 1)}), flush=True)
        del A, b, x
