#!/usr/bin/env python3
"""bench.py -- CG iterations/s and SpMV effective HBM GB/s on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one CG iteration (K1 SpMV+dots, K2 x/r update+dots, K3 direction update).
Default workload, the same at every N (strong scaling): the configuration BASELINE.json's
metric and target are quoted on -- CG on the 3D 7-point Laplacian 512^3 (134M rows, 938M
entries, fp64, b = 1, x0 = 0), matrix generated in HBM (synthetic); it fits one MI355X.
At N=1 the line also carries BASELINE.json configs[1] (256^3, Jacobi-PCG) under "secondary".
Prints ONE JSON line (rank 0).  The timed region starts with everything resident in HBM.

For N>1 the driver starts one rank per GPU through torch.distributed.run; torch is used
for the rendezvous/barrier only, the solve runs inside libmi355_krylov.so.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (kind, nx, ny, nz, precond, description)
    "lap7_256_jacobi_pcg": (7, 256, 256, 256, 1, "3D 7-pt Laplacian 256^3, Jacobi-preconditioned CG, fp64"),
    "lap7_512_cg": (7, 512, 512, 512, 0, "3D 7-pt Laplacian 512^3, CG, fp64 (north-star scaling target)"),
    "poisson5_1024_cg": (5, 1024, 1024, 1, 0, "2D 5-pt Poisson 1024^2, CG, fp64"),
    "lap27_256_cg": (27, 256, 256, 256, 0, "3D 27-pt stencil 256^3, CG, fp64"),
    # BASELINE configs[3]; 3.6 G entries: row blocks only (--gpus >= 2), one matrix's int32 row pointers cannot hold it
    "lap27_512_cg": (27, 512, 512, 512, 0, "3D 27-pt stencil 512^3, CG, fp64, row-block partitioned"),
}


def cpu_baseline(kind, nx, ny, nz, precond, iters):
    """the oracle's OpenMP CG on the host cores, bounded sample of the same workload: a z-slab
    of at most 64 planes of the same operator; the rate is scaled to the full row count"""
    import oracle
    t0 = time.time()
    nz_s = min(nz, 64)
    rp, ci, v = oracle.stencil_csr(kind, nx, ny, nz_s)
    n = rp.size - 1
    b = np.ones(n)
    oracle.pcg(rp, ci, v, b, precond=precond, tol=0.0, maxit=2, omp=True)   # page-in / warm-up
    t1 = time.time()
    o = oracle.pcg(rp, ci, v, b, precond=precond, tol=0.0, maxit=iters, omp=True)
    dt = time.time() - t1
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    frac = nz_s / nz
    return {"value": round(o["iterations"] / dt * frac, 3), "unit": "CG iterations/s", "cores": cores, "kind": "port",
            "sample": f"{iters} CG iterations on a {nx}x{ny}x{nz_s} slab of the same operator ({frac:.3f} of the rows; "
                      f"measured {o['iterations'] / dt:.2f} it/s on the slab, scaled by {frac:.3f}); oracle/oracle.c orc_pcg, "
                      f"OpenMP; setup {t1 - t0:.1f}s excluded"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=20)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "MK_BENCH_DEVICE" in os.environ:     # rehearsal of the N>1 path on a box with fewer GPUs
        local_rank = int(os.environ["MK_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch through torch.distributed.run")

    import torch
    import torch.distributed as dist
    force_dist = bool(os.environ.get("MK_BENCH_FORCE_DIST"))   # rehearsal: the row-block path with a world of one
    if world > 1 or force_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        torch.cuda.set_device(local_rank)

    import iterative_solvers_amd as mk
    mk.init(local_rank)

    if world > 1 or force_dist:
        from iterative_solvers_amd import distributed as mkd
        return mkd.bench_main(args, rank, local_rank, world, WORKLOADS, HBM_PEAK_GBS)

    def measure(name, steps, warmup):
        kind, nx, ny, nz, precond, desc = WORKLOADS[name]
        A = mk.Matrix.stencil(kind, nx, ny, nz)
        n = A.nrows
        b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)

        def run(iters):
            S = mk.PCG(max_iter=iters, tol=0.0, two_norm=1, rel_change=1, precond=precond, logging=0, check_every=50)
            S.setup(A, b, x)
            x.fill(0.0)
            mk.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = S.solve(A, b, x)           # returns after the final read-back (stream drained)
            mk.synchronize()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert res["iterations"] == iters, res
            return dt, res

        run(max(warmup, 1))
        dt, res = run(steps)
        return A, b, dt, res

    name = args.workload or "lap7_512_cg"
    kind, nx, ny, nz, precond, desc = WORKLOADS[name]
    A, b, dt, res = measure(name, args.steps, args.warmup)
    n, nnz = A.nrows, A.nnz
    info = A.info()
    ms_per_step = dt * 1e3 / args.steps
    iters_per_s = args.steps / dt

    # dominant kernel: the fused SpMV (+<s,p>,<p,p>) launch, timed alone with HIP events on
    # the library's own stream
    s = mk.Vector(n)
    reps = 50
    for _ in range(5):
        A.spmv_dot(b, s, want=False)
    mk.timer_start()
    for _ in range(reps):
        A.spmv_dot(b, s, want=False)
    spmv_ms = mk.timer_stop() / reps
    kname = {1: "vector", 2: "stream", 3: "split", 4: "pattern", 5: "sweep"}.get(info.kernel, str(info.kernel))
    csr_bytes = 12 * nnz + 20 * n + 4           # val 8 + col 4 per entry, rowptr 4 + x 8 + y 8 per row
    # the pattern and sweep kernels (rows drawn from a table of (offsets, values) patterns) read none of the CSR
    # arrays: its algorithmic bytes are the 2-byte pattern id, x and y per row.  The roofline is priced
    # on what the kernel that ran has to move; the CSR-model rate is reported beside it.
    spmv_bytes = 18 * n + 4 if info.kernel in (4, 5) else csr_bytes
    achieved = spmv_bytes / (spmv_ms * 1e-3) / 1e9
    vec_bytes = (88 if precond else 72) * n     # K2 + K3: x, p, r, s (and 1/diag) streams
    iter_bytes = spmv_bytes + vec_bytes
    if info.kernel == 5 and not os.environ.get("MK_NO_FUSED_CG"):
        # fused step on sweep matrices: first sweep p' = z + beta p, dots of A p' (2 + 8 + 8 + 8 B/row), second sweep
        # r -= alpha A p' recomputed, x += alpha p' (2 + 8 + 16 + 16); 1/diag is one number on these operators
        iter_bytes = 68 * n
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(name + ":" + kname)

    out = {
        "metric": "CG iterations/s (fp64, fused SpMV+dot+axpy) with SpMV effective HBM GB/s",
        "value": round(iters_per_s, 2), "unit": "CG iterations/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": desc, "name": name, "rows": n, "nnz": nnz, "precond": "jacobi" if precond else "none",
                   "spmv_kernel": kname, "index_patterns": info.index_patterns, "value_patterns": info.value_patterns,
                   "parallelism": "1 GPU"},
        "iteration_effective_GBps": round(iter_bytes / (ms_per_step * 1e-3) / 1e9, 1),
        "iteration_csr_model_GBps": round((csr_bytes + vec_bytes) / (ms_per_step * 1e-3) / 1e9, 1),
        "roofline": {"bound": "hbm", "kernel": "spmv+dot (K1, %s kernel)" % kname, "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "bytes_per_launch": spmv_bytes, "avg_launch_ms": round(spmv_ms, 5),
                     "bytes_model": ("18 B/row: 2 B pattern id + 8 B x + 8 B y (what the pattern/sweep kernels read and write)"
                                     if info.kernel in (4, 5) else "SURVEY 8(d): 12 nnz + 20 n + 4"),
                     # the north star counts values + column indices + row pointers + x/y; priced that
                     # way a launch that reads 2 B of matrix data per row exceeds the peak by construction
                     "csr_model_bytes": csr_bytes, "csr_model_GBps": round(csr_bytes / (spmv_ms * 1e-3) / 1e9, 1),
                     "csr_model_frac": round(csr_bytes / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "final_rel_res_norm": res["rel_res_norm"],
    }
    # the timed region itself: one CG iteration = its kernels back to back (two sweeps on the fused path), priced on
    # the bytes they move; `traffic` is the PMC sum over the kernels of one iteration (profiles/traffic.json)
    it_traffic = None
    if os.path.exists(tpath):
        it_traffic = json.load(open(tpath)).get(name + ":" + kname + ":iteration")
    it_gbps = iter_bytes / (ms_per_step * 1e-3) / 1e9
    out["roofline_iteration"] = {"bound": "hbm", "kernel": "whole CG iteration (timed region of `value`)", "achieved": round(it_gbps, 1),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(it_gbps / HBM_PEAK_GBS, 4),
                                 "bytes_per_iteration": iter_bytes, "traffic": it_traffic}
    del A, b, s
    if args.workload is None:
        # BASELINE.json configs[1] beside the headline: 256^3, Jacobi-preconditioned CG, one GPU
        sname = "lap7_256_jacobi_pcg"
        A2, b2, dt2, res2 = measure(sname, args.steps, args.warmup)
        out["secondary"] = {sname: {"workload": WORKLOADS[sname][5], "value": round(args.steps / dt2, 2),
                                    "unit": "CG iterations/s", "ms_per_step": round(dt2 * 1e3 / args.steps, 4),
                                    "spmv_kernel": {1: "vector", 2: "stream", 3: "split", 4: "pattern", 5: "sweep"}.get(A2.info().kernel),
                                    "iteration_csr_model_GBps": round((12 * A2.nnz + 108 * A2.nrows) / (dt2 / args.steps) / 1e9, 1)}}
        del A2, b2
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(kind, nx, ny, nz, precond, args.cpu_iters)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
