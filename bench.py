#!/usr/bin/env python3
"""bench.py -- CG iterations/s and SpMV effective HBM GB/s on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one CG iteration.  Default workload, the same at every N (strong scaling): the configuration
BASELINE.json's metric and target are quoted on -- CG on the 3D 7-point Laplacian 512^3 (134M rows, 938M entries,
fp64, b = 1, x0 = 0), matrix generated in HBM (synthetic); it fits one MI355X.  Prints ONE JSON line (rank 0).
The timed region starts with everything resident in HBM.

At N = 1 the line also carries
  * `roofline`: the dominant kernel of a CG loop that reads the CSR arrays -- the same grid with VARIABLE coefficients
    (D A D, every row its own values, like the authors' SuiteSparse inputs), priced on SURVEY.md 8(d)'s bytes
    12 nnz + 20 n + 4, its duration taken by HIP events inside the solve loop (mk_pcg_set_kernel_timing);
  * `roofline_value_loop`: the kernels of the loop `value` times (constant coefficients: the two sweeps of the fused
    step, which read a 2-byte pattern id per row instead of the CSR arrays), priced on THEIR bytes;
  * `secondary`: BASELINE.json configs[1] (256^3 Jacobi-PCG) and configs[2] (R-MAT scale 24: COO->CSR + SpMV), and the product on two
    finite-element-like CSR matrices built on the host (tools/fem_probe.py: the shape of the reference's SuiteSparse inputs);
  * `cpu_baseline`: the oracle's OpenMP CG on the full 512^3 operator on the host cores.

`python bench.py --cpu-configs c1,c2,c3` is a separate mode (one JSON line per configuration, not the contract line): the
oracle's CPU solve / product / conversion of BASELINE configs[0..2] beside the GPU's, SURVEY.md 8(d)'s per-config baselines.

For N > 1 the driver starts one rank per GPU through torch.distributed.run; started plainly (`python bench.py --gpus N`,
no WORLD_SIZE in the environment) this script starts the N ranks itself.  torch is used for the rendezvous/barrier
only, the solve runs inside libmi355_krylov.so.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md 8(d) timing rules for the CPU leg: OMP_PROC_BIND=spread (mirrors scripts/job_amd:103), set before any OpenMP
# runtime is loaded (torch brings one, the oracle's libgomp another); a caller's own setting wins
os.environ.setdefault("OMP_PROC_BIND", "spread")
TIMED_SOLVES = 5    # median of 5 timed solves, x0 reset before each

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (a float4 copy measures 6.29)

WORKLOADS = {
    # name: (kind, nx, ny, nz, precond, scale_seed, description)
    "lap7_256_jacobi_pcg": (7, 256, 256, 256, 1, None, "3D 7-pt Laplacian 256^3, Jacobi-preconditioned CG, fp64"),
    "lap7_512_cg": (7, 512, 512, 512, 0, None, "3D 7-pt Laplacian 512^3, CG, fp64 (north-star scaling target)"),
    "lap7_512_var_cg": (7, 512, 512, 512, 0, 1, "3D 7-pt operator 512^3 with variable coefficients (D A D, per-row scale), CG, fp64"),
    "lap7_256_var_jacobi_pcg": (7, 256, 256, 256, 1, 1, "3D 7-pt operator 256^3 with variable coefficients, Jacobi-preconditioned CG, fp64"),
    "poisson5_1024_cg": (5, 1024, 1024, 1, 0, None, "2D 5-pt Poisson 1024^2, CG, fp64"),
    "lap27_256_cg": (27, 256, 256, 256, 0, None, "3D 27-pt stencil 256^3, CG, fp64"),
    "lap27_256_var_cg": (27, 256, 256, 256, 0, 1, "3D 27-pt operator 256^3 with variable coefficients (BASELINE configs[3]'s stencil, one GPU's share of rows, every entry its own value), CG, fp64"),
    # BASELINE configs[3]; 3.6 G entries: row blocks only (--gpus >= 2), one matrix's int32 row pointers cannot hold it
    "lap27_512_cg": (27, 512, 512, 512, 0, None, "3D 27-pt stencil 512^3, CG, fp64, row-block partitioned"),
}
KERNEL_NAMES = {1: "vector", 2: "stream", 3: "split", 4: "pattern", 5: "sweep", 6: "vsweep"}


def host_cpu():
    """model name and physical core count of the box (the reference's job scripts print lscpu: scripts/job_gpu:114-116)"""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, len(cores)


def cpu_baseline(kind, nx, ny, nz, precond, iters):
    """the oracle's OpenMP CG (oracle/oracle.c orc_pcg, the CPU restatement of the hypre loop) on the FULL operator of
    the headline workload, `iters` iterations after a 2-iteration warm-up; threads = what OpenMP finds on this box"""
    import oracle
    t0 = time.time()
    rp, ci, v = oracle.stencil_csr(kind, nx, ny, nz)
    n = rp.size - 1
    b = np.ones(n)
    oracle.pcg(rp, ci, v, b, precond=precond, tol=0.0, maxit=2, omp=True)   # page-in / warm-up
    t1 = time.time()
    o = oracle.pcg(rp, ci, v, b, precond=precond, tol=0.0, maxit=iters, omp=True)
    dt = time.time() - t1
    model, phys = host_cpu()
    threads = oracle.num_threads()
    return {"value": round(o["iterations"] / dt, 3), "unit": "CG iterations/s", "cores": threads, "kind": "port",
            "cpu_model": model, "physical_cores_of_the_machine": phys,
            "sample": f"{o['iterations']} CG iterations on the full {nx}x{ny}x{nz} operator ({n} rows, {rp[-1]} entries; the first "
                      f"{o['iterations']} of the iterations the GPU times), {dt:.1f} s; oracle/oracle.c orc_pcg, OpenMP with {threads} "
                      f"threads (OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS', 'unset')}, OMP_PROC_BIND="
                      f"{os.environ.get('OMP_PROC_BIND', 'unset')}); matrix generation + warm-up {t1 - t0:.1f} s not counted"}


def _median_time(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), r


def cpu_configs(mk, which):
    """`--cpu-configs c1,c2,c3`: SURVEY.md 8(d)'s per-config CPU baselines beside the GPU figure of the same config, one
    JSON line per config (a separate mode: the default run prints its single line and times only the headline's baseline).
    c1: full solve at tol 1e-6 and 1e-10; c2: SpMV + 50 iterations of Jacobi-PCG; c3: COO->CSR by the REFERENCE's own
    convert_coo_to_csr (oracle/_ref, compiled from /root/reference/src/sparse_matrix.c) and by the oracle's restatement,
    then SpMV.  CPU legs: the oracle's OpenMP kernels with the threads OpenMP finds; the conversion is serial like the
    reference's.  Medians of 5 (GPU) / 3 (CPU), except the conversions, which run once."""
    import oracle
    model, phys = host_cpu()
    threads = oracle.num_threads()
    head = {"cpu_model": model, "physical_cores_of_the_machine": phys, "omp_threads": threads,
            "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS", "unset"), "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND", "unset")}

    def gpu_spmv_ms(A, reps=20):
        x, y = mk.Vector(A.ncols, fill=1.0), mk.Vector(A.nrows)
        for _ in range(3):
            A.spmv(x, y)
        ts = []
        for _ in range(5):
            mk.timer_start()
            for _ in range(reps):
                A.spmv(x, y)
            ts.append(mk.timer_stop() / reps)
        return float(np.median(ts))

    def gpu_cg(A, precond, tol, maxit):
        n = A.nrows
        b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)
        ts, res = [], None
        for _ in range(6):                                  # the first solve is the warm-up
            S = mk.PCG(max_iter=maxit, tol=tol, precond=precond, logging=0, check_every=16 if tol > 0 else 50)
            S.setup(A, b, x)
            x.fill(0.0)                                     # x0 reset before every timed solve (8(d) timing rules)
            mk.synchronize()
            t0 = time.perf_counter()
            res = S.solve(A, b, x)
            mk.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts[1:])), res

    if "c1" in which:
        rp, ci, v = oracle.stencil_csr(5, 1024, 1024, 1)
        n = rp.size - 1
        b = np.ones(n)
        A = mk.Matrix.stencil(5, 1024, 1024, 1)
        o = dict(head, config="c1: 2D 5-pt Poisson 1024^2, unpreconditioned CG (solver 50), b = 1, x0 = 0", rows=n, nnz=int(rp[-1]),
                 cache_resident=True)
        x = np.ones(n)
        o["cpu_spmv_ms"] = round(_median_time(lambda: oracle.csr_spmv(rp, ci, v, x, omp=True), 5)[0] * 1e3, 3)
        o["gpu_spmv_ms"] = round(gpu_spmv_ms(A, 100), 5)
        for tol in (1e-6, 1e-10):
            dt, r = _median_time(lambda: oracle.pcg(rp, ci, v, b, tol=tol, maxit=5000, omp=True), 3)
            gdt, gr = gpu_cg(A, 0, tol, 5000)
            o[f"tol{tol:g}"] = {"cpu_iterations": r["iterations"], "cpu_seconds": round(dt, 4), "cpu_it_per_s": round(r["iterations"] / dt, 1),
                               "gpu_iterations": gr["iterations"], "gpu_seconds": round(gdt, 5), "gpu_it_per_s": round(gr["iterations"] / gdt, 1),
                               "cpu_rel_res": r["rel_res_norm"], "gpu_rel_res": gr["rel_res_norm"]}
        print(json.dumps(o), flush=True)
        del A
    if "c2" in which:
        rp, ci, v = oracle.stencil_csr(7, 256, 256, 256)
        n = rp.size - 1
        b, x = np.ones(n), np.ones(n)
        A = mk.Matrix.stencil(7, 256, 256, 256)
        o = dict(head, config="c2: 3D 7-pt Laplacian 256^3, Jacobi-preconditioned CG (solver 51), 50 iterations", rows=n, nnz=int(rp[-1]))
        o["cpu_spmv_ms"] = round(_median_time(lambda: oracle.csr_spmv(rp, ci, v, x, omp=True), 5)[0] * 1e3, 3)
        o["gpu_spmv_ms"] = round(gpu_spmv_ms(A), 5)
        dt, r = _median_time(lambda: oracle.pcg(rp, ci, v, b, precond=1, tol=0.0, maxit=50, omp=True), 3)
        gdt, gr = gpu_cg(A, 1, 0.0, 50)
        o.update(cpu_it_per_s=round(50 / dt, 2), gpu_it_per_s=round(50 / gdt, 1), cpu_rel_res=r["rel_res_norm"], gpu_rel_res=gr["rel_res_norm"])
        print(json.dumps(o), flush=True)
        del A
    if "c3" in which:
        scale, ne = 24, 1 << 28
        t0 = time.perf_counter()
        rows, cols, vals = oracle.rmat_coo(scale, ne, seed=42)
        gen_s = time.perf_counter() - t0
        nr = 1 << scale
        o = dict(head, config="c3: R-MAT scale 24, 2^28 entries, seed 42 (unsorted COO, duplicates kept): COO->CSR + SpMV", rows=nr, nnz=ne,
                 cpu_generate_s=round(gen_s, 1))
        if oracle.have_ref():
            t0 = time.perf_counter()
            rrp, rci, rv = oracle.ref_coo_to_csr(nr, nr, rows, cols, vals)
            o["cpu_reference_convert_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            o["cpu_reference_convert_kind"] = "reference: convert_coo_to_csr of sparse_matrix.c, compiled into oracle/_ref, serial"
        t0 = time.perf_counter()
        rp, ci, v = oracle.coo_to_csr(nr, rows, cols, vals)
        o["cpu_oracle_convert_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        if oracle.have_ref():
            o["reference_equals_oracle"] = bool(np.array_equal(rrp, rp) and np.array_equal(rci, ci) and np.array_equal(rv, v))
            del rrp, rci, rv
        del rows, cols, vals
        x = np.ones(nr)
        o["cpu_spmv_ms"] = round(_median_time(lambda: oracle.csr_spmv(rp, ci, v, x, omp=True), 5)[0] * 1e3, 2)
        ts = []
        for rep in range(4):
            coo = mk.DeviceCOO.rmat(scale, ne, seed=42)
            mk.synchronize()
            mk.timer_start()
            A = mk.Matrix.from_device_coo(coo)
            ts.append(mk.timer_stop())
            if rep < 3:
                del A
            del coo
        o["gpu_convert_ms"] = round(float(np.median(ts[1:])), 3)
        o["gpu_spmv_ms"] = round(gpu_spmv_ms(A, 10), 4)
        o["gpu_spmv_kernel"] = KERNEL_NAMES.get(A.info().kernel)
        y = mk.Vector(nr)
        A.spmv(mk.Vector(nr, fill=1.0), y)
        yc = oracle.csr_spmv(rp, ci, v, x, omp=True)
        scale_y = oracle.csr_spmv(rp, ci, np.abs(v), x, omp=True)           # |A| |x|: the rounding scale of each row
        o["gpu_vs_cpu_spmv_max_err_over_absA_absx"] = float(np.max(np.abs(y.get() - yc) / np.maximum(scale_y, 1e-300)))
        print(json.dumps(o), flush=True)


# ------------------------------------------------------------------------------ starting the ranks ourselves
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(ngpus, argv):
    """`python bench.py --gpus N` without a launcher: this parent (which never touches the GPU) starts N children of the
    same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's line, and returns the worst exit code"""
    import tempfile
    port = _free_port()
    procs, logs = [], []
    with tempfile.TemporaryDirectory(prefix="mk_bench_") as tmp:
        for r in range(ngpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpus), LOCAL_WORLD_SIZE=str(ngpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            # stdout / stderr of every rank go to files: no pipe can fill up while this parent polls
            fo, fe = open(os.path.join(tmp, f"out{r}"), "w+"), open(os.path.join(tmp, f"err{r}"), "w+")
            logs.append((fo, fe))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=fo, stderr=fe))
        # Poll ALL children: a rank that dies (bad device, out of memory, RCCL init failure) leaves the others blocked in the
        # rendezvous or in a collective for ever.  On the first non-zero exit the rest are terminated (they are fresh
        # children of this parent, nothing is re-executed) and the run fails with that rank's stderr tail.
        failed, rc = None, 0
        while failed is None and any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = r
                    break
            else:
                time.sleep(0.05)
        if failed is None:
            for r, p in enumerate(procs):
                if p.returncode != 0 and failed is None:
                    failed = r
        if failed is not None:
            rc = abs(procs[failed].returncode) or 1
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.time() + 10.0
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            fe = logs[failed][1]
            fe.flush(); fe.seek(0)
            tail = fe.read()[-4000:]
            sys.stderr.write(f"bench.py: rank {failed} of {ngpus} exited with code {procs[failed].returncode}; the other ranks were "
                             f"terminated.  Its stderr tail:\n{tail}\n")
        else:
            fo = logs[0][0]
            fo.flush(); fo.seek(0)
            sys.stdout.write(fo.read())
            sys.stdout.flush()
            for r, (_, fe) in enumerate(logs):      # warnings of the ranks stay visible
                fe.flush(); fe.seek(0)
                err = fe.read()
                if err.strip():
                    sys.stderr.write(err if r == 0 else "".join(f"[rank {r}] {ln}\n" for ln in err.splitlines()[-20:]))
        for fo, fe in logs:
            fo.close(); fe.close()
    return rc


def dry_run_rank(args, rank, world):
    """MK_BENCH_DRY_SPAWN=1: rehearsal of the launch logic on CPU (gloo): rendezvous, id broadcast, barrier, one line"""
    import torch
    import torch.distributed as dist
    from iterative_solvers_amd import distributed as mkd
    dist.init_process_group("gloo")
    if os.environ.get("MK_BENCH_DRY_FAIL_RANK") == str(rank):   # test: one rank dies after the rendezvous, the others wait in the barrier
        sys.stderr.write(f"dry run: rank {rank} fails on purpose\n")
        sys.stderr.flush()
        os._exit(7)
    ident = mkd.broadcast_bytes(bytes(range(256)) if rank == 0 else None, 256, 0)
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "max_rank": t.item(), "id_ok": ident == bytes(range(256)),
                          "steps": args.steps, "warmup": args.warmup}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=10)
    ap.add_argument("--cpu-configs", default=None, help="comma list of c1,c2,c3: per-config CPU baselines beside the GPU (own mode)")
    args = ap.parse_args()

    # MK_BENCH_FORCE_DIST=1 with --gpus 1 rehearses the whole self-launched multi-rank path (spawn, NCCL rendezvous,
    # RCCL communicators of the library, row-block solve) on a one-GPU box
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("MK_BENCH_FORCE_DIST")):
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "MK_BENCH_DEVICE" in os.environ:     # rehearsal of the N>1 path on a box with fewer GPUs
        local_rank = int(os.environ["MK_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("MK_BENCH_DRY_SPAWN"):
        return dry_run_rank(args, rank, world)

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

    if args.cpu_configs:
        return cpu_configs(mk, args.cpu_configs.split(","))

    traffic_db = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic_db = json.load(open(tpath))

    solve_times = {}
    spmv_only = {}

    def measure(name, steps, warmup, timing=True):
        """CG on workload `name`: a warm-up solve, the timed solve (wall clock around mk_pcg_solve, stream drained on
        both sides), then one more solve with HIP events around every kernel of iterations 2..33"""
        kind, nx, ny, nz, precond, seed, desc = WORKLOADS[name]
        A = mk.Matrix.stencil(kind, nx, ny, nz, scale_seed=seed)
        n = A.nrows
        b, x = mk.Vector(n, fill=1.0), mk.Vector(n, fill=0.0)

        def run(iters, timed_kernels=False):
            S = mk.PCG(max_iter=iters, tol=0.0, two_norm=1, rel_change=1, precond=precond, logging=0, check_every=50)
            if timed_kernels:
                S.set_kernel_timing(True)
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
            return dt, res, (S.kernel_timing() if timed_kernels else None)

        run(max(warmup, 1))
        runs = [run(steps) for _ in range(TIMED_SOLVES)]          # EXACTLY `steps` iterations each; the median solve is reported
        runs.sort(key=lambda t: t[0])
        dt, res, _ = runs[len(runs) // 2]
        solve_times[name] = [round(t[0] * 1e3 / steps, 4) for t in runs]
        kt = run(34, True)[2] if timing else None
        if timing and A.info().kernel == 6:
            # the plain product + dots of the same operator (mk_spmv_dot: MODE 1 of the kernel, the first step's K1), 20 launches
            # between two HIP events on the library's stream
            xs, ys = mk.Vector(n, fill=1.0), mk.Vector(n)
            for _ in range(3):
                A.spmv_dot(xs, ys, want=False)
            mk.timer_start()
            for _ in range(20):
                A.spmv_dot(xs, ys, want=False)
            spmv_only[name] = mk.timer_stop() / 20
            del xs, ys
        return A, dt, res, kt

    def kernel_blocks(name, A, kt):
        """one roofline block per kernel slot of the CG step, from the in-loop HIP-event averages"""
        info = A.info()
        n, nnz = A.nrows, A.nnz
        precond = WORKLOADS[name][4]
        kname = KERNEL_NAMES.get(info.kernel, str(info.kernel))
        csr = 12 * nnz + 20 * n + 4
        fused = info.kernel == 5 and not os.environ.get("MK_NO_FUSED_CG")
        if fused:
            # fused step on sweep matrices: first sweep p' = z + beta p and the dots of A p' (2 + 8 + 8 + 8 B/row),
            # second sweep r -= alpha A p' recomputed, x += alpha p' (2 + 8 + 16 + 16 B/row); K3 rides in the first
            halfx = not os.environ.get("MK_NO_HALVED_X") and not (kt is not None and len(kt) == 3)   # 
            slots = [("spmv_sweep_kernel MODE 5 (p' = z + beta p in the window fill, dots of A p')", 26 * n, "26 B/row: 2 B pattern id, r 8, p 8, p' 8"),
                     ("spmv_sweep_kernel MODE 7 / 8 alternating (r -= alpha A p recomputed, dots; the solution update x += alpha_prev p_prev + alpha p every second step)",
                      34 * n, "34 B/row on average: 2 B pattern id, p 8, r 8 + 8, and x 8 + 8 in every second step (26 B and 42 B)") if halfx else
                     ("spmv_sweep_kernel MODE 4 (r -= alpha A p recomputed, x += alpha p, dots)", 42 * n, "42 B/row: 2 B pattern id, p 8, r 8 + 8, x 8 + 8")]
            if kt is not None and len(kt) == 3:
                slots.append(("pcg_update_p_kernel (K3)", 24 * n, "24 B/row"))
        elif info.kernel == 6 and kt is not None and len(kt) == 2 and not os.environ.get("MK_NO_FUSED_K3"):
            # variable-coefficient grids: K3 (p' = z + beta p) and the solution update ride in the window fill of
            # the product; K2 streams s and r only.  Bytes by SURVEY 8(d)'s own accounting of the passes each kernel carries:
            # SpMV 12 nnz + 20 n + 4, direction pass 24 n, update pass 48 n.
            defx = not os.environ.get("MK_NO_DEFERRED_X") and not precond
            k1b = csr + 24 * n + (24 * n if defx else 0) + (8 * n if precond else 0)
            k2b = ((24 if defx else 48) + (8 if precond else 0)) * n
            slots = [("spmv_vsweep_kernel MODE 3 (K1 + K3" + (" + x update" if defx else "") + ": p' = z + beta p in the window fill, s = A p', <s,p'>, <p',p'>"
                      + (", x += alpha p_old" if defx else "") + ")", k1b,
                      "SURVEY 8(d): SpMV 12 nnz + 20 n + 4, direction pass 24 n" + (", the x half of the update pass (x read + write, p read) 24 n" if defx else "")
                      + (", 1/diag 8 n" if precond else "") + f"; the kernel itself moves 8 nnz + {36 + (16 if defx else 0) + (8 if precond else 0)} n "
                      "(no col[], p' and p_old read once)"),
                     ("pcg_update_r_kernel (K2 without x: r -= alpha s, <r,z>, <r,r>)" if defx else "pcg_update_xr_kernel (K2)", k2b,
                      "SURVEY 8(d): the r half of the update pass (r read + write, s read) 24 n" + (", 1/diag 8 n" if precond else "") if defx
                      else "x, r read+write, p, s read" + (", 1/diag" if precond else ""))]
        else:
            own = csr if info.kernel in (1, 2, 3, 6) else 18 * n + 4
            slots = [(f"spmv+dot, {kname} kernel (K1: s = A p, <s,p>, <p,p>)", own,
                      "SURVEY 8(d): 12 nnz + 20 n + 4" if own == csr else "18 B/row: 2 B pattern id + 8 B x + 8 B y"),
                     ("pcg_update_xr_kernel (K2)", (56 if precond else 48) * n, "x, r read+write, p, s read" + (", 1/diag" if precond else "")),
                     ("pcg_update_p_kernel (K3)", (32 if precond else 24) * n, "r, p read, p write" + (", 1/diag" if precond else ""))]
        out = []
        for k, (label, nbytes, model) in enumerate(slots):
            ms, cnt = (kt[k] if kt is not None and k < len(kt) else (0.0, 0))
            gbps = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None
            tr = traffic_db.get(f"{name}:slot{k}")
            out.append({"kernel": label, "bound": "hbm", "achieved": round(gbps, 1) if gbps else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbps / HBM_PEAK_GBS, 4) if gbps else None, "bytes_per_launch": nbytes, "bytes_model": model,
                        "avg_launch_ms": round(ms, 5), "launches_timed": cnt, "traffic": tr,
                        "traffic_frac": round(tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr and ms > 0 else None})
        return out, kname, csr

    name = args.workload or "lap7_512_cg"
    kind, nx, ny, nz, precond, seed, desc = WORKLOADS[name]
    A, dt, res, kt = measure(name, args.steps, args.warmup)
    n, nnz = A.nrows, A.nnz
    info = A.info()
    ms_per_step = dt * 1e3 / args.steps
    blocks, kname, csr_bytes = kernel_blocks(name, A, kt)
    vec_bytes = (88 if precond else 72) * n
    own_iter_bytes = sum(b["bytes_per_launch"] for b in blocks)
    out = {
        "metric": "CG iterations/s (fp64, fused SpMV+dot+axpy) with SpMV effective HBM GB/s",
        "value": round(args.steps / dt, 2), "unit": "CG iterations/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": desc, "name": name, "rows": n, "nnz": nnz, "precond": "jacobi" if precond else "none",
                   "spmv_kernel": kname, "index_patterns": info.index_patterns, "value_patterns": info.value_patterns,
                   "parallelism": "1 GPU"},
        "final_rel_res_norm": res["rel_res_norm"],
        "timing": {"rule": f"median of {TIMED_SOLVES} timed solves of exactly `steps` iterations each after one warm-up solve of `warmup` "
                           "iterations, x0 reset before every solve (SURVEY.md 8(d))", "ms_per_step_of_each_solve": solve_times[name]},
        # the loop `value` times, kernel by kernel, on the bytes those kernels move (NOT the CSR model: on constant-
        # coefficient grids they read a 2-byte pattern id per row and none of the CSR arrays)
        "roofline_value_loop": {"note": "kernels of the timed CG loop, durations by HIP events inside mk_pcg_solve "
                                        "(iterations 2..33 of a separate 34-iteration solve on the same stream)",
                                "kernels": blocks,
                                "iteration": {"bytes": own_iter_bytes, "achieved": round(own_iter_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                              "frac": round(own_iter_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                              "traffic": traffic_db.get(f"{name}:iteration"),
                                              "csr_model_bytes": csr_bytes + vec_bytes,
                                              "csr_model_GBps": round((csr_bytes + vec_bytes) / (ms_per_step * 1e-3) / 1e9, 1)}},
    }
    del A

    # ---- the roofline of the line: a CG loop whose dominant kernel reads the CSR value array.  Same grid, variable
    # coefficients (unless the workload asked for already is one).
    vname = name if seed is not None else ("lap7_512_var_cg" if (kind, nx, ny, nz) == (7, 512, 512, 512) else None)
    if vname is not None:
        if vname == name:
            vblocks, vk, vcsr, vdt, vn, vnnz, vinfo = blocks, kname, csr_bytes, dt, n, nnz, info
        else:
            A2, vdt, vres, vkt = measure(vname, args.steps, args.warmup)
            vblocks, vk, vcsr = kernel_blocks(vname, A2, vkt)
            vn, vnnz, vinfo = A2.nrows, A2.nnz, A2.info()
            del A2
        k1 = vblocks[0]
        vms = vdt * 1e3 / args.steps
        vprec = WORKLOADS[vname][4]
        viter = vcsr + (88 if vprec else 72) * vn      # 12 nnz + 92 n (+16 n with Jacobi)
        out["roofline"] = {
            "bound": "hbm", "workload": WORKLOADS[vname][6], "name": vname,
            "kernel": k1["kernel"] + " -- the dominant kernel of the CG loop on this operator; reads val[] (8 B/entry), "
                      "a 2-byte column-pattern id per row in place of col[] (lossless), 2 B of row pointer per row",
            "achieved": k1["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k1["frac"],
            "bytes_per_launch": k1["bytes_per_launch"], "bytes_model": k1["bytes_model"],
            "avg_launch_ms": k1["avg_launch_ms"], "launches_timed": k1["launches_timed"],
            "measured": "HIP events around the kernel inside mk_pcg_solve, on the library's stream",
            "traffic": k1["traffic"], "traffic_frac": k1["traffic_frac"],
            "target_0_75_met": bool(k1["frac"] is not None and k1["frac"] >= 0.75),
            "spmv_only": None if vname not in spmv_only else {
                "kernel": "spmv_vsweep_kernel MODE 1: the plain product + dots of the same operator (what the loop's first step runs)",
                "avg_launch_ms": round(spmv_only[vname], 5), "launches_timed": 20, "bytes_per_launch": vcsr,
                "achieved": round(vcsr / (spmv_only[vname] * 1e-3) / 1e9, 1), "frac": round(vcsr / (spmv_only[vname] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": traffic_db.get(f"{vname}:spmv_only")},
            "index_patterns": vinfo.index_patterns, "value_patterns": vinfo.value_patterns,
            "cg": {"value": round(args.steps / vdt, 2), "unit": "CG iterations/s", "ms_per_step": round(vms, 4),
                   "iteration_bytes": viter, "iteration_GBps": round(viter / (vms * 1e-3) / 1e9, 1),
                   "iteration_frac": round(viter / (vms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "ms_per_step_of_each_solve": solve_times[vname], "kernels": vblocks},
        }
    else:
        k1 = blocks[0]
        out["roofline"] = {"bound": "hbm", "kernel": k1["kernel"], "achieved": k1["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": k1["frac"], "traffic": k1["traffic"], "bytes_per_launch": k1["bytes_per_launch"],
                           "bytes_model": k1["bytes_model"], "avg_launch_ms": k1["avg_launch_ms"]}

    if args.workload is None and not args.no_secondary:
        sec = {}
        # BASELINE.json configs[1]: 256^3, Jacobi-preconditioned CG, one GPU -- constant and variable coefficients
        for sname in ("lap7_256_jacobi_pcg", "lap7_256_var_jacobi_pcg"):
            A2, dt2, res2, kt2 = measure(sname, args.steps, args.warmup)
            b2, k2, csr2 = kernel_blocks(sname, A2, kt2)
            sec[sname] = {"workload": WORKLOADS[sname][6], "value": round(args.steps / dt2, 2), "unit": "CG iterations/s",
                          "ms_per_step": round(dt2 * 1e3 / args.steps, 4), "spmv_kernel": k2,
                          "iteration_csr_model_GBps": round((12 * A2.nnz + 108 * A2.nrows) / (dt2 / args.steps) / 1e9, 1),
                          "kernel_ms": [b["avg_launch_ms"] for b in b2]}
            del A2
        # BASELINE.json configs[3]'s stencil with real coefficients on one GPU's share of rows (2^24 rows, 449 M entries): rows of
        # 27 entries on the vsweep kernel's one-row-per-lane form; product + dots timed alone, then the CG loop (K1 + K2 + K3)
        try:
            sname = "lap27_256_var_cg"
            A2, dt2, res2, kt2 = measure(sname, args.steps, args.warmup)
            x2, y2 = mk.Vector(A2.nrows, fill=1.0), mk.Vector(A2.nrows)
            for _ in range(3):
                A2.spmv_dot(x2, y2, want=False)
            mk.timer_start()
            for _ in range(20):
                A2.spmv_dot(x2, y2, want=False)
            ms2 = mk.timer_stop() / 20
            sb2 = 12 * A2.nnz + 20 * A2.nrows + 4
            sec[sname] = {"workload": WORKLOADS[sname][6], "value": round(args.steps / dt2, 2), "unit": "CG iterations/s",
                          "ms_per_step": round(dt2 * 1e3 / args.steps, 4), "spmv_kernel": KERNEL_NAMES.get(A2.info().kernel),
                          "iteration_frac": round((12 * A2.nnz + 92 * A2.nrows) / (dt2 / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                          "spmv_dot_ms": round(ms2, 4), "spmv_frac": round(sb2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "spmv_bytes_model": "SURVEY 8(d): 12 nnz + 20 n + 4", "kernel_ms": [round(k[0], 4) for k in kt2]}
            del A2, x2, y2
        except Exception as e:      # never lose the headline to a secondary measurement
            sec["lap27_256_var_cg"] = {"error": repr(e)}
        # BASELINE.json configs[2]: R-MAT scale 24, 2^28 edges: COO -> CSR on the device, then SpMV (split kernel)
        try:
            scale, ne = 24, 1 << 28
            coo = mk.DeviceCOO.rmat(scale, ne, seed=42)
            mk.synchronize()
            mk.timer_start()
            A3 = mk.Matrix.from_device_coo(coo)
            conv_ms = mk.timer_stop()
            n3, nnz3 = A3.nrows, A3.nnz
            x3, y3 = mk.Vector(n3, fill=1.0), mk.Vector(n3)
            mk.timer_start()
            A3.spmv(x3, y3)          # the first product also builds the hot-first column relabelling (one-off)
            first_ms = mk.timer_stop()
            for _ in range(2):
                A3.spmv(x3, y3)
            mk.timer_start()
            for _ in range(10):
                A3.spmv(x3, y3)
            spmv_ms = mk.timer_stop() / 10
            cbytes, sbytes = 32 * nnz3 + 12 * n3, 12 * nnz3 + 20 * n3 + 4
            sec["rmat_s24"] = {"workload": "R-MAT scale 24, 2^28 edges (a,b,c = 0.57,0.19,0.19; duplicates kept): COO->CSR + SpMV",
                               "coo_to_csr_ms": round(conv_ms, 3), "coo_to_csr_frac": round(cbytes / (conv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "coo_to_csr_bytes_model": "SURVEY 8(d): 32 nnz + 12 n",
                               "first_spmv_ms_with_one_off_analysis": round(first_ms, 3),
                               "spmv_ms": round(spmv_ms, 4), "spmv_frac": round(sbytes / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "spmv_kernel": KERNEL_NAMES.get(A3.info().kernel), "max_row_len": A3.info().max_row_len}
            del A3, x3, y3, coo
        except mk.MkError as e:      # never lose the headline to a secondary measurement
            sec["rmat_s24"] = {"error": str(e)}
        # finite-element-like CSR matrices (the shape of the reference's SuiteSparse inputs, scripts/bench_all_gpu:2), built on the
        # host by tools/fem_probe.py's generators: three unknowns per node of a 64^3 hexahedral mesh (81 entries per row, index
        # patterns) and three unknowns per node with random neighbour nodes (no patterns; rows of a node share their column list)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import fem_probe
            import numpy as np
            # ... and rows that share nothing but stay within 3000 columns of the diagonal (band kernel: sliding x window in LDS)
            for fname, (fn, rp, ci) in (("fem_hex3_64", fem_probe.hex3(64)), ("fem_rand3_500k", fem_probe.rand3(500000, 26, 300)),
                                        ("fem_rand_band_600k", fem_probe.rand(600000, 45, 3000))):
                vals = np.random.default_rng(0).normal(size=ci.size)
                A4 = mk.Matrix.from_csr(fn, fn, rp, ci, vals)
                i4 = A4.info()
                x4, y4 = mk.Vector(fn, fill=1.0), mk.Vector(fn)
                for _ in range(3):
                    A4.spmv(x4, y4)
                mk.timer_start()
                for _ in range(20):
                    A4.spmv(x4, y4)
                ms4 = mk.timer_stop() / 20
                b4 = 12 * ci.size + 20 * fn + 4
                sec[fname] = {"rows": fn, "nnz": int(ci.size), "mean_row_len": round(ci.size / fn, 1), "max_row_len": i4.max_row_len,
                              "spmv_ms": round(ms4, 4), "spmv_frac": round(b4 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "spmv_bytes_model": "SURVEY 8(d): 12 nnz + 20 n + 4", "spmv_kernel": KERNEL_NAMES.get(i4.kernel),
                              "index_patterns": i4.index_patterns, "shared_column_entries": i4.shared_columns,
                              "row_groups": i4.row_groups, "group_lanes": i4.group_lanes, "band_window": i4.band_window, "band_strips": i4.band_strips,
                              "stream_lanes": i4.stream_lanes, "stream_chunk": i4.stream_chunk}
                del A4, x4, y4, vals
        except Exception as e:      # never lose the headline to a secondary measurement
            sec["fem_like"] = {"error": repr(e)}
        out["secondary"] = sec
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(kind, nx, ny, nz, precond, args.cpu_iters)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
