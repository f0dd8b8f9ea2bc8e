"""Row-block distribution: ctypes front-end of the mk_comm_* / mk_dist_* entry points and the
multi-GPU leg of bench.py.  torch.distributed is used for the rendezvous only (broadcast of the
RCCL unique id, barriers, max-over-ranks timing); halo exchange and all-reduce run inside
libmi355_krylov.so on RCCL communicators of its own."""
import ctypes as C
import json
import os
import time

import numpy as np

from . import (MkError, Matrix, PCG, Vector, _ck, init, load, synchronize)  # noqa: F401

_sigs_done = False


def _lib():
    global _sigs_done
    L = load()
    if not _sigs_done:
        P = C.c_void_p
        L.mk_comm_get_unique_id.argtypes = [P]
        L.mk_comm_create.argtypes = [P, C.c_int, C.c_int, C.POINTER(P)]
        L.mk_comm_create_local.argtypes = [C.c_int, C.POINTER(P)]
        L.mk_comm_destroy.argtypes = [P]
        L.mk_comm_self_test.argtypes = [P]
        L.mk_comm_all_reduce_host.argtypes = [P, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), C.c_int, C.c_int]
        L.mk_dist_matrix_create.argtypes = [P, C.c_int64, P, P, C.POINTER(P)]
        L.mk_dist_matrix_destroy.argtypes = [P]
        L.mk_dist_matrix_get_info.argtypes = [P] + [C.POINTER(C.c_int64)] * 5 + [C.POINTER(C.c_int)]
        L.mk_dist_matrix_set_spmv_kernel.argtypes = [P, C.c_int, C.c_int]
        L.mk_dist_vector_create.argtypes = [P, C.POINTER(P)]
        L.mk_dist_spmv.argtypes = [C.c_int, C.POINTER(P), C.POINTER(P), C.POINTER(P)]
        L.mk_dist_pcg_solve.argtypes = [C.c_int, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P)]
        L.mk_dist_jacobi_solve.argtypes = [C.c_int, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P)]
        L.mk_dist_bicgstab_solve.argtypes = [C.c_int, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P)]
        L.mk_dist_gmres_solve.argtypes = [C.c_int, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P)]
        L.mk_dist_cgnr_solve.argtypes = [C.c_int, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P)]
        _sigs_done = True
    return L


DIST_EXPORTS = ["mk_comm_get_unique_id", "mk_comm_create", "mk_comm_create_local", "mk_comm_self_test", "mk_comm_all_reduce_host", "mk_comm_destroy",
                "mk_dist_matrix_create", "mk_dist_matrix_destroy", "mk_dist_matrix_get_info", "mk_dist_matrix_set_spmv_kernel",
                "mk_dist_vector_create", "mk_dist_spmv", "mk_dist_pcg_solve", "mk_dist_jacobi_solve",
                "mk_dist_bicgstab_solve", "mk_dist_gmres_solve", "mk_dist_cgnr_solve"]


# ------------------------------------------------------------------ host-side helpers (no GPU)
def slab_row_start(nx, ny, nz, world):
    """Row blocks for a lexicographic nx*ny*nz stencil: the nnz-balanced rule of solvers.c:214-231
    gives (to within a row) equal z-slabs, so cut at whole planes."""
    plane = nx * ny
    return np.array([(nz * r // world) * plane for r in range(world + 1)], dtype=np.int64)


def even_row_start(n, world):
    return np.array([n * r // world for r in range(world + 1)], dtype=np.int64)


def broadcast_bytes(payload, nbytes, src=0):
    """ship a small byte string from rank src to every rank through torch.distributed"""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    if dist.get_rank() == src:
        t.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(t, src)
    return bytes(t.cpu().numpy().tobytes())


# ------------------------------------------------------------------------------- wrappers
class Comm:
    def __init__(self, handle, nranks, rank, local=False):
        self.h, self.nranks, self.rank, self.local = handle, nranks, rank, local

    @classmethod
    def from_torch(cls):
        """one rank per process: rank 0 draws the RCCL ids, torch.distributed ships them"""
        import torch.distributed as dist
        L = _lib()
        rank, world = dist.get_rank(), dist.get_world_size()
        buf = C.create_string_buffer(256)
        if rank == 0:
            _ck(L.mk_comm_get_unique_id(buf))
        ident = broadcast_bytes(buf.raw if rank == 0 else None, 256, 0)
        h = C.c_void_p()
        _ck(L.mk_comm_create(ident, world, rank, C.byref(h)))
        return cls(h, world, rank)

    @classmethod
    def single_rccl(cls):
        """a one-rank RCCL communicator (drives the RCCL code path on a single GPU)"""
        init()
        L = _lib()
        buf = C.create_string_buffer(256)
        _ck(L.mk_comm_get_unique_id(buf))
        h = C.c_void_p()
        _ck(L.mk_comm_create(buf.raw, 1, 0, C.byref(h)))
        return cls(h, 1, 0)

    def self_test(self):
        _ck(_lib().mk_comm_self_test(self.h))

    def all_reduce_host(self, values, op="sum"):
        """MPI_Allreduce of a few host doubles (solvers.c:760 takes the max of the solve times this way)"""
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        _ck(_lib().mk_comm_all_reduce_host(self.h, v, v.size, 1 if op == "max" else 0))
        return v

    @classmethod
    def local_world(cls, nranks):
        init()
        arr = (C.c_void_p * nranks)()
        _ck(_lib().mk_comm_create_local(nranks, arr))
        return [cls(C.c_void_p(arr[r]), nranks, r, True) for r in range(nranks)]

    def __del__(self):
        if getattr(self, "h", None):
            load().mk_comm_destroy(self.h)
            self.h = None


class DistMatrix:
    def __init__(self, block, n_global, row_start, comm):
        """block: Matrix holding this rank's rows with global column ids (consumed)"""
        rs = np.ascontiguousarray(row_start, dtype=np.int64)
        self.h = C.c_void_p()
        _ck(_lib().mk_dist_matrix_create(block.h, n_global, rs.ctypes.data, comm.h if comm else None, C.byref(self.h)))
        block.h = None      # ownership moved into the library
        self.comm, self.row_start, self.n_global = comm, rs, n_global

    def diag_info(self):
        from . import MatrixInfo
        i = MatrixInfo()
        _ck(_lib().mk_dist_matrix_get_diag_info(self.h, C.byref(i)))
        return i

    def info(self):
        v = [C.c_int64() for _ in range(5)]
        k = C.c_int()
        _ck(_lib().mk_dist_matrix_get_info(self.h, *[C.byref(a) for a in v], C.byref(k)))
        return dict(n_local=v[0].value, n_ghost=v[1].value, n_offd_rows=v[2].value, nnz_diag=v[3].value,
                    nnz_offd=v[4].value, n_neighbors=k.value)

    def set_spmv_kernel(self, kernel, lanes_per_row=0):
        _ck(_lib().mk_dist_matrix_set_spmv_kernel(self.h, kernel, lanes_per_row))

    def vector(self, fill=None, values=None):
        v = Vector.__new__(Vector)
        v.h = C.c_void_p()
        _ck(_lib().mk_dist_vector_create(self.h, C.byref(v.h)))
        v.n = int(load().mk_vector_size(v.h))
        if values is not None:
            v.set(values)
        elif fill is not None:
            v.fill(fill)
        return v

    def __del__(self):
        if getattr(self, "h", None):
            load().mk_dist_matrix_destroy(self.h)
            self.h = None


def _arr(objs):
    a = (C.c_void_p * len(objs))()
    for i, o in enumerate(objs):
        a[i] = o.h
    return a


def dist_spmv(As, xs, ys):
    _ck(_lib().mk_dist_spmv(len(As), _arr(As), _arr(xs), _arr(ys)))


def dist_pcg_solve(solvers, As, bs, xs):
    L = _lib()
    _ck(L.mk_dist_pcg_solve(len(As), _arr(solvers), _arr(As), _arr(bs), _arr(xs)))
    out = []
    for S in solvers:
        it, nrm, cv, bd = C.c_longlong(), C.c_double(), C.c_int(), C.c_int()
        _ck(L.mk_pcg_get_num_iterations(S.h, C.byref(it)))
        _ck(L.mk_pcg_get_final_relative_residual_norm(S.h, C.byref(nrm)))
        _ck(L.mk_pcg_get_converged(S.h, C.byref(cv), C.byref(bd)))
        out.append(dict(iterations=it.value, rel_res_norm=nrm.value, converged=cv.value, breakdown=bd.value))
    return out


def dist_jacobi_solve(solvers, As, bs, xs):
    L = _lib()
    _ck(L.mk_dist_jacobi_solve(len(As), _arr(solvers), _arr(As), _arr(bs), _arr(xs)))
    out = []
    for S in solvers:
        it, nrm, cv = C.c_longlong(), C.c_double(), C.c_int()
        _ck(L.mk_jacobi_get_num_iterations(S.h, C.byref(it)))
        _ck(L.mk_jacobi_get_final_relative_residual_norm(S.h, C.byref(nrm)))
        _ck(L.mk_jacobi_get_converged(S.h, C.byref(cv)))
        out.append(dict(iterations=it.value, rel_res_norm=nrm.value, converged=cv.value))
    return out


def dist_bicgstab_solve(solvers, As, bs, xs):
    L = _lib()
    _ck(L.mk_dist_bicgstab_solve(len(As), _arr(solvers), _arr(As), _arr(bs), _arr(xs)))
    out = []
    for S in solvers:
        it, nrm, cv, bd = C.c_longlong(), C.c_double(), C.c_int(), C.c_int()
        _ck(L.mk_bicgstab_get_num_iterations(S.h, C.byref(it)))
        _ck(L.mk_bicgstab_get_final_relative_residual_norm(S.h, C.byref(nrm)))
        _ck(L.mk_bicgstab_get_converged(S.h, C.byref(cv), C.byref(bd)))
        out.append(dict(iterations=it.value, rel_res_norm=nrm.value, converged=cv.value, breakdown=bd.value))
    return out


def dist_gmres_solve(solvers, As, bs, xs):
    L = _lib()
    _ck(L.mk_dist_gmres_solve(len(As), _arr(solvers), _arr(As), _arr(bs), _arr(xs)))
    out = []
    for S in solvers:
        it, nrm, cv = C.c_longlong(), C.c_double(), C.c_int()
        _ck(L.mk_gmres_get_num_iterations(S.h, C.byref(it)))
        _ck(L.mk_gmres_get_final_relative_residual_norm(S.h, C.byref(nrm)))
        _ck(L.mk_gmres_get_converged(S.h, C.byref(cv)))
        out.append(dict(iterations=it.value, rel_res_norm=nrm.value, converged=cv.value))
    return out


def dist_cgnr_solve(solvers, As, bs, xs):
    L = _lib()
    _ck(L.mk_dist_cgnr_solve(len(As), _arr(solvers), _arr(As), _arr(bs), _arr(xs)))
    out = []
    for S in solvers:
        it, nrm, cv = C.c_longlong(), C.c_double(), C.c_int()
        _ck(L.mk_cgnr_get_num_iterations(S.h, C.byref(it)))
        _ck(L.mk_cgnr_get_final_relative_residual_norm(S.h, C.byref(nrm)))
        _ck(L.mk_cgnr_get_converged(S.h, C.byref(cv)))
        out.append(dict(iterations=it.value, rel_res_norm=nrm.value, converged=cv.value))
    return out


# --------------------------------------------------------------------- bench.py, N > 1 ranks
def bench_main(args, rank, local_rank, world, workloads, hbm_peak):
    """strong scaling of one global stencil problem over `world` GPUs (one rank each)"""
    import torch
    import torch.distributed as dist
    name = args.workload or "lap7_512_cg"
    kind, nx, ny, nz, precond, seed, desc = workloads[name]
    n = nx * ny * nz
    rs = slab_row_start(nx, ny, nz, world)
    comm = Comm.from_torch()
    comm.self_test()
    block = Matrix.stencil(kind, nx, ny, nz, row_begin=int(rs[rank]), row_end=int(rs[rank + 1]), scale_seed=seed)
    local_nnz = block.nnz
    A = DistMatrix(block, n, rs, comm)
    info = A.info()
    b, x = A.vector(fill=1.0), A.vector(fill=0.0)

    def run(iters, single=False):
        S = PCG(max_iter=iters, tol=0.0, two_norm=1, rel_change=1, precond=precond, logging=0, check_every=50)
        if single:
            S.set_single_reduction(True)
        x.fill(0.0)
        synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        res = dist_pcg_solve([S], [A], [b], [x])[0]
        synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert res["iterations"] == iters, res
        res["single_reduction"] = S.single_reduction_used()
        return float(t.item()), res

    # Warm-up (untimed): both forms of the CG step -- hypre_PCGSolve's two reductions per iteration (the library's default) and
    # the single-reduction step (mk_pcg_set_single_reduction: one all-reduce of five doubles, 52 instead of 68 B per row) -- then
    # a short calibration solve of each; the timed solves use the faster one (the times are maxima over the ranks, so every
    # rank picks the same).  MK_BENCH_CG_VARIANT=two|single skips the choice.
    want = os.environ.get("MK_BENCH_CG_VARIANT", "auto")
    run(max(args.warmup, 1), False)
    calib = {}
    use_single = False
    if want != "two":
        _, r1 = run(max(args.warmup, 1), True)
        if r1["single_reduction"]:          # (declined when a rank's diag block is not on the sweep kernel)
            if want == "single":
                use_single = True
            else:
                k = max(10, min(args.steps, 50))
                calib["two_reductions_ms_per_step"] = round(run(k, False)[0] * 1e3 / k, 4)
                calib["single_reduction_ms_per_step"] = round(run(k, True)[0] * 1e3 / k, 4)
                use_single = calib["single_reduction_ms_per_step"] < calib["two_reductions_ms_per_step"]
    solves = [run(args.steps, use_single) for _ in range(5)]
    solves.sort(key=lambda q: q[0])
    dt, res = solves[2]
    nnz_t = torch.tensor([local_nnz], dtype=torch.int64, device="cuda")
    dist.all_reduce(nnz_t)
    nnz = int(nnz_t.item())
    dinfo = A.diag_info()
    kname = {1: "vector", 2: "stream", 3: "split", 4: "pattern", 5: "sweep", 6: "vsweep"}.get(dinfo.kernel, str(dinfo.kernel))
    if rank == 0:
        ms = dt * 1e3 / args.steps
        csr_bytes = 12 * nnz + (108 if precond else 92) * n
        # rank 0's diag block decides the byte model reported (every slab of a stencil takes the same kernel): the fused
        # sweep step moves 68 B per row and iteration (+8 with a non-uniform Jacobi diagonal), the CSR kernels the CSR model
        fused = kname == "sweep"
        iter_bytes = ((60 if os.environ.get("MK_NO_HALVED_X") == "1" else 52) if use_single else 68) * n if fused else csr_bytes
        model = (("52 B/row: two sweeps per iteration (10 B: <A r, r>; 34 / 50 B: p' = z + beta p, r -= alpha A p', and every second step x += both steps' terms; 60 B/row with MK_NO_HALVED_X=1), A p' recomputed instead of "
                  "stored; halo and off-rank rows not counted" if use_single else
                  "68 B/row: two sweeps per iteration (26 B + 42 B), A p recomputed instead of stored; halo and off-rank rows not counted")
                 if fused else "CSR model: 12 nnz + 92 n (+16 n with Jacobi)")
        gbps = iter_bytes / (ms * 1e-3) / 1e9
        out = {
            "metric": "CG iterations/s (fp64, fused SpMV+dot+axpy) with SpMV effective HBM GB/s",
            "value": round(args.steps / dt, 2), "unit": "CG iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "name": name, "rows": n, "nnz": nnz, "precond": "jacobi" if precond else "none",
                       "spmv_kernel": kname, "index_patterns": dinfo.index_patterns, "value_patterns": dinfo.value_patterns,
                       "parallelism": f"row-block x{world} (z-slabs), RCCL halo + all-reduce",
                       "cg_step": "single reduction per iteration (mk_pcg_set_single_reduction)" if use_single else "two reductions per iteration (hypre_PCGSolve's order)",
                       "cg_step_calibration": calib,
                       "rank0": info},
            "timing": {"rule": "median of 5 timed solves of exactly `steps` iterations each (maximum over the ranks per solve), x0 reset before every solve",
                       "ms_per_step_of_each_solve": [round(q[0] * 1e3 / args.steps, 4) for q in solves]},
            "iteration_effective_GBps": round(gbps, 1),
            "iteration_csr_model_GBps": round(csr_bytes / (ms * 1e-3) / 1e9, 1),
            "roofline": {"bound": "hbm", "kernel": "whole CG iteration, all GPUs (wall clock over ranks, communication included)",
                         "achieved": round(gbps, 1), "peak": hbm_peak * world, "unit": "GB/s",
                         "frac": round(gbps / (hbm_peak * world), 4), "traffic": None, "bytes_model": model,
                         "csr_model_bytes": csr_bytes, "csr_model_GBps": round(csr_bytes / (ms * 1e-3) / 1e9, 1),
                         "csr_model_frac": round(csr_bytes / (ms * 1e-3) / 1e9 / (hbm_peak * world), 4)},
            "final_rel_res_norm": res["rel_res_norm"],
        }
        print(json.dumps(out), flush=True)
    dist.barrier()
    del A, b, x
    dist.destroy_process_group()
