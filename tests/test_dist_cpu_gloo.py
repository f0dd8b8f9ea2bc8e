"""world_size-2 (and 3) gloo tests on CPU of the multi-rank HOST path of the library: rendezvous, broadcast of the
communicator id bytes, and the exchange planning that mk_dist_matrix_create runs (include/mk_krylov.h mk_halo_*:
ghost list, owner lookup, receive / send counts, row_start checks).  Every rank plans with the LIBRARY's functions,
the ranks then exchange exactly what the plan says over gloo, and the assembled row-block product must equal the
oracle's global product bit for bit.  Without mk_halo_* (or with a wrong plan) the test fails; the device half
(diag/offd split, pack kernel, RCCL calls) is covered by tests/test_dist_gpu.py."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _plan_lib():
    import iterative_solvers_amd as mk
    L = mk.load()
    i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    L.mk_halo_check_row_start.argtypes = [C.c_int, i64, C.c_int64]
    L.mk_halo_ghost_columns.argtypes = [C.c_int64, i32, i32, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.mk_halo_plan_recv.argtypes = [C.c_int, C.c_int, i64, i32, C.c_int64, i32, i32]
    L.mk_halo_plan_send.argtypes = [C.c_int, C.c_int, i32, i32, i32]
    return mk, L


def _worker(rank, world, port, case, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mk, L = _plan_lib()
    from iterative_solvers_amd import distributed as mkd
    ok = True
    # 1. id bytes travel from rank 0 to all ranks unchanged
    got = mkd.broadcast_bytes(bytes(range(256)) if rank == 0 else None, 256, 0)
    ok &= got == bytes(range(256))
    # 2. the matrix and its nnz-balanced row blocks (mk_partition_rows; every rank computes the same cuts)
    if case == "stencil":
        rp, ci, v = oracle.stencil_csr(7, 10, 9, 8)
        v = v * np.random.default_rng(1).uniform(0.5, 1.5, v.size)
    else:
        r, c, vv = oracle.rmat_coo(10, 1 << 14, seed=5)
        rp, ci, v = oracle.coo_to_csr(1 << 10, r, c, vv)
    n = rp.size - 1
    rs = mk.partition_rows(rp, world)
    ok &= L.mk_halo_check_row_start(world, rs, n) == 0
    ok &= L.mk_halo_check_row_start(world, rs[::-1].copy(), n) != 0
    rs_all = [torch.zeros(world + 1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(rs_all, torch.from_numpy(rs.copy()))
    ok &= all(torch.equal(t, rs_all[0]) for t in rs_all)
    lo, hi = int(rs[rank]), int(rs[rank + 1])
    brp = np.ascontiguousarray(rp[lo:hi + 1] - rp[lo], dtype=np.int32)
    bci = np.ascontiguousarray(ci[rp[lo]:rp[hi]], dtype=np.int32)
    bv = v[rp[lo]:rp[hi]]
    # 3. the library's plan: ghosts, receive counts by owner, send counts from the all-gathered table
    ng = C.c_int64()
    ok &= L.mk_halo_ghost_columns(hi - lo, brp, bci, lo, hi, n, None, 0, C.byref(ng)) == 0
    ghosts = np.zeros(max(ng.value, 1), np.int32)
    ok &= L.mk_halo_ghost_columns(hi - lo, brp, bci, lo, hi, n, ghosts.ctypes.data, ghosts.size, C.byref(ng)) == 0
    ghosts = ghosts[:ng.value]
    recv_cnt, recv_off = np.zeros(world, np.int32), np.zeros(world + 1, np.int32)
    ok &= L.mk_halo_plan_recv(world, rank, rs, ghosts, ghosts.size, recv_cnt, recv_off) == 0
    table = [torch.zeros(world, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(table, torch.from_numpy(recv_cnt.copy()))
    recv_all = np.ascontiguousarray(torch.stack(table).numpy(), dtype=np.int32)
    send_cnt, send_off = np.zeros(world, np.int32), np.zeros(world + 1, np.int32)
    ok &= L.mk_halo_plan_send(world, rank, recv_all.reshape(-1), send_cnt, send_off) == 0
    # 4. exchange as planned: the ids each owner must send (what build_plan_rccl does with ncclSend/Recv) ...
    send_ids = [np.zeros(int(send_cnt[q]), np.int32) for q in range(world)]
    reqs = []
    for q in range(world):
        if recv_cnt[q]:
            reqs.append(dist.isend(torch.from_numpy(ghosts[recv_off[q]:recv_off[q + 1]].copy()), q))
        if send_cnt[q]:
            reqs.append(dist.irecv(torch.from_numpy(send_ids[q]), q))
    for rq in reqs:
        rq.wait()
    ok &= all(((s >= lo) & (s < hi)).all() for s in send_ids)          # I am asked for my own rows only
    # ... then the halo of x: pack by the send list, receive into the ghost tail (halo_exchange in dist.hip)
    x_full = np.random.default_rng(0).normal(size=n)
    x_ext = np.concatenate([x_full[lo:hi], np.full(ghosts.size, np.nan)])
    bufs = [np.ascontiguousarray(x_full[s]) for s in send_ids]
    reqs = []
    for q in range(world):
        if send_cnt[q]:
            reqs.append(dist.isend(torch.from_numpy(bufs[q]), q))
        if recv_cnt[q]:
            reqs.append(dist.irecv(torch.from_numpy(x_ext[(hi - lo) + recv_off[q]:(hi - lo) + recv_off[q + 1]]), q))
    for rq in reqs:
        rq.wait()
    ok &= not np.isnan(x_ext).any()
    # 5. the block product in local numbering (owned columns first, ghosts behind in sorted order) against the oracle
    own = (bci >= lo) & (bci < hi)
    lci = np.where(own, bci - lo, (hi - lo) + np.searchsorted(ghosts, bci)).astype(np.int32)
    y_loc = oracle.csr_spmv(brp, lci, bv, x_ext) if hi > lo else np.empty(0)
    ys = [None] * world
    dist.all_gather_object(ys, y_loc)
    ok &= np.array_equal(np.concatenate(ys), oracle.csr_spmv(rp, ci, v, x_full))
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "stencil"), (3, "rmat")])
def test_gloo_ranks_exchange_by_the_library_plan(world, case):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, case, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world))


def test_plan_functions_reject_bad_input():
    mk, L = _plan_lib()
    rs = np.array([0, 4, 8], np.int64)
    ghosts = np.array([1, 5], np.int32)                    # 5 belongs to rank 1 itself
    cnt, off = np.zeros(2, np.int32), np.zeros(3, np.int32)
    assert L.mk_halo_plan_recv(2, 1, rs, ghosts, 2, cnt, off) != 0 and b"own rows" in mk.load().mk_last_error_string()
    assert L.mk_halo_plan_recv(2, 0, rs, np.array([4, 9], np.int32), 2, cnt, off) != 0        # 9 is outside every range
    assert L.mk_halo_plan_recv(2, 0, rs, np.array([4, 7], np.int32), 2, cnt, off) == 0 and cnt.tolist() == [0, 2] and off.tolist() == [0, 0, 2]
    assert L.mk_halo_check_row_start(2, np.array([0, 9, 8], np.int64), 8) != 0
    assert L.mk_halo_check_row_start(2, np.array([1, 4, 8], np.int64), 8) != 0
    rp, ci = np.array([0, 2], np.int32), np.array([0, 12], np.int32)
    ng = C.c_int64()
    assert L.mk_halo_ghost_columns(1, rp, ci, 0, 1, 8, None, 0, C.byref(ng)) != 0               # column 12 of 8
