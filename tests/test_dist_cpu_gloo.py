"""world_size-2 gloo test on CPU of the multi-rank host path: rendezvous, broadcast of the
communicator id bytes, the row-block partition, and the row-block SpMV semantics (each rank
multiplies its own rows after gathering the x entries of the other rank)."""
import os
import socket

import numpy as np
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


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import iterative_solvers_amd as mk
    from iterative_solvers_amd import distributed as mkd
    # 1. id bytes travel from rank 0 to all ranks unchanged
    payload = bytes(range(256)) if rank == 0 else None
    got = mkd.broadcast_bytes(payload, 256, 0)
    ok = got == bytes(range(256))
    # 2. every rank computes the same partition with the library's host partitioner
    rp, ci, v = oracle.stencil_csr(7, 10, 10, 10)
    n = rp.size - 1
    rs = mk.partition_rows(rp, world)
    rs_all = [torch.zeros(world + 1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(rs_all, torch.from_numpy(rs.copy()))
    ok = ok and all(torch.equal(t, rs_all[0]) for t in rs_all)
    ok = ok and np.array_equal(mkd.slab_row_start(10, 10, 10, world), [0, 500, 1000])
    # 3. row-block product: own rows, global columns, x assembled from all ranks
    lo, hi = int(rs[rank]), int(rs[rank + 1])
    x_full = np.random.default_rng(0).normal(size=n)
    parts = [torch.zeros(int(rs[q + 1] - rs[q]), dtype=torch.float64) for q in range(world)]
    dist.all_gather(parts, torch.from_numpy(x_full[lo:hi].copy())) if len(set(p.numel() for p in parts)) == 1 else None
    if len(set(p.numel() for p in parts)) != 1:           # uneven blocks: gather object-wise
        objs = [None] * world
        dist.all_gather_object(objs, x_full[lo:hi])
        parts = [torch.from_numpy(np.asarray(o)) for o in objs]
    x_g = np.concatenate([p.numpy() for p in parts])
    y_loc = oracle.csr_spmv(rp[lo:hi + 1] - rp[lo], ci[rp[lo]:rp[hi]], v[rp[lo]:rp[hi]], x_g)
    ys = [None] * world
    dist.all_gather_object(ys, y_loc)
    y = np.concatenate(ys)
    ok = ok and np.array_equal(y, oracle.csr_spmv(rp, ci, v, x_full))
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_two_rank_gloo_host_path():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] and out[1]
