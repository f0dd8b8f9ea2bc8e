#!/usr/bin/env python3
"""SpMV on synthetic FEM-like matrices (the shape of the authors' SuiteSparse inputs, scripts/bench_all_gpu:2 of the
reference: Bump_2911, Serena, ldoor -- 40-80 entries per row, every entry its own value), built on the host with numpy
and uploaded as CSR.  Two generators:
  hex3 <nodes_per_side>            3 dof per node on a hexahedral mesh: 27 node neighbours x 3x3 blocks, up to 81 per row
  nine <side>                      2-D 9-point stencil
  rand3 <nodes> <mean_nb> <halfwin> 3 unknowns per node, random neighbour nodes within +-halfwin, full 3x3 blocks
  skew <rows> <mean_len> <every> <long_len>  short rows with one long row every so many rows
  rand <rows> <mean_len> <halfwin> row lengths uniform in mean*[2/3,4/3], columns in blocks of 3 anywhere in +-halfwin
Prints the time of every eligible kernel and the fraction of the CSR byte model (12 nnz + 20 n + 4) over 8 TB/s.
usage: fem_probe.py [hex3 64 | rand 2000000 45 3000] ...   (several cases may follow each other)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iterative_solvers_amd as mk


def hex3(N):
    nn = N ** 3
    idx = np.arange(nn)
    ix, iy, iz = idx % N, (idx // N) % N, idx // (N * N)
    rr, cc = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = (ix + dx >= 0) & (ix + dx < N) & (iy + dy >= 0) & (iy + dy < N) & (iz + dz >= 0) & (iz + dz < N)
                rr.append(idx[ok])
                cc.append(idx[ok] + dx + dy * N + dz * N * N)
    nrow, ncol = np.concatenate(rr), np.concatenate(cc)
    rows = (nrow[:, None, None] * 3 + np.arange(3)[None, :, None] + np.zeros((1, 1, 3), np.int64)).reshape(-1)
    cols = (ncol[:, None, None] * 3 + np.arange(3)[None, None, :] + np.zeros((1, 3, 1), np.int64)).reshape(-1)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order].astype(np.int32)
    n = 3 * nn
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    return n, rp, cols


def rand(n, mean, hw):
    rng = np.random.default_rng(1)
    lens = rng.integers(mean * 2 // 3, mean * 4 // 3 + 1, n)
    rp = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(rp[-1])
    rows = np.repeat(np.arange(n), lens)
    base = rows + rng.integers(-hw, hw + 1, nnz)
    ci = np.clip(base - base % 3 + (np.arange(nnz) % 3), 0, n - 1).astype(np.int32)
    ci = ci[np.lexsort((ci, rows))]
    return n, rp.astype(np.int32), ci


def randfat(n, mean, hw, every, fat):
    """`rand` with one row of `fat` entries every `every` rows (constraint rows among regular ones)"""
    rng = np.random.default_rng(4)
    lens = rng.integers(mean * 2 // 3, mean * 4 // 3 + 1, n)
    lens[every // 2::every] = fat
    rp = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(rp[-1])
    rows = np.repeat(np.arange(n), lens)
    ci = np.clip(rows + rng.integers(-hw, hw + 1, nnz), 0, n - 1).astype(np.int32)
    ci = ci[np.lexsort((ci, rows))]
    return n, rp.astype(np.int32), ci


def rand3(nodes, mean_nb, hw):
    """3 unknowns per node; every node couples to itself and ~mean_nb random nodes within +-hw, as full 3x3 blocks: the three
    rows of a node share their column list, no two nodes share a pattern"""
    rng = np.random.default_rng(2)
    k = rng.integers(mean_nb * 2 // 3, mean_nb * 4 // 3 + 1, nodes)
    src = np.repeat(np.arange(nodes), k)
    dst = np.clip(src + rng.integers(-hw, hw + 1, src.size), 0, nodes - 1)
    pairs = np.unique(np.concatenate([src, np.arange(nodes)]) * nodes + np.concatenate([dst, np.arange(nodes)]))   # sorted, no duplicates, diagonal in
    src, dst = pairs // nodes, pairs % nodes
    cnt = np.bincount(src, minlength=nodes)                     # node neighbours per node
    nrp = np.concatenate([[0], np.cumsum(cnt)])
    # row 3 i + d holds, for every neighbour node j in order, columns 3 j, 3 j + 1, 3 j + 2
    rows_len = np.repeat(cnt * 3, 3)
    rp = np.concatenate([[0], np.cumsum(rows_len)]).astype(np.int64)
    node_cols = (dst[:, None] * 3 + np.arange(3)[None, :]).reshape(-1)          # per node: its column list
    idx = np.repeat(np.arange(nodes), 3)                                        # row -> node
    starts = 3 * nrp[idx]
    take = np.repeat(starts, rows_len) + (np.arange(rp[-1]) - np.repeat(rp[:-1], rows_len))
    ci = node_cols[take].astype(np.int32)
    return 3 * nodes, rp.astype(np.int32), ci


def skew(n, mean, every, long_len):
    """short rows (about `mean` entries near the diagonal) with one row of `long_len` entries every `every` rows: the shape of
    circuit matrices (scircuit, hvdc1/2 of the authors' list)"""
    rng = np.random.default_rng(3)
    lens = rng.integers(max(1, mean // 2), mean * 3 // 2 + 1, n)
    lens[::every] = long_len
    rp = np.concatenate([[0], np.cumsum(lens)])
    rows = np.repeat(np.arange(n), lens)
    ci = np.clip(rows + rng.integers(-2000, 2001, rows.size), 0, n - 1).astype(np.int32)
    ci = ci[np.lexsort((ci, rows))]
    return n, rp.astype(np.int32), ci


def run(name, n, rp, ci):
    nnz = ci.size
    v = np.random.default_rng(0).normal(size=nnz)
    A = mk.Matrix.from_csr(n, n, rp, ci, v)
    mk.synchronize(); t0 = time.perf_counter()
    i = A.info()          # first use: row statistics, kernel choice, index-pattern search
    mk.synchronize(); t_an = (time.perf_counter() - t0) * 1e3
    x, y = mk.Vector(n, fill=1.0), mk.Vector(n)
    b = 12 * nnz + 20 * n + 4
    out = [f"{name}: rows {n} mean {nnz / n:.1f} max {i.max_row_len} default {i.kernel} patterns {i.index_patterns} lanes {i.stream_lanes} chunk {i.stream_chunk} shared {i.shared_columns / max(nnz, 1):.2f} groups {i.row_groups} glanes {i.group_lanes} window {i.band_window} strips {i.band_strips} analysis {t_an:.2f} ms |"]
    cases = [(0, A, "k0"), (1, A, "k1"), (2, A, "k2"), (3, A, "k3")]
    if i.band_window > 0:   # the same matrix without the band plan: the stream kernel proper
        os.environ["MK_NO_BAND"] = "1"
        B = mk.Matrix.from_csr(n, n, rp, ci, v)
        B.info()
        del os.environ["MK_NO_BAND"]
        cases.append((2, B, "k2-noband"))
    for k, A, label in cases:
        try:
            A.set_spmv_kernel(k)
        except mk.MkError:
            continue
        beta = float(os.environ.get("FEM_PROBE_BETA", "0"))   # (y = A x + beta y: the accumulating form reads y as well)
        for _ in range(3):
            A.spmv(x, y, 1.0, beta)
        mk.timer_start()
        for _ in range(20):
            A.spmv(x, y, 1.0, beta)
        ms = mk.timer_stop() / 20
        out.append(f"{label} {ms * 1e3:.0f} us {b / ms / 1e6 / 8000:.3f}")
    print(" ".join(out), flush=True)


def nine(N):
    """2-D 9-point stencil on an N x N grid, every entry its own value"""
    idx = np.arange(N * N)
    ix, iy = idx % N, idx // N
    rr, cc = [], []
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            ok = (ix + dx >= 0) & (ix + dx < N) & (iy + dy >= 0) & (iy + dy < N)
            rr.append(idx[ok]); cc.append(idx[ok] + dx + dy * N)
    rows, cols = np.concatenate(rr), np.concatenate(cc)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order].astype(np.int32)
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=N * N))]).astype(np.int32)
    return N * N, rp, cols


def main():
    args = sys.argv[1:] or ["hex3", "64"]
    mk.init(0)
    while args:
        if args[0] == "nine":
            run(f"nine {args[1]}", *nine(int(args[1]))); args = args[2:]
        elif args[0] == "skew":
            run(f"skew {args[1]} {args[2]} {args[3]} {args[4]}", *skew(*(int(a) for a in args[1:5]))); args = args[5:]
        elif args[0] == "rand3":
            run(f"rand3 {args[1]} {args[2]} {args[3]}", *rand3(int(args[1]), int(args[2]), int(args[3]))); args = args[4:]
        elif args[0] == "randfat":
            run(f"randfat {' '.join(args[1:6])}", *randfat(*(int(a) for a in args[1:6]))); args = args[6:]
        elif args[0] == "hex3":
            run(f"hex3 {args[1]}", *hex3(int(args[1]))); args = args[2:]
        else:
            run(f"rand {args[1]} {args[2]} {args[3]}", *rand(int(args[1]), int(args[2]), int(args[3]))); args = args[4:]


if __name__ == "__main__":
    main()
