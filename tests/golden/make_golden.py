#!/usr/bin/env python3
"""Generate tests/golden/*.mtx inputs and *.npz expected outputs.

The expected outputs come from the REFERENCE ITSELF: /root/reference/src/sparse_matrix.c +
mmio.c compiled by oracle/Makefile into oracle/_ref/libref_sparse.so and called through
ctypes (fast_load_from_mtx_file, sort_coo_row, convert_coo_to_csr).  Run in the build
container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

Fixtures are data: small Matrix Market inputs written by this script and the arrays the
reference produced for them.  No reference source text is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402


def write_mtx(path, nrows, ncols, entries, field="real", sym="general", comment=None):
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate {field} {sym}\n")
        if comment:
            f.write(f"% {comment}\n")
        f.write(f"{nrows} {ncols} {len(entries)}\n")
        for e in entries:
            if field == "pattern":
                f.write(f"{e[0] + 1} {e[1] + 1}\n")
            else:
                f.write(f"{e[0] + 1} {e[1] + 1} {e[2]!r}\n")


def cases():
    rng = np.random.default_rng(20211105)
    out = {}
    # 1. general, row-sorted, columns ascending: 2D 5-point Poisson 6x5 (config-1 shape, tiny)
    rp, ci, v = oracle.stencil_csr(5, 6, 5)
    ent = [(r, int(ci[k]), float(v[k])) for r in range(30) for k in range(rp[r], rp[r + 1])]
    out["poisson5_6x5_sorted"] = (30, 30, ent, "real", "general")
    # 2. same entries shuffled (stability: order inside a row must follow file order)
    perm = rng.permutation(len(ent))
    out["poisson5_6x5_shuffled"] = (30, 30, [ent[i] for i in perm], "real", "general")
    # 3. duplicates kept + empty rows + single entry in the last row + rectangular
    ent3 = []
    for _ in range(60):
        r = int(rng.choice([0, 1, 2, 4, 5, 9, 9, 9, 12]))
        ent3.append((r, int(rng.integers(0, 11)), float(np.round(rng.normal(), 6))))
    ent3 += [(4, 3, 1.25), (4, 3, -2.5), (4, 3, 1.25)]          # exact duplicates
    ent3 = [ent3[i] for i in rng.permutation(len(ent3))]
    ent3.append((13, 10, 7.0))                                   # last row: one entry, last line
    out["ragged_dups_empty_rows"] = (14, 11, ent3, "real", "general")
    # 4. symmetric real, lower triangle given in column-major order as SuiteSparse files are
    n = 9
    ent4 = []
    for c in range(n):
        for r in range(c, n):
            if r == c or rng.random() < 0.35:
                ent4.append((r, c, float(np.round(4.0 if r == c else -rng.random(), 5))))
    out["symmetric_lower"] = (n, n, ent4, "real", "symmetric")
    # 5. pattern general
    ent5 = [(int(rng.integers(0, 7)), int(rng.integers(0, 7))) for _ in range(25)]
    out["pattern_general"] = (7, 7, ent5, "pattern", "general")
    # 6. pattern symmetric
    ent6 = []
    for c in range(8):
        for r in range(c, 8):
            if r == c or rng.random() < 0.3:
                ent6.append((r, c))
    out["pattern_symmetric"] = (8, 8, ent6, "pattern", "symmetric")
    # 7. one row holds most entries (skewed), reverse order
    ent7 = [(3, c, float(c + 1)) for c in range(40)][::-1] + [(0, 0, 1.0), (7, 39, 2.0), (3, 5, -9.0)]
    out["skewed_one_long_row"] = (8, 40, ent7, "real", "general")
    # 8. a matrix with no entries at all
    out["empty_matrix"] = (5, 5, [], "real", "general")
    # 9. 1x1
    out["one_by_one"] = (1, 1, [(0, 0, 3.5)], "real", "general")
    # 10. larger random (multi-digit radix path on the device: 70000 rows > 2^16)
    nr = 70000
    m = 40000
    rr = rng.integers(0, nr, m)
    cc = rng.integers(0, nr, m)
    vv = np.round(rng.normal(size=m), 4)
    out["random_70k"] = (nr, nr, [(int(rr[i]), int(cc[i]), float(vv[i])) for i in range(m)], "real", "general")
    # 11. the sort-by-key scratch input the reference holds (test/array_sorting.cpp:35-36): ten (row key, value) pairs in file
    #     order; the column is the pair's position, so the sorted output shows the stable order directly
    keys = [3, 4, 2, 2, 1, 1, 0, 0, 2, 4]
    vals = [1.2, 22.3, 6.4, 1.0052, 0.22, 6.1, 7.02, 52.0, 1.2, 10.0]
    out["array_sorting_kat"] = (5, 10, [(keys[i], i, vals[i]) for i in range(10)], "real", "general")
    return out


def main():
    oracle.build(force=True)
    assert oracle.have_ref(), "needs /root/reference (build container)"
    for name, (nr, nc, ent, field, sym) in cases().items():
        mtx = os.path.join(HERE, name + ".mtx")
        if os.path.exists(mtx) and os.path.exists(os.path.join(HERE, name + ".npz")) and "--all" not in sys.argv:
            continue   # committed fixtures stay byte-identical; --all regenerates every one
        write_mtx(mtx, nr, nc, ent, field, sym, comment=f"fixture {name} (tests/golden/make_golden.py)")
        nrows, ncols, rows, cols, vals = oracle.ref_load_mtx(mtx)
        sr, sc, sv = oracle.ref_sort_coo_row(nrows, ncols, rows, cols, vals)
        rp, ci, cv = oracle.ref_coo_to_csr(nrows, ncols, rows, cols, vals)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), nrows=nrows, ncols=ncols,
                            coo_rows=rows, coo_cols=cols, coo_vals=vals,
                            sorted_rows=sr, sorted_cols=sc, sorted_vals=sv,
                            csr_rowptr=rp, csr_col=ci, csr_val=cv)
        print(f"{name}: {nrows}x{ncols} nnz={rows.size}")
    # SpMV known-answer input held by the reference (test/sblas_mv_rd.c:15-32); the file
    # carries no expected output, y below is derived by hand from those arrays.
    np.savez(os.path.join(HERE, "sblas_mv_kat.npz"),
             aval=np.array([1.1, 1.2, 2.2, 2.3, 3.3, 3.4, 3.5, 3.7, 4.1, 4.4, 5.3, 5.5, 5.6]),
             iaind=np.array([0, 1, 1, 2, 2, 3, 4, 6, 0, 3, 2, 4, 5], np.int32),
             iaptr=np.array([0, 2, 4, 8, 10, 13], np.int32),
             x=np.array([1.0, 2, 3, 4, 5, 6, 7]),
             y=np.array([1.1 * 1 + 1.2 * 2, 2.2 * 2 + 2.3 * 3, 3.3 * 3 + 3.4 * 4 + 3.5 * 5 + 3.7 * 7,
                         4.1 * 1 + 4.4 * 4, 5.3 * 3 + 5.5 * 5 + 5.6 * 6]))


if __name__ == "__main__":
    main()
