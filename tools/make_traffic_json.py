#!/usr/bin/env python3
"""Turn the PMC passes of tools/r03_profiles.sh (gpurun_out/<tag>_pmc_*) into profiles/<tag>_pmc_cg_loops.json,
profiles/<tag>_pmc_convert_rmat.json, profiles/<tag>_pmc_row_group.json and the per-kernel HBM bytes bench.py reports as
`traffic` (profiles/traffic.json).  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB: MI355X_MICROARCH.md, HBM section (on gfx950
FETCH_SIZE counts half of a wide coalesced read stream).  usage: make_traffic_json.py [tag]   (default r03)"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
# kernel slot k of the CG step bench.py times on each workload (first match wins), plus the stand-alone product
# (a slot may name several kernels that take turns in it: the halved solution update alternates MODE 7 and MODE 8; their
# per-launch bytes are averaged over all their launches)
SLOTS = {"lap7_512_cg": ["spmv_sweep_kernel<5", ("spmv_sweep_kernel<7", "spmv_sweep_kernel<8", "spmv_sweep_kernel<4")],
         "lap7_512_var_cg": ["spmv_vsweep_kernel<3", "pcg_update_r_kernel"]}
EXTRA = {"lap7_512_var_cg": {"spmv_only": "spmv_vsweep_kernel<1"}}


def passes(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(ROOT, pattern)):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def summary(d):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    hbm = int((2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024)
    return n, dict(launches=n, FETCH_SIZE_KB=round(m.get("FETCH_SIZE", 0), 1), WRITE_SIZE_KB=round(m.get("WRITE_SIZE", 0), 1),
                   TCC_REQ=int(m.get("TCC_REQ_sum", 0)), TCC_HIT=int(m.get("TCC_HIT_sum", 0)), TCC_MISS=int(m.get("TCC_MISS_sum", 0)),
                   hbm_bytes_per_launch=hbm)


out = {"what": f"rocprofv3 --pmc, one pass per counter group (FETCH_SIZE | WRITE_SIZE | TCC_REQ/HIT/MISS), on tools/cgk_probe.py 512x512x512 "
               "(constant coefficients: fused sweep step; seed 1: variable coefficients: vsweep MODE 3 = product + K3 + deferred x update, "
               "pcg_update_r_kernel; MODE 1 = the stand-alone product + dots); HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "kernels": {}}
traffic = {}
multi = {}
for tag, name in (("512x512x512_0_0", "lap7_512_cg"), ("512x512x512_1_0", "lap7_512_var_cg")):
    agg = passes(f"gpurun_out/{TAG}_pmc_{tag}_*/**/pmc_counter_collection.csv") or passes(f"gpurun_out/{TAG}_pmc_{tag}_*/pmc_counter_collection.csv")
    total = 0
    for k, d in sorted(agg.items()):
        if not any(s in k for s in ("sweep", "pcg_update")):
            continue
        n, rec = summary(d)
        if n < 10:
            continue
        out["kernels"][f"{name}: {k}"] = rec
        for s, pref in enumerate(SLOTS[name]):
            if isinstance(pref, tuple):
                if any(q in k for q in pref):
                    multi.setdefault((name, s), []).append((rec["launches"], rec["hbm_bytes_per_launch"]))
            elif pref in k:
                traffic[f"{name}:slot{s}"] = rec["hbm_bytes_per_launch"]
                total += rec["hbm_bytes_per_launch"]
        for key, pref in EXTRA.get(name, {}).items():
            if pref in k:
                traffic[f"{name}:{key}"] = rec["hbm_bytes_per_launch"]
    for (nm, s), lst in multi.items():
        if nm == name:
            avg = int(sum(a * b for a, b in lst) / max(1, sum(a for a, _ in lst)))
            traffic[f"{name}:slot{s}"] = avg
            total += avg
    traffic[f"{name}:iteration"] = total
traffic["_note"] = ("HBM bytes per launch of kernel slot k of the CG step and per iteration, = (2*FETCH_SIZE + WRITE_SIZE) KB from separate "
                    f"rocprofv3 --pmc passes (tools/{TAG}_profiles.sh; raw values in profiles/{TAG}_pmc_cg_loops.json)")
n, nnz = 512 ** 3, 7 * 512 ** 3 - 6 * 512 ** 2
out["algorithmic_bytes"] = {"SURVEY 8(d) SpMV 12 nnz + 20 n + 4": 12 * nnz + 20 * n + 4, "SURVEY 8(d) CG iteration 12 nnz + 92 n": 12 * nnz + 92 * n,
                            "vsweep MODE 1 moves 8 nnz + 20 n": 8 * nnz + 20 * n, "vsweep MODE 3 with the x update moves 8 nnz + 52 n": 8 * nnz + 52 * n,
                            "pcg_update_r_kernel 24 n": 24 * n, "sweep mode 5: 26 n": 26 * n, "sweep mode 4 / 8: 42 n": 42 * n, "sweep mode 7: 26 n": 26 * n}
json.dump(out, open(os.path.join(ROOT, f"profiles/{TAG}_pmc_cg_loops.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(ROOT, "profiles/traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))

for what, pat, keep, dest, note in (
        ("convert", f"gpurun_out/{TAG}_pmc_convert_*", ("radix", "hist", "scan", "rowptr", "validate", "split", "gather_x", "relabel", "degree"),
         f"profiles/{TAG}_pmc_convert_rmat.json", "tools/rmat_probe.py: R-MAT scale 24, 2^28 entries: COO->CSR on the device, then SpMV (split kernel)"),
        ("group", f"gpurun_out/{TAG}_pmc_group_*", ("spmv_group", "spmv_stream"),
         f"profiles/{TAG}_pmc_row_group.json", "tools/group_probe.py 500000 26 300: FE-like matrix (3 unknowns per node, random neighbour nodes; "
                                              "bench.py's secondary.fem_rand3_500k) through the row-group kernel"),
        ("vsweep27", f"gpurun_out/{TAG}_pmc_vsweep27_*", ("spmv_vsweep", "spmv_stream"),
         f"profiles/{TAG}_pmc_vsweep_27pt.json", "MK_PROBE_KIND=27 tools/vsweep_probe.py 256x256x256: 27-point operator with variable coefficients "
                                                "(2^24 rows, 449 M entries) through the long-row form of the vsweep kernel (one row per lane) and, for "
                                                "comparison, the stream kernel; algorithmic bytes 12 nnz + 20 n + 4 = 5.72 GB, the vsweep kernel moves 8 nnz + 22 n + 4 n")):
    agg = passes(pat + "/**/pmc_counter_collection.csv") or passes(pat + "/pmc_counter_collection.csv")
    o = {"what": "rocprofv3 --pmc, one pass per counter group, on " + note + "; HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": {}}
    for k, d in sorted(agg.items()):
        if any(s in k for s in keep):
            o["kernels"][k] = summary(d)[1]
    if what == "convert":
        nn, ne = 1 << 24, 1 << 28
        o["algorithmic_bytes"] = {"COO->CSR 32 nnz + 12 n": 32 * ne + 12 * nn, "SpMV 12 nnz + 20 n + 4": 12 * ne + 20 * nn + 4}
        conv = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in o["kernels"].items() if any(s in k for s in ("radix", "hist", "scan", "rowptr", "validate")))
        o["conversion_hbm_bytes_all_kernels_of_the_probe"] = conv
        o["hbm_over_algorithmic"] = round(conv / (32 * ne + 12 * nn), 2)
    if o["kernels"]:
        json.dump(o, open(os.path.join(ROOT, dest), "w"), indent=1)
        print(dest, "written:", list(o["kernels"])[:6])
