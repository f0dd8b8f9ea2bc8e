#!/usr/bin/env python3
"""Turn the PMC passes of tools/r02_convert_profiles.sh (gpurun_out/r02_pmc_convert_*) into profiles/r02_pmc_convert_rmat.json:
per kernel of BASELINE configs[2] (R-MAT scale 24: COO->CSR, then the split-kernel SpMV) the launches, counter averages and
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB (MI355X_MICROARCH.md, HBM section), next to SURVEY 8(d)'s bytes."""
import collections, csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(ROOT, "gpurun_out/r02_pmc_convert_*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for r in csv.DictReader(open(os.path.join(ROOT, "gpurun_out/r02_convert_prof/convert_kernel_stats.csv"))):
    dur[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
nnz, n = 1 << 28, 1 << 24
out = {"what": "rocprofv3 --pmc, one pass per counter group, on tools/rmat_probe.py (R-MAT scale 24, 2^28 entries, seed 42); durations from the "
               "--kernel-trace --stats run of the same program (profiles/r02_convert_kernel_stats.csv). total_* = sum over the launches of one run.",
       "algorithmic_bytes": {"COO->CSR, SURVEY 8(d) 32 nnz + 12 n": 32 * nnz + 12 * n, "SpMV, SURVEY 8(d) 12 nnz + 20 n + 4": 12 * nnz + 20 * n + 4},
       "kernels": {}}
conv_hbm = conv_ms = 0.0
for k, d in sorted(agg.items()):
    if not k.startswith("mk::"):
        continue
    cnt = len(next(iter(d.values())))
    tot = {c: sum(v) for c, v in d.items()}
    hbm_total = (2 * tot.get("FETCH_SIZE", 0) + tot.get("WRITE_SIZE", 0)) * 1024
    calls, ms = dur.get(k, (cnt, 0.0))
    out["kernels"][k] = dict(launches=cnt, total_ms=round(ms, 3), avg_ms=round(ms / max(calls, 1), 4), FETCH_SIZE_KB_total=round(tot.get("FETCH_SIZE", 0), 1),
                             WRITE_SIZE_KB_total=round(tot.get("WRITE_SIZE", 0), 1), TCC_REQ=int(tot.get("TCC_REQ_sum", 0)), TCC_HIT=int(tot.get("TCC_HIT_sum", 0)),
                             TCC_MISS=int(tot.get("TCC_MISS_sum", 0)), hbm_bytes_total=int(hbm_total), hbm_bytes_per_launch=int(hbm_total / max(cnt, 1)),
                             hbm_GBps=round(hbm_total / (ms * 1e-3) / 1e9, 1) if ms else None)
    if any(t in k for t in ("radix_", "scan_", "rowptr_from_sorted", "coo_check")):
        conv_hbm += hbm_total
        conv_ms += ms
# the probe converts twice with these kernels (the matrix itself and the column relabelling's small class sort); the class sort is ~1 % of it
out["coo_to_csr_kernels_total"] = {"ms": round(conv_ms, 3), "hbm_bytes": int(conv_hbm), "hbm_over_algorithmic": round(conv_hbm / (32 * nnz + 12 * n), 2),
                                   "note": "radix_* + scan_* + rowptr_from_sorted + coo_check, both conversions of the run (matrix: 3 passes of 8 bits; "
                                           "column classes: 1 pass)"}
json.dump(out, open(os.path.join(ROOT, "profiles/r02_pmc_convert_rmat.json"), "w"), indent=1)
shutil.copy(os.path.join(ROOT, "gpurun_out/r02_convert_prof/convert_kernel_stats.csv"), os.path.join(ROOT, "profiles/r02_convert_kernel_stats.csv"))
print(json.dumps(out, indent=1))
