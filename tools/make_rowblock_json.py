#!/usr/bin/env python3
"""usage: make_rowblock_json.py [tag] (default r03).  gpurun_out/<tag>_dist_prof (rocprofv3 --kernel-trace of tools/dist_probe.py 512 512 512 8) -> profiles/r02_rowblock_slab_kernels.json:
per kernel of the row-block CG iteration the launches and average duration, only launches on 512x512x64 slabs (the
single-matrix solve of the same probe is cut off by grid size)."""
import collections, csv, glob, json, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open((glob.glob(os.path.join(ROOT, f"gpurun_out/{TAG}_dist_prof/**/dist_kernel_trace.csv"), recursive=True) + glob.glob(os.path.join(ROOT, f"gpurun_out/{TAG}_dist_prof/dist_kernel_trace.csv")))[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the distributed part starts with the first split_count_kernel (mk_dist_matrix_create)
start = next(i for i, r in enumerate(rows) if "split_count_kernel" in r["Kernel_Name"])
agg = collections.defaultdict(list)
for r in rows[start:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
keep = ("spmv_sweep_kernel", "offd_", "pack", "finalize_slab", "copyBuffer", "local_allreduce")
out = {"what": "rocprofv3 --kernel-trace of tools/dist_probe.py 512 512 512 8: CG on 512^3 as 8 z-slabs (512x512x64 each) driven as a local "
               "world on one GPU (device copies stand in for RCCL, a side stream for the communication stream); kernels launched after the "
               "row blocks were built; probe result in `probe`",
       "probe_without_profiler": next((l.strip() for l in open(os.path.join(ROOT, f"gpurun_out/{TAG}_dist_probe_plain.txt")) if "ms/iteration" in l), None)
       if os.path.exists(os.path.join(ROOT, f"gpurun_out/{TAG}_dist_probe_plain.txt")) else None,
       "probe": next(l.strip() for l in open(os.path.join(ROOT, f"gpurun_out/{TAG}_dist_probe.txt")) if "ms/iteration" in l),
       "kernels": {k: {"launches": len(v), "avg_us": round(sum(v) / len(v), 2)} for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))
                   if any(t in k for t in keep)}}
json.dump(out, open(os.path.join(ROOT, f"profiles/{TAG}_rowblock_slab_kernels.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
