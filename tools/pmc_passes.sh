#!/bin/bash
# usage: tools/pmc_passes.sh <outdir> <probe args...>   -- one rocprofv3 --pmc pass per counter group
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAVES_EQ_64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- python3 ${PROBE:-tools/spmv_probe.py} "$@" > "$out.p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out.p$i.log"; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "spmv" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:40s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
