#!/bin/bash
# usage: tools/pmc_groups.sh <outdir> <probe.py> <args...> -- one rocprofv3 --pmc pass per counter group, then the per-kernel means
out=$1; probe=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- python3 "$probe" "$@" > "$out/p$i.log" 2>&1 || { echo "pass $i ($grp) failed"; tail -5 "$out/p$i.log"; exit 1; }
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
FETCH_SIZE TCC_HIT_sum
TCC_MISS_sum TCC_REQ_sum
LIST
python3 - "$out" "${KERNEL_FILTER:-spmv}" <<'PY'
import csv, glob, sys, collections
out, filt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if filt not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:40s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
