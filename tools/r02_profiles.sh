#!/bin/bash
# Round-2 evidence: (1) rocprofv3 kernel stats of the default bench.py run, (2) HBM traffic by PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes) of the CG loops bench.py times.
# usage (on the GPU box): bash tools/r02_profiles.sh   -> gpurun_out/r02_*
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/r02_bench_prof -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/r02_bench_under_rocprof.json 2> $O/r02_bench_under_rocprof.err
for w in "512x512x512 0 0" "512x512x512 1 0"; do
  tag=$(echo $w | tr ' ' '_')
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    c1=$(echo $ctr | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $ctr -d $O/r02_pmc_${tag}_$c1 -o pmc --output-format csv -- python3 $R/tools/cgk_probe.py $w > /dev/null 2>&1
  done
done
echo done
