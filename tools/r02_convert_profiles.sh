#!/bin/bash
# Round-2 evidence for BASELINE configs[2] (R-MAT scale 24, 2^28 entries): rocprofv3 kernel stats of the COO->CSR
# conversion and the split-kernel SpMV, and their HBM traffic by separate PMC passes.
# usage (on the GPU box): bash tools/r02_convert_profiles.sh   -> gpurun_out/r02_convert_*
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/r02_convert_prof -o convert --output-format csv -- python3 $R/tools/rmat_probe.py > $O/r02_convert_probe.txt 2>&1
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  c1=$(echo $ctr | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $ctr -d $O/r02_pmc_convert_$c1 -o pmc --output-format csv -- python3 $R/tools/rmat_probe.py > /dev/null 2>&1
done
echo done
