#!/bin/bash
# kernel trace of the row-block probe with the single-reduction step beside the default one (gpurun_out/r03s_dist_prof)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export MK_PROBE_SINGLE=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r03s_dist_prof -o dist --output-format csv -- python3 $R/tools/dist_probe.py 512 512 512 8 40 > $O/r03s_dist_probe.txt 2>&1 || exit 1
echo done
