cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export MK_PROBE_SINGLE=1
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $ctr -d $O/sr_pmc_$ctr -o pmc --output-format csv -- python3 $R/tools/dist_probe.py 512 512 512 8 20 > /dev/null 2>&1 || exit 1
done
echo done
