#!/bin/bash
# Round-3 evidence: (1) rocprofv3 kernel stats of the default bench.py run, (2) HBM traffic by PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes) of the CG loops bench.py times, (3) the same for the
# R-MAT conversion + product and for the FE-like matrix the row-group kernel serves, (4) kernel trace of the row-block probe.
# usage (on the GPU box): bash tools/r03_profiles.sh   -> gpurun_out/r03_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/r03_bench_prof -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/r03_bench_under_rocprof.json 2> $O/r03_bench_under_rocprof.err || exit 1
for w in "512x512x512 0 0" "512x512x512 1 0"; do
  tag=$(echo $w | tr ' ' '_')
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    c1=$(echo $ctr | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $ctr -d $O/r03_pmc_${tag}_$c1 -o pmc --output-format csv -- python3 $R/tools/cgk_probe.py $w > /dev/null 2>&1 || exit 1
  done
done
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  c1=$(echo $ctr | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $ctr -d $O/r03_pmc_convert_$c1 -o pmc --output-format csv -- python3 $R/tools/rmat_probe.py > /dev/null 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc $ctr -d $O/r03_pmc_group_$c1 -o pmc --output-format csv -- python3 $R/tools/group_probe.py 500000 26 300 > /dev/null 2>&1 || exit 1
done
export MK_PROBE_SINGLE=1   # the probe also times the single-reduction step
timeout -k 10 200 python3 $R/tools/dist_probe.py 512 512 512 8 100 > $O/r03_dist_probe_plain.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/r03_dist_prof -o dist --output-format csv -- python3 $R/tools/dist_probe.py 512 512 512 8 40 > $O/r03_dist_probe.txt 2>&1 || exit 1
unset MK_PROBE_SINGLE
# (5) FE-like matrices, (6) 27-point variable coefficients on the long-row vsweep kernel: time and HBM traffic
timeout -k 10 700 python3 $R/tools/fem_probe.py hex3 64 hex3 100 nine 3000 rand3 700000 15 300 rand3 500000 26 300 rand3 500000 26 30 rand 2000000 45 3000 rand 3000000 24 1500 rand 2000000 45 7000 rand 1500000 80 20000 randfat 2000000 45 3000 2000 1000 rand 6000000 7 1000 > $O/r03_fem_like_matrices.txt 2>&1 || exit 1
export MK_PROBE_KIND=27
timeout -k 10 200 python3 $R/tools/vsweep_probe.py 256x256x256 512x512x64 > $O/r03_vsweep27_probe.txt 2>&1 || exit 1
timeout -k 10 200 python3 $R/tools/vcg_probe.py 0 256,256,256 512,512,64 >> $O/r03_vsweep27_probe.txt 2>&1 || exit 1
for ctr in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -s KILL 200 rocprofv3 --pmc $ctr -d $O/r03_pmc_vsweep27_$ctr -o pmc --output-format csv -- python3 $R/tools/vsweep_probe.py 256x256x256 > /dev/null 2>&1 || exit 1
done
echo done
