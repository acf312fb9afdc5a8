#!/usr/bin/env bash
# round 5: the evidence of the HEADLINE line (bench.py, N = 1): rocprofv3 kernel-trace stats of the same command, and the
# FETCH_SIZE / WRITE_SIZE passes (their own runs, counters only) -> gpurun_out/; then on the build box
#   python scripts/pmc_traffic.py r05  &&  cp gpurun_out/prof_r05/*kernel_stats.csv profiles/r05_kernel_stats.csv
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ops --no-scaling-model) > $O/prof_$TAG.log 2>&1
(timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$TAG -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ops --no-scaling-model) > $O/pmc_fetch_$TAG.log 2>&1
(timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ops --no-scaling-model) > $O/pmc_write_$TAG.log 2>&1
cd $R
grep -h '"metric"' $O/prof_$TAG.log | tail -1 | cut -c1-400
for f in $(find $O/prof_$TAG -name '*kernel_stats.csv'); do head -6 $f; done
