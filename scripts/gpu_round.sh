#!/usr/bin/env bash
# One GPU visit: gpu tests, the bench line, rocprofv3 kernel-trace stats and PMC passes.
# Usage (from the repo root on the GPU box):  bash scripts/gpu_round.sh [tag]
TAG=${1:-r1}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out; mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5) > $O/tests_$TAG.log 2>&1
(timeout 600 python bench.py 2>&1 | tail -1) > $O/bench_$TAG.json 2>$O/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline) > $O/prof_$TAG.log 2>&1
# PMC passes: their own runs, counters only (FETCH_SIZE and WRITE_SIZE do not fit one pass)
(timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$TAG -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline) > $O/pmc_fetch_$TAG.log 2>&1
(timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline) > $O/pmc_write_$TAG.log 2>&1
cd $R
cat $O/tests_$TAG.log $O/bench_$TAG.json
find $O/prof_$TAG $O/pmc_fetch_$TAG $O/pmc_write_$TAG -type f | head -20
for f in $(find $O/prof_$TAG -name '*kernel_stats.csv'); do head -8 $f; done
