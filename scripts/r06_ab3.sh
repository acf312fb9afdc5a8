#!/usr/bin/env bash
# round 6: k_filter A/B with the stage times of the call (BSK_BENCH_PROFILE=1: HIP events around every stage)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_filter_gpu.py tests/test_grep_subseq_gpu.py tests/test_locate_gpu.py tests/test_golden_gpu.py tests/test_cli.py -q -m gpu -x -n 4 2>&1 | tail -3
BSK_FUZZ_SEEDS=12 python -m pytest tests/test_fuzz_gpu.py -q -m gpu -n 4 2>&1 | tail -2
export BSK_BENCH_PROFILE=1
{
  for v in "" "-DBSK_FILTER_WAVES=6" "-DBSK_FILTER_WAVES=7" "-DBSK_FILTER_WAVES=6" ""; do
    bash scripts/variant_src.sh stream_filter.hip "$v" grep,locate
  done
} > $O/r06_ab3_filter.txt 2>&1
grep -E "==|grep|locate" $O/r06_ab3_filter.txt | sed 's/"k_index_compact.*//'
