#!/usr/bin/env bash
# round 6: k_filter with the verification from LDS -- tests, then the A/B in one visit (as built / verification from memory /
# 5 waves per SIMD)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_filter_gpu.py tests/test_grep_subseq_gpu.py tests/test_locate_gpu.py tests/test_golden_gpu.py -q -m gpu -x -n 4 2>&1 | tail -6 > $O/r06_ab2_tests.log
BSK_FUZZ_SEEDS=12 python -m pytest tests/test_fuzz_gpu.py -q -m gpu -n 4 2>&1 | tail -3 >> $O/r06_ab2_tests.log
{
  bash scripts/variant_src.sh stream_filter.hip "" grep,locate
  bash scripts/variant_src.sh stream_filter.hip "-DBSK_FILTER_VERIFY_LDS=0" grep,locate
  bash scripts/variant_src.sh stream_filter.hip "-DBSK_FILTER_WAVES=5" grep,locate
  bash scripts/variant_src.sh stream_filter.hip "-DBSK_FILTER_DIAG=1" grep
  bash scripts/variant_src.sh stream_filter.hip "" grep,locate
} > $O/r06_ab2_filter.txt 2>&1
cat $O/r06_ab2_tests.log; grep -E "==|grep|locate" $O/r06_ab2_filter.txt
