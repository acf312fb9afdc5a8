#!/usr/bin/env bash
# round 6, one visit: (1) the tests the last full run failed + the translate tests on the predict-and-verify search,
# (2) translate on records that differ with / without the prediction, at three range sizes, (3) k_filter: the pass as it is,
# without its verification, without its search (measurement switches BSK_FILTER_DIAG), (4) the FETCH_SIZE discrimination
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
python -m pytest tests/test_out_slices_gpu.py tests/test_rmdup_xcheck_gpu.py tests/test_translate_stream_gpu.py tests/test_translate_uniform_gpu.py tests/test_translate_light_gpu.py tests/test_translate_wide_gpu.py tests/test_host_c_gpu.py -q -m gpu -x 2>&1 | tail -15 > $O/r06_ab1_tests.log
BSK_FUZZ_SEEDS=12 python -m pytest tests/test_fuzz_gpu.py -q -m gpu -n 4 2>&1 | tail -5 >> $O/r06_ab1_tests.log
{
  echo "== translate, records that differ (50 GB)"
  python scripts/bench_translate_var.py 50 3
  BSK_TRANSLATE_PROBE=off python scripts/bench_translate_var.py 50 3
  BSK_MIN_RANGE_BYTES=524288 python scripts/bench_translate_var.py 50 3
  BSK_MIN_RANGE_BYTES=2097152 python scripts/bench_translate_var.py 50 3
  BSK_MIN_RANGE_BYTES=262144 python scripts/bench_translate_var.py 50 3
} > $O/r06_ab1_translate.txt 2>&1
{
  bash scripts/variant_src.sh stream_filter.hip "" grep
  bash scripts/variant_src.sh stream_filter.hip "-DBSK_FILTER_DIAG=1" grep
  bash scripts/variant_src.sh stream_filter.hip "-DBSK_FILTER_DIAG=2" grep
  bash scripts/variant_src.sh stream_filter.hip "" grep
} > $O/r06_ab1_filter.txt 2>&1
bash scripts/fetch_calibration.sh r06 > $O/r06_ab1_calib.txt 2>&1
tail -20 $O/r06_ab1_tests.log; cat $O/r06_ab1_translate.txt; grep -E "==|grep" $O/r06_ab1_filter.txt; tail -25 $O/r06_ab1_calib.txt
