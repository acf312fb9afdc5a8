#!/usr/bin/env bash
# round 4, visit c: translate on uniform records without a table (UniformLayout)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests/test_translate_uniform_gpu.py tests/test_translate_light_gpu.py tests/test_translate_rmdup_gpu.py tests/test_golden_gpu.py -q -x -n 4 2>&1 | tail -15) > $O/r04c_tests.log 2>&1
cat $O/r04c_tests.log
bash scripts/timeline_ops.sh translate 1.0 r04tlc > $O/r04tlc_translate.txt 2>&1; tail -25 $O/r04tlc_translate.txt
