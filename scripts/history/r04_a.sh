#!/usr/bin/env bash
# round 4, first GPU visit: the strict line-break check of k_translate_wide (new test + its cost at C4), the N-rank bench
# legs (grep @ C3, rmdup @ C5) with two ranks sharing the GPU
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_translate_light_gpu.py tests/test_bench_gpu.py tests/test_multirank_gpu.py -q -x 2>&1 | tail -8) > $O/r04a_tests.log 2>&1
python scripts/bench_ops.py 1 3 translate 2>&1 | tail -1 > $O/r04a_ops_translate.json
cat $O/r04a_tests.log; head -c 1500 $O/r04a_ops_translate.json; echo
