#!/usr/bin/env bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
(timeout 1200 python -m pytest tests/test_translate_light_gpu.py tests/test_translate_wide_gpu.py tests/test_translate_rmdup_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py tests/test_cli.py tests/test_store_gpu.py -q -x 2>&1 | tail -12)
export BSK_BENCH_PROFILE=1
python scripts/bench_ops.py 1 3 translate 2>&1 | tail -1 | cut -c1-600
