#!/usr/bin/env bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
(timeout 900 python -m pytest tests/test_names_gpu.py tests/test_seq_gpu.py -q -x 2>&1 | tail -5)
export BSK_BENCH_PROFILE=1
echo "== tile in LDS"; python scripts/bench_ops.py 1 3 seq 2>&1 | tail -1 | cut -c1-400
echo "== BSK_NAMES_LDS=off"; BSK_NAMES_LDS=off python scripts/bench_ops.py 1 3 seq 2>&1 | tail -1 | cut -c1-400
