#!/usr/bin/env bash
# round 5, visit 2: k_rmdup_place (one pass for sizes / comparison / offsets / segment list) against round 4's five passes
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_rmdup_keys_gpu.py tests/test_translate_rmdup_gpu.py tests/test_segcopy_gpu.py tests/test_multirank_gpu.py tests/test_fuzz_gpu.py -q -x 2>&1 | tail -8) > $O/r05b_tests.log 2>&1
cat $O/r05b_tests.log
export BSK_BENCH_PROFILE=1
echo "== five passes"; BSK_RMDUP_PLACE=off python scripts/bench_ops.py 1.0 3 rmdup 2>&1 | tail -1 | cut -c1-1500
echo "== one pass"; python scripts/bench_ops.py 1.0 3 rmdup 2>&1 | tail -1 | cut -c1-1500
