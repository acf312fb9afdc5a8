#!/usr/bin/env bash
# round 5, visit 1: the chain-free grouping key of the byte-verifying rmdup (hash_dev.hpp) against XXH64 in the same pass
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_rmdup_keys_gpu.py tests/test_translate_rmdup_gpu.py tests/test_segcopy_gpu.py tests/test_multirank_gpu.py -q -x 2>&1 | tail -8) > $O/r05a_tests.log 2>&1
cat $O/r05a_tests.log
export BSK_BENCH_PROFILE=1
echo "== xxh64 (round 4)"; BSK_RMDUP_HASH=xxh64 python scripts/bench_ops.py 1.0 3 rmdup 2>&1 | tail -1 | cut -c1-1500
echo "== grouping key"; python scripts/bench_ops.py 1.0 3 rmdup 2>&1 | tail -1 | cut -c1-1500
for f in "-DBSK_RMSTREAM_WAVES_G=4"; do bash scripts/variant_src.sh stream_rmdup.hip "$f" rmdup; done
