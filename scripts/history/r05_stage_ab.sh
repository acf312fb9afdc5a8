#!/usr/bin/env bash
# RESULT: no difference in time (k_rmdup_stream 6.24-6.42 both, k_subseq_stream 8.31-8.40 both).  Kept for what it found: WITHOUT a
# compiler barrier behind the carry read the compiler sank that load under the tile stores (per thread they do not alias; across
# the lanes of the wave they do) and lines that began in the tile before hashed wrong -- the old order was safe only by luck.
# round 5: TileLds::stage with the carry read first / written last (tree) against read -> write -> tile (gpurun_alt/libbsk_base.so)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rmdup_keys_gpu.py tests/test_subseq_stream_gpu.py tests/test_translate_rmdup_gpu.py -q -x -m gpu 2>&1 | tail -2
for rep in 1 2 3; do for v in base tree; do
  if [ $v = base ]; then export BSK_LIB=$PWD/gpurun_alt/libbsk_base.so; else unset BSK_LIB; fi
  echo -n "$v: "; BSK_BENCH_PROFILE=1 python scripts/bench_ops.py 1 5 rmdup,subseq 2>&1 | tail -1 | grep -o 'k_rmdup_stream[^,]*\|k_subseq_stream[^,}]*' | tr '\n' ' '; echo
done; done
