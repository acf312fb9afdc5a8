#!/usr/bin/env bash
# RESULT: slower -- FASTA-5k 50 GB 11.1 -> 11.5 ms, FASTA-1k 20 GB 4.90 -> 4.94 (three alternating runs); the two uniform branches
# per piece cost more than the 64-bit compares they skip.  The source change is not in the tree.
# round 5: stats -a on FASTA with the skip-region compares behind a wave-uniform flag (tree) against before (gpurun_alt/libbsk_base.so)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_stats_fasta2_gpu.py tests/test_stats_gpu.py -q -x -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
  echo -n "base: "; BSK_LIB=$PWD/gpurun_alt/libbsk_base.so python scripts/bench_stats_fasta.py 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print([ (k[:14], v['ms']) for k,v in d.items() if '-a' in k])"
  echo -n "tree: "; python scripts/bench_stats_fasta.py 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print([ (k[:14], v['ms']) for k,v in d.items() if '-a' in k])"
done
