#!/usr/bin/env bash
# what smaller ranges cost the streaming passes (a staging-in-cache output path would want ~64 KiB ranges instead of 512 KiB)
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for r in 0 19 37 74 148; do
  echo "== ranges_per_wave=$r"
  BSK_RANGES_PER_WAVE=$r BSK_MIN_RANGE_BYTES=16384 python scripts/bench_ops.py 1 3 subseq,seq 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print('   %-40s %8.2f ms  %s' % (k[:40], v['ms'], v['note']))"
done
