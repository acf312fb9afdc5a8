#!/usr/bin/env bash
# one rank's step at the shard sizes of 2 / 4 / 8 GPUs (50 / 25 / 12.5 GB): ranges per wave
cd $GRAFT_REPO_ROOT
for gb in 12.5 25 50; do
  for r in 0 6 8 12 16 24; do
    BSK_RANGES_PER_WAVE=$r python bench.py --gb $gb --steps 20 --warmup 3 --no-cpu-baseline --no-ops 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('gb $gb ranges_per_wave $r: step %.3f ms kernel %.3f prep %.3f frac %.3f exact %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['k_prep_avg_launch_ms'], d['roofline']['frac'], d['bit_exact_vs_expected_row']))"
  done
done
