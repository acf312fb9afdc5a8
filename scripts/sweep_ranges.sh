#!/usr/bin/env bash
# k_stats launch time vs shard size and ranges per wave (tail / start-up cost of the persistent grid)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
for GB in 3.125 6.25 12.5 25 100; do
  for RPW in 2 4 8 16; do
    BSK_RANGES_PER_WAVE=$RPW timeout 300 python bench.py --gb $GB --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('GB=$GB rpw=$RPW step_ms=%.3f k_stats_ms=%.3f prep_ms=%.3f frac=%.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['k_prep_avg_launch_ms'], d['roofline']['frac']))"
  done
done
