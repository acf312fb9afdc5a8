#!/usr/bin/env bash
# A/B of k_stats build variants on the GPU box (one process per variant, same data)
cd "$GRAFT_REPO_ROOT"
for v in ${VARIANTS:-"" "-DBSK_STATS_WAVES=8" "-DBSK_STATS_WAVES=6" "-DBSK_NPIECE=3" "-DBSK_NPIECE=6"}; do
  bash scripts/variant.sh "$v" > /dev/null 2>&1
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-40s stats %.2f ms (frac %.3f)   stats -a %.2f ms' % ('$v' or 'default', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['stats_all']['k_stats_avg_launch_ms']))"
done
bash scripts/variant.sh "" > /dev/null 2>&1
