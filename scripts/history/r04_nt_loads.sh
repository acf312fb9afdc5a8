#!/usr/bin/env bash
# non-temporal tile loads in the streaming skeleton (the shard is read once): k_stats / k_stats -a at C2
cd $GRAFT_REPO_ROOT
for f in "" "-DBSK_LOAD_NT=1"; do
  bash scripts/variant.sh "$f"
  echo "== stream_stats.hip $f"
  python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-ops 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('   stats step %.3f ms kernel %.3f frac %.4f | -a step %.3f kernel %.3f  exact %s %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['stats_all']['ms_per_step'], d['stats_all']['k_stats_avg_launch_ms'], d['bit_exact_vs_expected_row'], d['stats_all']['verified']))"
done
