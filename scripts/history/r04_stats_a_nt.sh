#!/usr/bin/env bash
# (the knob BSK_STATS_A_NT was taken out again: 20.45 against 20.76 - 20.90 ms at 5 waves per SIMD with one spilled register; 4 waves 21.8)
# `stats -a` with non-temporal tile loads at 4 / 5 waves per SIMD (5: one spilled register) against the ordinary loads
cd $GRAFT_REPO_ROOT
for f in "" "-DBSK_STATS_A_NT=1" "-DBSK_STATS_A_NT=1 -DBSK_STATS_WAVES_ALL=4" "-DBSK_STATS_WAVES_ALL=4" ""; do
  bash scripts/variant.sh "$f"
  echo "== stream_stats.hip $f"
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-ops 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('   stats kernel %.3f | -a step %.3f kernel %.3f  exact %s %s' % (d['roofline']['avg_launch_ms'], d['stats_all']['ms_per_step'], d['stats_all']['k_stats_avg_launch_ms'], d['bit_exact_vs_expected_row'], d['stats_all']['verified']))"
done
