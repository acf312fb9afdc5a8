#!/usr/bin/env bash
# per-kernel times of scripts/bench_ops.py for a subset of commands:  bash scripts/prof_ops.sh translate 0.25
OPS=${1:-translate}; SCALE=${2:-0.25}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$OPS -o ops -- python $R/scripts/bench_ops.py $SCALE 1 $OPS > $O/prof_$OPS.out 2>&1
cd $R
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof_$OPS/ops_kernel_stats.csv")))
for r in rows[:14]:
    print("%-64s calls=%s avg_ms=%.3f total_ms=%.1f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6))
PY
tail -2 $O/prof_$OPS.out
