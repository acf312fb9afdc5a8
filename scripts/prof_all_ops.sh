#!/usr/bin/env bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of every command of scripts/bench_ops.py, one profile per command
SCALE=${1:-1.0}; TAG=${2:-r01d}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for OP in ${3:-seq subseq grep locate rmdup translate}; do
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$OP -o ops -- python $R/scripts/bench_ops.py $SCALE 2 $OP > $O/prof_${TAG}_$OP.out 2>&1
  echo "== $OP"; tail -1 $O/prof_${TAG}_$OP.out | head -c 1500; echo
  python - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof_${TAG}_$OP/ops_kernel_stats.csv")))
for r in rows[:9]:
    print("  %-70s calls=%s avg_ms=%.3f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e6))
PY
done
