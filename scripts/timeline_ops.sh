#!/usr/bin/env bash
# kernel timeline of ONE operator call (the last one of scripts/bench_ops.py): every launch with its start offset, its
# duration and the idle gap in front of it -- where the host's read-backs and launch latencies sit
#   bash scripts/timeline_ops.sh grep 1.0 [tag]
OPS=${1:-grep}; SCALE=${2:-1.0}; TAG=${3:-tl}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_$OPS
BSK_TIMELINE=1 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_$OPS -o ops -- python $R/scripts/bench_ops.py $SCALE 2 $OPS > $O/${TAG}_$OPS.out 2>&1
cd $R
python - <<PY
import csv, glob, re
f = glob.glob("$O/${TAG}_$OPS/**/ops_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# the calls are 20 ms apart (BSK_TIMELINE=1): the last call = everything behind the last idle stretch of > 10 ms
first = 0
for i in range(len(ks) - 1, 0, -1):
    if ks[i][1] - ks[i - 1][2] > 10_000_000:
        first = i
        break
def short(name):
    m = re.search(r"\\b(k_\\w+|__amd_rocclr_\\w+|trampoline_kernel|\\w+_kernel\\w*)", name)
    base = m.group(1) if m else name[:60]
    t = re.search(r"k_\\w+<([^>]{0,40})", name)
    return base + ("<" + t.group(1) + ">" if t else "")
t0 = ks[first][1]
prev_end = t0
busy = 0
lines = []
for name, s, e in ks[first:]:
    lines.append("%9.3f  +%7.3f gap  %8.3f ms  %s" % ((s - t0) / 1e6, (s - prev_end) / 1e6, (e - s) / 1e6, short(name)))
    busy += e - s
    prev_end = max(prev_end, e)
span = (prev_end - t0) / 1e6
print("\\n".join(lines))
print("call span %.3f ms, kernels %.3f ms, idle inside the call %.3f ms, %d launches" % (span, busy / 1e6, span - busy / 1e6, len(ks) - first))
PY
tail -1 $O/${TAG}_$OPS.out | head -c 600; echo
