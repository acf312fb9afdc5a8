#!/usr/bin/env bash
# the driver-style bench line (N = 1, with the "ops" object) -> gpurun_out/<tag>_bench.json ; usage: bash scripts/r03_bench.sh r03b [bench args]
TAG=${1:-r03}; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1700 python bench.py "$@" 2>$O/${TAG}_bench.err | tail -1) > $O/${TAG}_bench.json
python - <<PY
import json
d=json.load(open("$O/${TAG}_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","bit_exact_vs_expected_row")}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["stats_all"]["ms_per_step"], d["stats_all"]["verified"])
ops=d.get("ops",{})
for k,e in ops.items():
    if isinstance(e,dict) and "ms" in e: print("%-26s %8.3f ms (min %.3f) frac %.4f exact %s  %s" % (k,e["ms"],e["ms_min"],e["frac"],e["exact"],e["kernels_ms_per_call"]))
    else: print(k,e)
PY
tail -3 $O/${TAG}_bench.err
