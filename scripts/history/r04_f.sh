#!/usr/bin/env bash
# round 4, visit f: whole-record sinks (stats, names): suite + bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -q -x -n 4 2>&1 | tail -12) > $O/r04f_tests.log 2>&1
cat $O/r04f_tests.log
(timeout 900 python bench.py --no-cpu-baseline 2>$O/r04f_bench.err | tail -1) > $O/r04f_bench.json; tail -3 $O/r04f_bench.err
python - <<PY
import json
d=json.load(open("$O/r04f_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","bit_exact_vs_expected_row")}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["stats_all"]["ms_per_step"], d["stats_all"]["k_stats_avg_launch_ms"], d["stats_all"]["verified"])
for k,e in d["ops"].items():
    if isinstance(e,dict) and "ms" in e:
        print(k, e["ms"], e["frac"], e["exact"], "host", e.get("host_ms_per_call"), e["kernels_ms_per_call"], e.get("rmdup_keys_two_key",{}).get("ms"))
PY
