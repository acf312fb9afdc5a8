#!/usr/bin/env bash
# quick GPU visit: gpu tests + bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -${2:-6}) > $O/quick_tests.log 2>&1
(timeout 600 python bench.py --no-cpu-baseline ${1:-} 2>&1 | tail -1) > $O/quick_bench.json 2>$O/quick_bench.err
cat $O/quick_tests.log; python - <<PY
import json
try:
    d=json.load(open("$O/quick_bench.json"))
    print({k:d[k] for k in ("value","gb_per_s","ms_per_step","bit_exact_vs_expected_row")}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["stats_all"])
except Exception as e:
    print("bench parse failed", e); print(open("$O/quick_bench.json").read()[-2000:]); print(open("$O/quick_bench.err").read()[-2000:])
PY
