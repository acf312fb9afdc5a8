#!/usr/bin/env bash
# round 5: the whole GPU suite, then the driver line (what the driver runs at the end of the round)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
( time (timeout 2400 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | grep -v "^  File" | tail -15) ) > $O/r05_full_tests.log 2>&1
cat $O/r05_full_tests.log
T0=$(date +%s); timeout 1500 python bench.py 2>$O/r05_full_bench.err | tail -1 > $O/r05_full_bench.json; echo "bench.py wall: $(( $(date +%s) - T0 )) s" | tee $O/r05_full_bench.wall
python - <<PY
import json
d=json.load(open("$O/r05_full_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","bit_exact_vs_expected_row")}, d["roofline"]["frac"], d["cpu_baseline"]["value"], d.get("cpu_baseline_all_cores",{}).get("value"))
for k,v in d["ops"].items():
    if isinstance(v,dict) and "ms" in v:
        cb=v.get("cpu_baseline") or {}
        print("%-46s %8.3f ms frac %.4f exact %s traffic %s cpu1 %s all %s" % (k, v["ms"], v["frac"], v["exact"], v.get("traffic_over_algorithmic"), cb.get("value"), (cb.get("all_cores") or {}).get("value", (cb.get("all_cores") or {}).get("error"))))
    else: print(k, v)
PY
tail -3 $O/r05_full_bench.err
