#!/usr/bin/env bash
# round 5, visit 8: the full driver line (CPU baselines on worker processes, scaling_model, the new legs)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
( time (timeout 1500 python bench.py 2>$O/r05h_bench.err | tail -1) > $O/r05h_bench.json ) 2>&1 | tail -3
python - <<PY
import json
d=json.load(open("$O/r05h_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","bit_exact_vs_expected_row")}, d["roofline"]["frac"], d["cpu_baseline"]["value"], d.get("cpu_baseline_all_cores",{}).get("value"))
print(json.dumps(d.get("scaling_model"))[:1500])
for k,v in d["ops"].items():
    if isinstance(v,dict) and "ms" in v:
        cb=v.get("cpu_baseline") or {}
        print("%-46s %8.3f ms frac %.4f exact %s cpu1 %s all %s" % (k, v["ms"], v["frac"], v["exact"], cb.get("value"), (cb.get("all_cores") or {}).get("value", (cb.get("all_cores") or {}).get("error"))))
    else: print(k, v)
print(json.dumps(d.get("end_to_end"))[:600])
PY
tail -3 $O/r05h_bench.err
