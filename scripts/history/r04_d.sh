#!/usr/bin/env bash
# round 4, visit d: rmdup with the byte-verifying default + leaner preparation; bench legs (cpu baselines, end to end)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -q -x -n 4 2>&1 | tail -12) > $O/r04d_tests.log 2>&1
cat $O/r04d_tests.log
bash scripts/timeline_ops.sh rmdup 1.0 r04tld > $O/r04tld_rmdup.txt 2>&1; tail -42 $O/r04tld_rmdup.txt
(timeout 900 python bench.py 2>$O/r04d_bench.err | tail -1) > $O/r04d_bench.json; tail -3 $O/r04d_bench.err
python - <<PY
import json
d=json.load(open("$O/r04d_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","bit_exact_vs_expected_row")}, d["roofline"]["frac"], d["stats_all"]["ms_per_step"], d["stats_all"]["verified"])
for k,e in d["ops"].items():
    if isinstance(e,dict) and "ms" in e:
        print(k, e["ms"], e["frac"], e["exact"], "host", e.get("host_ms_per_call"), e["kernels_ms_per_call"])
        print("    cpu:", e.get("cpu_baseline",{}).get("value"), e.get("cpu_baseline",{}).get("all_cores",{}).get("value"), e.get("rmdup_keys_two_key",{}).get("ms"))
print(json.dumps(d.get("end_to_end"), indent=0)[:1500])
PY
