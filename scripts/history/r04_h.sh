#!/usr/bin/env bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -q -x -n 4 2>&1 | tail -4) > $O/r04h_tests.log 2>&1; cat $O/r04h_tests.log
(timeout 900 python bench.py --no-cpu-baseline 2>$O/r04h_bench.err | tail -1) > $O/r04h_bench.json; tail -2 $O/r04h_bench.err
python - <<PY
import json
d=json.load(open("$O/r04h_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","bit_exact_vs_expected_row")}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["stats_all"]["ms_per_step"], d["stats_all"]["k_stats_avg_launch_ms"], d["stats_all"]["verified"])
for k,e in d["ops"].items():
    if isinstance(e,dict) and "ms" in e:
        print(k, e["ms"], e["frac"], e["exact"], e["kernels_ms_per_call"], e.get("rmdup_keys_two_key",{}).get("ms"))
PY
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_r04h -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ops) > $O/pmc_fetch_r04h.log 2>&1
(timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_r04h -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ops) > $O/pmc_write_r04h.log 2>&1
cd $R; python scripts/pmc_traffic.py r04h
