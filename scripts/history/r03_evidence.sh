#!/usr/bin/env bash
# Evidence of a round-3 state (one GPU visit):  bash scripts/r03_evidence.sh <tag> [ops list | none]
#   1. the driver-style bench line with cpu baseline and the "ops" object      -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command                     -> gpurun_out/prof_<tag>/
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes) of the stats legs   -> gpurun_out/pmc_{fetch,write}_<tag>/
#   4. scripts/ops_evidence.sh <tag>: per-command timing, kernel stats, PMC     -> gpurun_out/ (merge on the build box)
TAG=${1:-r03}; OPS=${2:-seq,subseq,grep,locate,rmdup,translate}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1700 python bench.py 2>$O/${TAG}_bench.err | tail -1) > $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline) > $O/prof_$TAG.log 2>&1
(timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$TAG -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ops) > $O/pmc_fetch_$TAG.log 2>&1
(timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ops) > $O/pmc_write_$TAG.log 2>&1
cd $R
[ "$OPS" != "none" ] && bash scripts/ops_evidence.sh $TAG $OPS > $O/${TAG}_ops_evidence.log 2>&1
python - <<PY
import json
d=json.load(open("$O/${TAG}_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","bit_exact_vs_expected_row")}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["stats_all"]["ms_per_step"], d.get("cpu_baseline",{}).get("value"))
for k,e in d.get("ops",{}).items():
    print("%-26s %8.3f ms frac %.4f exact %s" % (k,e["ms"],e["frac"],e["exact"]) if "ms" in e else (k,e))
PY
for f in $(find $O/prof_$TAG -name '*kernel_stats.csv'); do head -12 $f | cut -c1-160; done
