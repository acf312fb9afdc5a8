#!/usr/bin/env bash
# Evidence for the hot-path operators at the BASELINE sizes (one GPU visit):
#   bash scripts/ops_evidence.sh r02 [ops]
# 1. scripts/bench_ops.py           -> times, algorithmic GB/s, fraction of the 8 TB/s roof
# 2. rocprofv3 --kernel-trace --stats of the same command -> profiles/<tag>_ops_kernel_stats.csv
# 3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes)  -> HBM bytes per command (gfx950 read correction)
# merged into profiles/<tag>_ops.json: one "roofline" object per command (achieved, frac, traffic, traffic / algorithmic).
TAG=${1:-r02}; OPS=${2:-seq,subseq,grep,locate,rmdup,translate}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O $R/profiles
cd $R
python scripts/bench_ops.py 1 3 $OPS > $O/ops_$TAG.json 2> $O/ops_$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ops_$TAG -o ops -- python $R/scripts/bench_ops.py 1 1 $OPS > $O/prof_ops_$TAG.log 2>&1
cd $R
cp $O/prof_ops_$TAG/ops_kernel_stats.csv profiles/${TAG}_ops_kernel_stats.csv 2>/dev/null
for op in $(echo $OPS | tr ',' ' '); do
  bash scripts/pmc_ops_traffic.sh $op 1.0 $O/traffic_${TAG}_$op.json > $O/traffic_${TAG}_$op.txt 2>&1
done
python - "$O" "$TAG" "$OPS" <<'PY'
import json, sys, os
O, tag, ops = sys.argv[1:4]
res = json.load(open(f"{O}/ops_{tag}.json"))
key_of = {"seq": "seq -n", "subseq": "subseq", "grep": "grep -s", "locate": "locate", "rmdup": "rmdup", "translate": "translate"}
out = {"source": "scripts/ops_evidence.sh: bench_ops.py 1 3 (times: mean of 3 calls after a warm-up, whole operator call, data resident in HBM); "
                 "traffic: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench_ops.py 1 1 (two calls of the command), "
                 "summed over the command's kernels and halved; fetch x2 = gfx950 correction (profiles/r01_calibration_stream_read.json)",
       "ops": {}}
for op in ops.split(","):
    tf = f"{O}/traffic_{tag}_{op}.json"
    traffic = None
    if os.path.exists(tf):
        t = json.load(open(tf))["kernels"]
        traffic = sum(v["total_GB_all_dispatches"] for v in t.values()) / 2.0  # two calls of the command
    for name, v in res.items():
        if name.startswith(key_of.get(op, op)):
            alg = (v["in_GB"] + v["out_GB"])
            v["roofline"] = {"bound": "hbm", "achieved": v["algorithmic_GBps"], "peak": 8000.0, "unit": "GB/s", "frac": v["frac_of_8TBps"],
                             "algorithmic_GB": round(alg, 3), "traffic_GB": None if traffic is None else round(traffic, 2),
                             "traffic_over_algorithmic": None if traffic is None else round(traffic / alg, 3)}
            out["ops"][name] = v
json.dump(out, open(f"profiles/{tag}_ops.json", "w"), indent=1)
for k, v in out["ops"].items():
    r = v["roofline"]
    print("%-52s %8.2f ms  frac %.3f  traffic %s GB (x%s)" % (k[:52], v["ms"], r["frac"], r["traffic_GB"], r["traffic_over_algorithmic"]))
PY
