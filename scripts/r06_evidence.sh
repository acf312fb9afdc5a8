#!/usr/bin/env bash
# round 6 evidence (GPU box, one visit):
#   1. the headline line: rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes of bench.py (scripts/r05_headline_evidence.sh)
#   2. the FETCH_SIZE calibration with the discrimination probes
#   3. the ops: bench_ops.py timings, kernel-trace stats, FETCH / WRITE passes per command -- with the results as ordered
#      slices (BSK_OUT=slices: the contract bench.py's entries for seq -n / subseq / rmdup are quoted on)
#   4. the kernels the ops evidence does not reach (FASTA stats, one-pass translate of records that differ)
#   5. SQ budgets of the streaming passes
# afterwards, on the build box:  python scripts/pmc_traffic.py r06 ; BSK_OUT=slices python scripts/ops_traffic_merge.py r06 ; copy the csv / json files to profiles/
TAG=${1:-r06}; OPS=${2:-seq,subseq,grep,rmdup,translate}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
bash scripts/r05_headline_evidence.sh $TAG > $O/${TAG}_headline.txt 2>&1
bash scripts/fetch_calibration.sh $TAG > $O/${TAG}_calib.txt 2>&1
cp $O/${TAG}_fetch_calibration.json profiles/${TAG}_fetch_calibration.json
export BSK_OUT=slices
python scripts/bench_ops.py 1 3 $OPS > $O/ops_$TAG.json 2> $O/ops_$TAG.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ops_$TAG -o ops -- python $R/scripts/bench_ops.py 1 1 $OPS > $O/prof_ops_$TAG.log 2>&1 )
for op in $(echo $OPS | tr ',' ' '); do
  bash scripts/pmc_ops_traffic.sh $op 1.0 $O/traffic_${TAG}_$op.json > $O/traffic_${TAG}_$op.txt 2>&1
done
python scripts/ops_traffic_merge.py $TAG $OPS > $O/${TAG}_ops_traffic.txt 2>&1
cp profiles/${TAG}_ops_traffic.json $O/
unset BSK_OUT
# 4. extra traffic
( cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmcx6_fa_$c -o pmc -- python $R/scripts/bench_stats_fasta.py > $O/pmcx6_fa_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/pmcx6_tr_$c -o pmc -- python $R/scripts/bench_translate_var.py 50 1 > $O/pmcx6_tr_$c.log 2>&1
done )
python - "$O" "$TAG" <<'PY'
import csv, collections, glob, hashlib, json, os, sys
O, tag = sys.argv[1:3]
ROOT = os.getcwd()
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pmcx6_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("bsk::", "")
        if name.startswith("k_stats<false") or name.startswith("k_translate_stream") or name.startswith("k_stats_stitch"):
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {"source": "scripts/r06_evidence.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); KiB counters; reads x 2 (a request moves "
                 "the 128-byte line: profiles/r06_fetch_calibration.json); the largest dispatches of each kernel (the 50 GB inputs)", "kernels": {}}
ALG = {"k_stats<false": 49999997088, "k_translate_stream": 49999997088 + 100800000000}
for (k, cn), v in sorted(agg.items()):
    top = max(v); big = [x for x in v if x >= 0.7 * top]
    out["kernels"].setdefault(k, {})[cn + "_KiB"] = sum(big) / len(big)
for k, e in out["kernels"].items():
    rd = e.get("FETCH_SIZE_KiB", 0) * 1024 * 2; wr = e.get("WRITE_SIZE_KiB", 0) * 1024
    e["read_GB"] = round(rd / 1e9, 2); e["write_GB"] = round(wr / 1e9, 2)
    key = next((a for a in ALG if k.startswith(a)), None)
    if key:
        e["algorithmic_GB"] = round(ALG[key] / 1e9, 2); e["traffic_over_algorithmic"] = round((rd + wr) / ALG[key], 3)
        if key == "k_translate_stream":
            e["read_over_input"] = round(rd / 49999997088, 3)
    print(k, e)
h = hashlib.sha256()
for f in sorted(os.listdir(f"{ROOT}/bigseqkit_amd/csrc")):
    if f.endswith((".hip", ".hpp", ".inc")):
        h.update(open(f"{ROOT}/bigseqkit_amd/csrc/{f}", "rb").read())
out["kernel_sources_sha256"] = h.hexdigest()
json.dump(out, open(f"{O}/{tag}_extra_traffic.json", "w"), indent=1)
PY
# 5. SQ budgets
sed "s/r05sq/${TAG}sq/g; s/r05_sq_budgets/${TAG}_sq_budgets/g; s/pmc_r05sq/pmc_${TAG}sq/g; s/--no-cpu-baseline --no-ops/--no-cpu-baseline --no-ops --no-scaling-model/g" scripts/r05_sq_budgets.sh > /tmp/sqb.sh
bash /tmp/sqb.sh > $O/${TAG}_sq_budgets.txt 2>&1
tail -3 $O/${TAG}_headline.txt; tail -8 $O/${TAG}_ops_traffic.txt; tail -12 $O/${TAG}_sq_budgets.txt
