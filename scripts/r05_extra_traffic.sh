#!/usr/bin/env bash
# round 5: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the kernels the ops evidence does not reach: the FASTA stats
# passes (scripts/bench_stats_fasta.py) and the one-pass translate of records that differ (scripts/bench_translate_var.py)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmcx_fa_$c -o pmc -- python $R/scripts/bench_stats_fasta.py > $O/pmcx_fa_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/pmcx_tr_$c -o pmc -- python $R/scripts/bench_translate_var.py 50 1 > $O/pmcx_tr_$c.log 2>&1
done
cd $R
python - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(list)
for f in glob.glob("$O/pmcx_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("bsk::", "")
        if name.startswith("k_stats<false") or name.startswith("k_translate_stream") or name.startswith("k_stats_stitch"):
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {"source": "scripts/r05_extra_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); KiB counters; reads x 2 "
                 "(coalesced 16-byte streams: profiles/r05_fetch_calibration.json); the largest dispatches of each kernel (the 50 GB inputs)",
       "kernels": {}}
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
    print(k, e)
json.dump(out, open("$O/r05_extra_traffic.json", "w"), indent=1)
PY
