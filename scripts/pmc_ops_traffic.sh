#!/usr/bin/env bash
# HBM traffic (rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in their own passes) of the kernels of some bench_ops.py commands:
#   bash scripts/pmc_ops_traffic.sh translate 1.0 [out.json]
# Per kernel: dispatches, mean FETCH_SIZE / WRITE_SIZE (KiB) per dispatch, and bytes with the gfx950 read correction
# (x2 for 16 B/lane coalesced reads, profiles/r01_calibration_stream_read.json; writes as reported).
OPS=${1:-translate}; SCALE=${2:-1.0}; OUT=${3:-}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
TAG=$(echo $OPS | tr ',' '_')
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmct_${TAG}_f -o pmc -- python $R/scripts/bench_ops.py $SCALE 1 $OPS > $O/pmct_${TAG}_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmct_${TAG}_w -o pmc -- python $R/scripts/bench_ops.py $SCALE 1 $OPS > $O/pmct_${TAG}_w.log 2>&1
cd $R
python - "$O" "$TAG" "$OPS" "$SCALE" "$OUT" <<'PY'
import csv, glob, json, sys, collections
O, tag, ops, scale, out = sys.argv[1:6]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for counter, d in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
    for f in glob.glob(f"{O}/pmct_{tag}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("bsk::", "").split("(")[0]
            if name.startswith("k_synth") or "rocclr" in name or "at::native" in name: continue
            agg[name][counter].append(float(r["Counter_Value"]))
res = {}
for name, c in agg.items():
    f, w = c.get("FETCH_SIZE", []), c.get("WRITE_SIZE", [])
    fm = sum(f) / len(f) if f else 0.0
    wm = sum(w) / len(w) if w else 0.0
    res[name] = {"dispatches": max(len(f), len(w)), "fetch_KiB_mean": round(fm, 1), "write_KiB_mean": round(wm, 1),
                 "fetch_GB_corrected_x2": round(fm * 1024 * 2 / 1e9, 3), "write_GB": round(wm * 1024 / 1e9, 3),
                 "total_GB_all_dispatches": round((sum(f) * 2 + sum(w)) * 1024 / 1e9, 3)}
doc = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of scripts/bench_ops.py {scale} 1 {ops} "
                 "(two calls per command: warm-up + 1 timed); fetch x2 = gfx950 correction for 16 B/lane coalesced reads",
       "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["total_GB_all_dispatches"]))}
if out: json.dump(doc, open(out, "w"), indent=1)
for k, v in list(doc["kernels"].items())[:14]:
    print("%-44s n=%-3d fetch %9.3f GB  write %9.3f GB" % (k[:44], v["dispatches"], v["fetch_GB_corrected_x2"], v["write_GB"]))
PY
