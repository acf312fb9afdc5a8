#!/usr/bin/env bash
# FETCH_SIZE on gfx950 for the gather shapes of this engine (scripts/experiments/fetch_calib.hip): the counter under
# rocprofv3 --pmc FETCH_SIZE against the bytes every kernel is KNOWN to touch -> gpurun_out/<tag>_fetch_calibration.json
# (copy to profiles/).  Usage on the GPU box: bash scripts/fetch_calibration.sh r05
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
[ -x $R/scripts/experiments/bin/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/scripts/experiments/fetch_calib.hip -o $R/scripts/experiments/bin/fetch_calib
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_$TAG -o pmc -- $R/scripts/experiments/bin/fetch_calib > $O/calib_$TAG.txt 2> $O/calib_$TAG.err
cd $R
python - "$O" "$TAG" <<'PY'
import csv, glob, json, sys, collections
O, tag = sys.argv[1:3]
known = {}
for line in open(f"{O}/calib_{tag}.txt"):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        known[d["kernel"]] = d
vals = collections.defaultdict(list)
for f in glob.glob(f"{O}/calib_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024.0)   # KiB -> bytes
out = {"source": "scripts/fetch_calibration.sh: rocprofv3 --pmc FETCH_SIZE over scripts/experiments/fetch_calib (8 GB of 317-byte "
                 "records; two launches per kernel, the second one reported); bytes_per_counted_byte = what one reported byte "
                 "stands for if the kernel moved exactly the distinct 64-byte sectors it touched",
       "kernels": {}}
for k, d in known.items():
    v = vals.get(k, [])
    if not v:
        continue
    raw = v[-1]
    out["kernels"][k] = dict(d, fetch_size_bytes=int(raw), factor_vs_requested=round(d["requested_bytes"] / raw, 4),
                             factor_vs_sectors64=round(d["sectors64_bytes"] / raw, 4), factor_vs_lines128=round(d["lines128_bytes"] / raw, 4))
json.dump(out, open(f"{O}/{tag}_fetch_calibration.json", "w"), indent=1)
for k, d in out["kernels"].items():
    print("%-20s FETCH_SIZE %8.3f GB  requested %8.3f  sectors64 %8.3f  lines128 %8.3f   factors %.3f / %.3f / %.3f" % (
        k, d["fetch_size_bytes"] / 1e9, d["requested_bytes"] / 1e9, d["sectors64_bytes"] / 1e9, d["lines128_bytes"] / 1e9,
        d["factor_vs_requested"], d["factor_vs_sectors64"], d["factor_vs_lines128"]))
PY
