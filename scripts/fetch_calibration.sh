#!/usr/bin/env bash
# FETCH_SIZE on gfx950 for the gather shapes of this engine (scripts/experiments/fetch_calib.hip): the counter under
# rocprofv3 --pmc FETCH_SIZE against the bytes every kernel is KNOWN to touch -> gpurun_out/<tag>_fetch_calibration.json
# (copy to profiles/).  Usage on the GPU box: bash scripts/fetch_calibration.sh r05
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
[ -x $R/scripts/experiments/bin/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/scripts/experiments/fetch_calib.hip -o $R/scripts/experiments/bin/fetch_calib
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_$TAG -o pmc -- $R/scripts/experiments/bin/fetch_calib > $O/calib_$TAG.txt 2> $O/calib_$TAG.err
# round 6: the fabric request counters themselves, where this rocprofv3 knows them (their own pass; a failure is recorded, not fatal)
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B --output-format csv -d $O/calib_${TAG}_ea -o pmc -- $R/scripts/experiments/bin/fetch_calib > $O/calib_${TAG}_ea.txt 2> $O/calib_${TAG}_ea.err || echo "TCC_EA0_RDREQ pass failed" >> $O/calib_${TAG}_ea.err
# ... and the probes once more without a profiler: the timings the discrimination rests on
$R/scripts/experiments/bin/fetch_calib > $O/calib_${TAG}_plain.txt 2>&1
cd $R
python - "$O" "$TAG" <<'PY'
import csv, glob, json, sys, collections
O, tag = sys.argv[1:3]
known, ms = {}, {}
for line in open(f"{O}/calib_{tag}.txt"):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        if "kernel" in d:
            known[d["kernel"]] = d
try:
    for line in open(f"{O}/calib_{tag}_plain.txt"):
        line = line.strip()
        if line.startswith("{"):
            d = json.loads(line)
            if "timing" in d:
                ms[d["timing"]] = d["ms"]
except OSError:
    pass
ea = collections.defaultdict(dict)
for f in glob.glob(f"{O}/calib_{tag}_ea/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ea[r["Kernel_Name"].split("(")[0]].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
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
for k, d in out["kernels"].items():
    if k in ms:
        d["ms_unprofiled"] = ms[k]
    if k in ea:
        d["fabric_requests"] = {c: v[-1] for c, v in ea[k].items()}
if "calib_stream16" in ms and "calib_one_sector" in ms:
    t_s, t_1, t_2a, t_2t = ms["calib_stream16"], ms["calib_one_sector"], ms.get("calib_two_sectors_apart", 0), ms.get("calib_two_sectors_together", 0)
    rate = known["calib_stream16"]["requested_bytes"] / (t_s * 1e-3)
    out["discrimination"] = {
        "stream_rate_GBps": round(rate / 1e9, 1),
        "one_sector_ms_over_stream_ms": round(t_1 / t_s, 3),
        "two_sectors_apart_ms_over_stream_ms": round(t_2a / t_s, 3) if t_2a else None,
        "two_sectors_together_ms_over_stream_ms": round(t_2t / t_s, 3) if t_2t else None,
        "bytes_moved_by_one_sector_probe_at_the_stream_rate": int(rate * t_1 * 1e-3),
        "bytes_if_requests_move_sectors": known["calib_one_sector"]["sectors64_bytes"],
        "bytes_if_requests_move_lines": known["calib_one_sector"]["lines128_bytes"],
        "reading": "a probe that asks for ONE 64-byte sector of every 128-byte line: ~0.5 x the stream's time = a request moves its "
                   "sector (FETCH_SIZE x 1 for such gathers), ~1.0 x = it moves the line (FETCH_SIZE x 2, as for streams)",
        "conclusion": ("a request moves the 128-byte LINE: bytes = 2 x FETCH_SIZE for every access shape" if t_1 / t_s >= 0.75 else
                       "a request moves the touched 64-byte sector: bytes = FETCH_SIZE x the per-shape factor against sectors64")}
json.dump(out, open(f"{O}/{tag}_fetch_calibration.json", "w"), indent=1)
print(json.dumps(out.get("discrimination"), indent=1))
for k, d in out["kernels"].items():
    print("%-20s FETCH_SIZE %8.3f GB  requested %8.3f  sectors64 %8.3f  lines128 %8.3f   factors %.3f / %.3f / %.3f" % (
        k, d["fetch_size_bytes"] / 1e9, d["requested_bytes"] / 1e9, d["sectors64_bytes"] / 1e9, d["lines128_bytes"] / 1e9,
        d["factor_vs_requested"], d["factor_vs_sectors64"], d["factor_vs_lines128"]))
PY
