#!/usr/bin/env bash
# build box, after `gpurun -- bash scripts/r06_evidence.sh r06`: what the evidence run left under gpurun_out/ becomes the
# tracked files under profiles/ (bench.py reads them and refuses traffic figures whose kernel-source hash is not the tree's)
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r06}
python scripts/pmc_traffic.py $TAG > /dev/null
BSK_OUT=slices python scripts/ops_traffic_merge.py $TAG
cp gpurun_out/prof_$TAG/stats_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
cp gpurun_out/prof_ops_$TAG/ops_kernel_stats.csv profiles/${TAG}_ops_kernel_stats.csv
cp gpurun_out/${TAG}_extra_traffic.json gpurun_out/${TAG}_sq_budgets.json gpurun_out/${TAG}_fetch_calibration.json profiles/
python - "$TAG" <<'PY'
import hashlib, json, os, sys
tag = sys.argv[1]
h = hashlib.sha256()
for f in sorted(os.listdir("bigseqkit_amd/csrc")):
    if f.endswith((".hip", ".hpp", ".inc")):
        h.update(open(f"bigseqkit_amd/csrc/{f}", "rb").read())
h3 = hashlib.sha256()   # (the headline file carries the hash of the three sources of k_stats: scripts/pmc_traffic.py)
for f in ("stream_stats.hip", "stream_core_dev.hpp", "anchor_wave_dev.hpp"):
    h3.update(open(f"bigseqkit_amd/csrc/{f}", "rb").read())
for f in ("pmc_traffic", "ops_traffic", "extra_traffic"):
    d = json.load(open(f"profiles/{tag}_{f}.json"))
    got = d.get("kernel_sources_sha256")
    want = h3.hexdigest() if f == "pmc_traffic" else h.hexdigest()
    print("%-16s %s" % (f, "hash matches the tree" if got == want else "STALE: " + str(got)[:12] + " != " + want[:12]))
PY
