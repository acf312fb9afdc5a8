#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of scripts/gpu_round.sh (FETCH_SIZE, WRITE_SIZE; separate runs) into
HBM bytes per k_stats launch.  Usage: python scripts/pmc_traffic.py <tag>   (reads gpurun_out/pmc_{fetch,write}_<tag>/)

Corrections (MI355X_MICROARCH.md, HBM/rocprofv3 section + profiles/r01_calibration_stream_read.json):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of 16 B/lane coalesced reads
(calibrated with bsk_selftest_stream_read: 9,765,637 KB reported for 19,999,997,952 B read), hence x2 for reads."""
import csv
import glob
import json
import sys

tag = sys.argv[1]
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/gpu_round.sh {tag}) of "
                 "`python bench.py --steps 3 --warmup 1 --no-cpu-baseline`; bytes = FETCH_SIZE x 1024 x 2 "
                 "(gfx950 half-count of 16 B/lane coalesced reads, profiles/r01_calibration_stream_read.json) + WRITE_SIZE x 1024",
       "raw": {}, "per_launch": {}}
names = {"k_stats<true, false, true": "stats", "k_stats<true, true, true": "stats -a"}  # (<FASTQ, ALL, DPP[, ROLES]>)
for counter, d in (("FETCH_SIZE", "pmc_fetch_"), ("WRITE_SIZE", "pmc_write_")):
    for f in glob.glob(f"gpurun_out/{d}{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            for k, label in names.items():
                if k in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    out["raw"].setdefault(label, {}).setdefault(counter, []).append(float(row["Counter_Value"]))
for label, r in out["raw"].items():
    mean = {c: sum(v) / len(v) for c, v in r.items()}
    out["raw"][label] = {c: {"dispatches": len(v), "mean_KB": mean[c]} for c, v in r.items()}
    fetch = mean.get("FETCH_SIZE", 0) * 1024 * 2
    write = mean.get("WRITE_SIZE", 0) * 1024
    out["per_launch"][label] = {"fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write}
# the sources of the measured kernel: bench.py marks the figure as stale when they have changed since
import hashlib
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in ("stream_stats.hip", "stream_core_dev.hpp", "anchor_wave_dev.hpp"):
    h.update(open(os.path.join(root, "bigseqkit_amd", "csrc", f), "rb").read())
out["kernel_sources_sha256"] = h.hexdigest()
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["per_launch"], indent=1))
