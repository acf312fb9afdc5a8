#!/usr/bin/env python3
"""Merge the outputs of scripts/ops_evidence.sh (gpurun_out/ops_<tag>.json, traffic_<tag>_<op>.json, prof_ops_<tag>/) into
profiles/<tag>_ops.json (one `roofline` object per command) and profiles/<tag>_ops_kernel_stats.csv.
Usage: python scripts/ops_evidence_merge.py r02 [ops]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1]
ops = (sys.argv[2] if len(sys.argv) > 2 else "seq,subseq,grep,locate,rmdup,translate").split(",")
res = json.load(open(f"{O}/ops_{tag}.json"))
key_of = {"seq": "seq -n", "subseq": "subseq", "grep": "grep -s", "locate": "locate", "rmdup": "rmdup", "translate": "translate"}
out = {"source": "scripts/ops_evidence.sh: bench_ops.py 1 3 (mean of 3 calls after a warm-up; the whole operator call -- record table, "
                 "sizes, scan, emit -- HIP-synchronised wall clock, data resident in HBM).  traffic: rocprofv3 --pmc FETCH_SIZE / "
                 "--pmc WRITE_SIZE in separate passes over bench_ops.py 1 1 (two calls of the command), summed over the command's "
                 "kernels and halved.  fetch is reported raw (x1) and with the gfx950 correction (x2) that is calibrated for "
                 "16 B/lane coalesced streaming reads (profiles/r01_calibration_stream_read.json); kernels that also gather with "
                 "narrower or unaligned loads lie between the two, so traffic_GB is an upper bound",
       "ops": {}}
for op in ops:
    tf = f"{O}/traffic_{tag}_{op}.json"
    traffic = raw = None
    per_kernel = None
    if os.path.exists(tf):
        t = json.load(open(tf))["kernels"]
        traffic = sum(v["total_GB_all_dispatches"] for v in t.values()) / 2.0
        raw = sum((v["fetch_GB_corrected_x2"] / 2.0 + v["write_GB"]) * v["dispatches"] for v in t.values()) / 2.0
        per_kernel = {k: {"fetch_GB_x2": v["fetch_GB_corrected_x2"], "write_GB": v["write_GB"], "dispatches_per_call": v["dispatches"] / 2}
                      for k, v in list(t.items())[:8]}
    for name, v in res.items():
        if name.startswith(key_of.get(op, op)):
            alg = v["in_GB"] + v["out_GB"]
            v["roofline"] = {"bound": "hbm", "achieved": v["algorithmic_GBps"], "peak": 8000.0, "unit": "GB/s", "frac": v["frac_of_8TBps"],
                             "algorithmic_GB": round(alg, 3),
                             "traffic_GB": None if traffic is None else round(traffic, 2),
                             "traffic_GB_fetch_uncorrected": None if raw is None else round(raw, 2),
                             "traffic_over_algorithmic": None if traffic is None else round(traffic / alg, 3)}
            if per_kernel:
                v["kernels_traffic_per_dispatch"] = per_kernel
            out["ops"][name] = v
json.dump(out, open(f"{ROOT}/profiles/{tag}_ops.json", "w"), indent=1)
src = f"{O}/prof_ops_{tag}/ops_kernel_stats.csv"
if os.path.exists(src):
    shutil.copy(src, f"{ROOT}/profiles/{tag}_ops_kernel_stats.csv")
for k, v in out["ops"].items():
    r = v["roofline"]
    print("%-52s %8.2f ms  frac %.3f  traffic %s GB (x%s)" % (k[:52], v["ms"], r["frac"], r["traffic_GB"], r["traffic_over_algorithmic"]))
