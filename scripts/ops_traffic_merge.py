#!/usr/bin/env python3
"""HBM traffic per command with a FETCH_SIZE factor per access SHAPE.
Inputs: gpurun_out/traffic_<tag>_<op>.json (scripts/pmc_ops_traffic.sh: raw FETCH_SIZE / WRITE_SIZE per kernel),
gpurun_out/ops_<tag>.json (scripts/bench_ops.py: sizes and times), profiles/<tag>_fetch_calibration.json
(scripts/fetch_calibration.sh).  Output: profiles/<tag>_ops_traffic.json, read by bench.py (`traffic` per ops entry).

What FETCH_SIZE counts on gfx950 (profiles/r05_fetch_calibration.json): REQUESTS of the L2 to memory, 64 bytes each --
whether a request moves one 64-byte sector or both sectors of a 128-byte line.  A coalesced 16 B/lane stream asks for whole
lines: bytes = 2 x FETCH_SIZE (the guide's factor).  A gather asks for the sectors it touches: one 16-byte header load per
317-byte record 1.105 x, the 4-lane 16-byte gathers of the byte comparison 1.129 x, the segmented copy's unaligned but
contiguous reads 1.838 x (against the distinct sectors each probe is known to touch).  Per kernel:
    bytes read = 2 x (the part of FETCH_SIZE that its coalesced streams account for: the input it reads once, its table rows)
               + g x (the rest),  g = the factor of its gather shape
The x2-everywhere figure of rounds 2-4 stays in the file as `upper_bound`.
Usage: python scripts/ops_traffic_merge.py r05 [ops]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1]
ops = (sys.argv[2] if len(sys.argv) > 2 else "seq,subseq,grep,rmdup,translate").split(",")
calj = json.load(open(f"{ROOT}/profiles/{tag}_fetch_calibration.json"))
cal = calj["kernels"]
# Round 6 settled what a request moves (profiles/r06_fetch_calibration.json "discrimination": a probe that asks for ONE
# 64-byte sector of every 128-byte line takes 0.89 x the time of the streaming read of the same buffer and TCC_EA0_RDREQ
# counts one request per LINE, none of 32 bytes; the header probe's time x the stream's rate = the bytes of the LINES it
# touches): a request moves the LINE.  FETCH_SIZE tallies every request at 64 bytes, so bytes = 2 x FETCH_SIZE for every
# shape -- the "upper bound" of round 5 was the figure.  A calibration file without the discrimination (round 5) keeps the
# per-shape factors against the touched sectors.
moves_lines = calj.get("discrimination", {}).get("one_sector_ms_over_stream_ms", 0) >= 0.75
fk = "factor_vs_lines128" if moves_lines else "factor_vs_sectors64"
G = {"header": cal["calib_header16"][fk], "quad": cal["calib_quad_gather"][fk],
     "seg": cal["calib_seg_read"][fk], "stream": cal["calib_stream16"][fk]}
if moves_lines:
    G = {k: 2.0 for k in G}   # (one request = one 128-byte line, counted as 64 bytes: exactly 2, whatever part of the line was asked for)
res = json.load(open(f"{O}/ops_{tag}.json"))
key_of = {"seq": "seq -n", "subseq": "subseq", "grep": "grep -s", "locate": "locate", "rmdup": "rmdup", "translate": "translate"}


def shape_of(kernel, entry):
    """(coalesced bytes this kernel is known to stream, gather shape of the rest)"""
    n_in, nrec = entry["in_GB"] * 1e9, entry["records"]
    if kernel.startswith(("k_names<", "k_subseq_stream", "k_rmdup_stream", "k_filter", "k_index<", "k_stats", "k_fasta_starts")):
        return n_in, "header"          # the input once, coalesced; the rest: header / tile-edge probes
    if kernel.startswith(("k_rmdup_place", "k_rmdup_verify")):
        return 24.0 * nrec, "quad"     # table rows (first, l_head, l_seq, aux, start) coalesced; the rest: the byte comparison
    if kernel.startswith("k_seg_copy"):
        return 16.0 * nrec, "seg"      # segment sources + offsets coalesced; the rest: the survivors' text
    return None, "stream"              # everything else is counted as a coalesced stream (x2)


out = {"source": __doc__.split("Usage:")[0].strip(), "factors": G, "requests_move": "lines (2 x FETCH_SIZE everywhere)" if moves_lines else "sectors (per-shape factors)", "output_contract": os.environ.get("BSK_OUT", "contiguous"), "ops": {}}
for op in ops:
    tf = f"{O}/traffic_{tag}_{op}.json"
    if not os.path.exists(tf):
        continue
    t = json.load(open(tf))["kernels"]
    for name, v in res.items():
        if not name.startswith(key_of.get(op, op)):
            continue
        alg = (v["in_GB"] + v["out_GB"]) * 1e9
        if os.environ.get("BSK_OUT") == "slices" and op == "rmdup":
            alg = v["in_GB"] * 1e9 + 16.0 * v["records"]   # (the survivors are slices of the shard: not moved; bench.py's figure)
        if os.environ.get("BSK_OUT") == "slices" and op == "grep":
            import re
            m = re.search(r"hits=(\d+)", v.get("note", ""))
            if m:
                alg = v["in_GB"] * 1e9 + 16.0 * int(m.group(1))   # (the hits stay in the shard: 16 bytes of slice list each)
        kern, total, upper = {}, 0.0, 0.0
        for k, kv in t.items():
            calls = kv["dispatches"] / 2.0              # (two calls of the command per pass)
            raw = kv["fetch_KiB_mean"] * 1024.0 * calls  # FETCH_SIZE bytes per command call
            wr = kv["write_KiB_mean"] * 1024.0 * calls
            stream, shape = shape_of(k, v)
            if stream is None:
                rd = 2.0 * raw
            else:
                s_raw = min(raw, stream / 2.0)
                rd = 2.0 * s_raw + G[shape] * (raw - s_raw)
            total += rd + wr
            upper += 2.0 * raw + wr
            if rd + wr > 0.002 * alg:
                kern[k] = {"fetch_size_GB": round(raw / 1e9, 3), "read_GB": round(rd / 1e9, 3), "write_GB": round(wr / 1e9, 3),
                           "shape": shape, "launches_per_call": calls}
        out["ops"][name] = {"algorithmic_GB": round(alg / 1e9, 3), "traffic_GB": round(total / 1e9, 2),
                            "traffic_over_algorithmic": round(total / alg, 3), "upper_bound_GB_x2_everywhere": round(upper / 1e9, 2),
                            "upper_bound_over_algorithmic": round(upper / alg, 3), "ms": v["ms"], "kernels": kern}
h = hashlib.sha256()
for f in sorted(os.listdir(f"{ROOT}/bigseqkit_amd/csrc")):
    if f.endswith((".hip", ".hpp", ".inc")):
        h.update(open(f"{ROOT}/bigseqkit_amd/csrc/{f}", "rb").read())
out["kernel_sources_sha256"] = h.hexdigest()
json.dump(out, open(f"{ROOT}/profiles/{tag}_ops_traffic.json", "w"), indent=1)
for k, v in out["ops"].items():
    print("%-52s %8.2f ms  traffic %7.2f GB = %.3f x algorithmic   (x2 everywhere: %.3f x)" % (
        k[:52], v["ms"], v["traffic_GB"], v["traffic_over_algorithmic"], v["upper_bound_over_algorithmic"]))
