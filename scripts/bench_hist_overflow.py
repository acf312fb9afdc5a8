#!/usr/bin/env python3
"""stats on records LONGER than the dense length histogram (hist_cap = 65 536 bins): their lengths go to the overflow list
(PARITY.md "Limits": exact, slower) -- how much slower?  2 GB of FASTA wrapped at 60: records of 70 001 bases (every one
on the list) against records of 5 001 bases (none).  Usage: python scripts/bench_hist_overflow.py"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bigseqkit_amd as bsk


def fasta(nbases, total):
    rng = np.random.default_rng(1)
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=nbases)
    lines = [seq[i:i + 60].tobytes() for i in range(0, nbases, 60)]
    rec = b">r some description\n" + b"\n".join(lines) + b"\n"
    t = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    return t.repeat(max(1, total // len(rec))), len(rec)


for nb in (5001, 70001):
    t, rb = fasta(nb, 2 << 30)
    fr = bsk.SeqFrame(bsk.FORMAT_FASTA, [t])
    for opts in (bsk.SeqKitStatsOptions(), bsk.SeqKitStatsOptions().All(True)):
        bsk.StatsString("x", "N/A", fr, opts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            s = bsk.StatsString("x", "N/A", fr, opts)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(json.dumps({"bases_per_record": nb, "records": t.numel() // rb, "GB": round(t.numel() / 1e9, 2), "all": "a" in opts.to_json().lower() and '"All": true' in opts.to_json(), "ms_whole_call": round(ms, 3), "row": s.split("\n")[1][:90]}))
