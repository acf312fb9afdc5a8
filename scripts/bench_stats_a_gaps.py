#!/usr/bin/env python3
"""`stats -a` on FASTQ by line roles: what the gap count costs by data and gap letters (HBM-resident, mean of 5 calls).
The per-letter compare only runs for wave rows that hold a byte at or below the largest gap letter on a sequence line:
clean reads with the default letters never take it, reads with gaps or gap letters above the bases do.
    python scripts/bench_stats_a_gaps.py [GB]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bigseqkit_amd as bsk

GB = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
rng = np.random.default_rng(5)


def reads(alphabet, L, n=20000):
    parts = []
    A = np.frombuffer(alphabet, dtype=np.uint8)
    for i in range(n):
        parts += [np.frombuffer(b"@r%05d/1\n" % i, dtype=np.uint8), rng.choice(A, L), np.frombuffer(b"\n+\n", dtype=np.uint8),
                  rng.integers(35, 75, L, dtype=np.uint8), np.frombuffer(b"\n", dtype=np.uint8)]
    block = np.concatenate(parts)
    return torch.from_numpy(block).cuda().repeat(max(1, int(GB * 1e9 / block.size)))


def stats(t, opts):
    o = bsk.SeqKitStatsOptions()
    for k, v in opts.items(): getattr(o, k)(v)
    fr = bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]); bsk.stats_map(fr, o); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): bsk.stats_map(fr, o)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3


for label, alphabet, L in (("ACGT reads of 150", b"ACGT", 150), ("1 % '-' in reads of 150", b"ACGT" * 25 + b"-", 150),
                           ("ACGT reads of 36", b"ACGT", 36), ("ACGT reads of 10 000", b"ACGT", 10000)):
    t = reads(alphabet, L)
    for opts in ({}, {"All": True}, {"All": True, "GapLetters": "N-"}, {"All": True, "GapLetters": "ACGT"}):
        ms = stats(t, opts)
        print("%-26s %-40s %8.2f ms  %6.0f GB/s" % (label, json.dumps(opts), ms, t.numel() / ms / 1e6), flush=True)
    del t
