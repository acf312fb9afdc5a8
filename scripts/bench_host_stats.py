#!/usr/bin/env python3
"""End-to-end `stats` when the boundary is handed HOST buffers (PCIe-inclusive; never the headline number):
pinned vs pageable memory, chunked double-buffered H2D overlapped with the kernels.  Usage: bench_host_stats.py [GB]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
REC = 317
nrec = int(gb * 1e9) // REC
n = nrec * REC
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
check(lib.bsk_synth_device(0, 42, 0, 0, C.c_void_p(dev.data_ptr()), n, 0, None))
pinned = torch.empty(n, dtype=torch.uint8, pin_memory=True)
pinned.copy_(dev)
torch.cuda.synchronize()
res = {"bytes": n, "records": nrec}


def run(ptr, label, reps):
    with bsk.Operator("Stats", "{}", 0) as op:
        vlen = lib.bsk_stats_vector_len(op.ctx)
        keys, vals, cnt = (C.c_int64 * 4096)(), (C.c_int64 * 4096)(), C.c_size_t()
        best = None
        for _ in range(reps):
            check(lib.bsk_stats_reset(op.ctx, None), op.ctx)
            t0 = time.perf_counter()
            check(lib.bsk_stats_run(op.ctx, C.c_void_p(ptr), n, 0, bsk.FORMAT_FASTQ, 0, None, None), op.ctx)
            check(lib.bsk_stats_collect(op.ctx, None, keys, vals, 4096, C.byref(cnt)), op.ctx)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        m = dict(zip(keys[:cnt.value], vals[:cnt.value]))
        assert m.get(150) == nrec, m
        res[label] = {"s": round(best, 4), "GB_per_s": round(n / best / 1e9, 2), "M_records_per_s": round(nrec / best / 1e6, 1)}


run(pinned.data_ptr(), "pinned_host", 3)
pageable = pinned[: min(n, 4 * 10**9 // REC * REC)].clone().numpy() if False else None
sub = min(n, 4 * 10**9 // REC * REC)
pg = torch.empty(sub, dtype=torch.uint8)
pg.copy_(pinned[:sub])
n_saved, nrec_saved = n, nrec
n, nrec = sub, sub // REC
run(pg.data_ptr(), "pageable_host_4GB", 2)
n, nrec = n_saved, nrec_saved
t0 = time.perf_counter()
with bsk.Operator("Stats", "{}", 0) as op:
    check(lib.bsk_stats_reset(op.ctx, None), op.ctx)
    check(lib.bsk_stats_run(op.ctx, C.c_void_p(dev.data_ptr()), n, 1, bsk.FORMAT_FASTQ, 0, None, None), op.ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    check(lib.bsk_stats_run(op.ctx, C.c_void_p(dev.data_ptr()), n, 1, bsk.FORMAT_FASTQ, 0, None, None), op.ctx)
    torch.cuda.synchronize()
res["hbm_resident_same_bytes"] = {"s": round(time.perf_counter() - t0, 4)}
print(json.dumps(res))
