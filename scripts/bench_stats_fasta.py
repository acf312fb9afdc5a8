#!/usr/bin/env python3
"""`stats` on the synthetic FASTA configs (C1-like FASTA-1k, C4 FASTA-5k), HBM-resident.  Usage: bench_stats_fasta.py"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
res = {}
for name, kind, gb in (("FASTA-1k 1 GB (C1)", 1, 1.0), ("FASTA-1k 20 GB", 1, 20.0), ("FASTA-5k 50 GB (C4 input)", 2, 50.0)):
    rb = lib.bsk_synth_record_bytes(kind)
    n = int(gb * 1e9) // rb * rb
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    check(lib.bsk_synth_device(kind, 42, 0, 0, C.c_void_p(t.data_ptr()), n, 0, None))
    torch.cuda.synchronize()
    for opts in ({}, {"All": True}):
        with bsk.Operator("Stats", json.dumps(opts), 0) as op:
            keys, vals, cnt = (C.c_int64 * 65536)(), (C.c_int64 * 65536)(), C.c_size_t()
            def step():
                check(lib.bsk_stats_reset(op.ctx, None), op.ctx)
                check(lib.bsk_stats_run(op.ctx, C.c_void_p(t.data_ptr()), n, 1, bsk.FORMAT_FASTA, 0, None, None), op.ctx)
                check(lib.bsk_stats_collect(op.ctx, None, keys, vals, 65536, C.byref(cnt)), op.ctx)
            step(); step()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                step()
            dt = (time.perf_counter() - t0) / reps
            res[name + (" -a" if opts else "")] = {"ms": round(dt * 1e3, 3), "GBps": round(n / dt / 1e9, 1), "frac_of_8TBps": round(n / dt / 8e12, 3),
                                                  "M_records_per_s": round(n / rb / dt / 1e6, 1)}
    del t
print(json.dumps(res))
