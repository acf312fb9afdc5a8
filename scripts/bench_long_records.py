#!/usr/bin/env python3
"""Genome-like input: a few very long FASTA records (60-column lines).  Usage: bench_long_records.py [MB per record] [records]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
nrec = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(7)
parts = []
for r in range(nrec):
    L = int(mb * 1e6) // 61 * 60
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)]
    lines = np.empty((L // 60, 61), dtype=np.uint8)
    lines[:, :60] = bases.reshape(-1, 60)
    lines[:, 60] = 10
    parts.append(np.frombuffer(b">chr%d test\n" % (r + 1), dtype=np.uint8))
    parts.append(lines.reshape(-1))
data = np.concatenate(parts)
t = torch.from_numpy(data).cuda()
n = t.numel()
res = {"bytes": int(n), "records": nrec}
def run(name, op_name, fn, opts):
    out = _lib.Out()
    with bsk.Operator(op_name, json.dumps(opts), 0) as op:
        check(fn(op.ctx, C.c_void_p(t.data_ptr()), n, 1, 0, 0, None, C.byref(out)), op.ctx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(fn(op.ctx, C.c_void_p(t.data_ptr()), n, 1, 0, 0, None, C.byref(out)), op.ctx)
        torch.cuda.synchronize()
        res[name] = {"ms": round((time.perf_counter() - t0) * 1e3, 2), "out_bytes": out.len, "elements": out.records}
run("locate 12-mer", "Locate", lib.bsk_locate_run, {"Pattern": ["ACGTTGCAAGCT"]})
run("grep -s 12-mer", "Grep", lib.bsk_grep_run, {"Pattern": ["ACGTTGCAAGCT"], "BySeq": True})
run("seq (re-emit)", "SeqTransform", lib.bsk_seq_run, {})
run("subseq -r 1000:2000", "SubseqTransform", lib.bsk_subseq_run, {"Region": "1000:2000"})
run("translate -f 1", "Translate", lib.bsk_translate_run, {"Frame": ["1"], "AllowUnknownCodon": True})
with bsk.Operator("Stats", "{}", 0) as op:
    for _ in range(2):
        check(lib.bsk_stats_reset(op.ctx, None), op.ctx)
        t0 = time.perf_counter()
        check(lib.bsk_stats_run(op.ctx, C.c_void_p(t.data_ptr()), n, 1, 0, 0, None, None), op.ctx)
        torch.cuda.synchronize()
    res["stats"] = {"ms": round((time.perf_counter() - t0) * 1e3, 2)}
print(json.dumps(res))
