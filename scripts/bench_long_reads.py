"""The hot-path commands on long-read FASTQ (reads of 2-30 kb on one line, ~2 GB built by tiling 2 000 random reads):
a survey for slow paths; HBM-resident, mean of 3 calls after a warm-up."""
import ctypes as C, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

rng = np.random.default_rng(7)
parts = []
for i in range(2000):
    L = int(rng.integers(2000, 30000))
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L)
    qual = rng.integers(35, 75, L, dtype=np.uint8)
    parts += [np.frombuffer(("@read%05d runid=abc ch=%d\n" % (i, i % 512)).encode(), dtype=np.uint8), seq, np.frombuffer(b"\n+\n", dtype=np.uint8), qual, np.frombuffer(b"\n", dtype=np.uint8)]
block = np.concatenate(parts)
reps = int(float(sys.argv[1]) * 1e9 / block.size) if len(sys.argv) > 1 else 64
t = torch.from_numpy(block).cuda().repeat(reps)
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""

def run(name, fn, opts, reps=3):
    out = _lib.Out()
    with bsk.Operator(name, json.dumps(opts), 0) as op:
        check(fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, 1, 0, None, C.byref(out)), op.ctx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            check(fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, 1, 0, None, C.byref(out)), op.ctx); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out.len

G, L, S, R, SS, T, ST = ("Grep", lib.bsk_grep_run), ("Locate", lib.bsk_locate_run), ("SeqTransform", lib.bsk_seq_run), ("RmDup", lib.bsk_rmdup_run), ("SubseqTransform", lib.bsk_subseq_run), ("Translate", lib.bsk_translate_run), ("Sort", lib.bsk_sort_run)
P12 = "ACGTTGCAAGCT"
CASES = [(S, {}), (S, {"Name": True}), (S, {"Reverse": True, "Complement": True}), (S, {"MinLen": 10000}), (S, {"MinQual": 20}), (S, {"Seq": True}),
         (G, {"Pattern": [P12], "BySeq": True}), (G, {"Pattern": ["ACGTTGCAAGCTACGTAA"], "BySeq": True}), (G, {"Pattern": [P12], "BySeq": True, "MaxMismatch": 1}),
         (G, {"Pattern": ["read00077"]}), (L, {"Pattern": ["ACGTTGCAAGCTACGTAA"]}), (L, {"Pattern": [P12]}),
         (SS, {"Region": "1:1000"}), (SS, {"Region": "101:-101"}), (R, {"BySeq": True}), (R, {}), (T, {"Frame": ["1"]}), (T, {"Frame": ["6"]}),
         (ST, {"ByLength": True}), (ST, {})]
print("long reads: %.2f GB, %d records" % (t.numel() / 1e9, 2000 * reps), flush=True)
for (name, fn), opts in CASES:
    if ONLY not in name + " " + json.dumps(opts): continue
    try:
        ms, ol = run(name, fn, opts)
        print("%-16s %-66s %9.2f ms  out %6.2f GB  %6.0f GB/s" % (name, json.dumps(opts), ms, ol / 1e9, (t.numel() + ol) / ms / 1e6), flush=True)
    except Exception as e:
        print("%-16s %-66s ERROR %s" % (name, json.dumps(opts), str(e)[:90]), flush=True)
