"""The hot-path commands on input SHAPES other than the BASELINE layouts (a survey for slow paths): HBM-resident, mean of 3
calls after a warm-up.   python scripts/bench_shapes.py <shape> [GB] [filter]
shapes: chrom1line (FASTA, 8 records of ~GB/8 on ONE line each), chrom60 (the same wrapped at 60), tinyfa (FASTA records of
18-30 bases), protein (FASTA, 50-900 residues wrapped at 60), short36 (FASTQ reads of 36 bases), mixedfq (FASTQ, lengths 20-20 000)"""
import ctypes as C, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

shape = sys.argv[1]
GB = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
ONLY = sys.argv[3] if len(sys.argv) > 3 else ""
rng = np.random.default_rng(11)
A = np.frombuffer(b"ACGT", dtype=np.uint8)
NL = np.frombuffer(b"\n", dtype=np.uint8)

def b(s): return np.frombuffer(s.encode(), dtype=np.uint8)
def wrap(seq, w):
    if w == 0 or seq.size == 0: return np.concatenate([seq, NL])
    full = seq.size // w
    body = np.concatenate([seq[:full * w].reshape(full, w), np.full((full, 1), 10, np.uint8)], axis=1).reshape(-1)
    return np.concatenate([body, seq[full * w:], NL]) if seq.size % w else body

fastq = shape in ("short36", "mixedfq")
parts, nrec_block = [], 0
if shape in ("chrom1line", "chrom60"):
    per = int(GB * 1e9 / 8)
    for i in range(8):
        parts += [b(">chr%d assembled\n" % (i + 1)), wrap(rng.choice(A, per), 0 if shape == "chrom1line" else 60)]
    block, reps, nrec_block = np.concatenate(parts), 1, 8
else:
    for i in range(20000):
        if shape == "tinyfa":
            parts += [b(">p%05d\n" % i), wrap(rng.choice(A, int(rng.integers(18, 31))), 60)]
        elif shape == "protein":
            parts += [b(">sp|P%05d|PROT_%d some protein OS=Homo sapiens\n" % (i, i)), wrap(rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8), int(rng.integers(50, 900))), 60)]
        else:
            L = 36 if shape == "short36" else int(rng.choice([20, 50, 150, 250, 1000, 20000], p=[.2, .2, .3, .2, .09, .01]))
            parts += [b("@r%05d/1\n" % i), rng.choice(A, L), b("\n+\n"), rng.integers(35, 75, L, dtype=np.uint8), NL]
    block = np.concatenate(parts); reps = max(1, int(GB * 1e9 / block.size)); nrec_block = 20000
t = torch.from_numpy(block).cuda().repeat(reps)
fmt = 1 if fastq else 0

def run(name, fn, opts, reps=3):
    out = _lib.Out()
    with bsk.Operator(name, json.dumps(opts), 0) as op:
        check(fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, fmt, 0, None, C.byref(out)), op.ctx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            check(fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, fmt, 0, None, C.byref(out)), op.ctx); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out.len

def stats(opts):
    o = bsk.SeqKitStatsOptions()
    for k, v in opts.items(): getattr(o, k)(v)
    fr = bsk.SeqFrame(fmt, [t]); bsk.stats_map(fr, o); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): bsk.stats_map(fr, o)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 3 * 1e3, 0

G, L, S, R, SS, T, ST = ("Grep", lib.bsk_grep_run), ("Locate", lib.bsk_locate_run), ("SeqTransform", lib.bsk_seq_run), ("RmDup", lib.bsk_rmdup_run), ("SubseqTransform", lib.bsk_subseq_run), ("Translate", lib.bsk_translate_run), ("Sort", lib.bsk_sort_run)
PAT = "MKVLA" if shape == "protein" else "ACGTTGCAAGCT"
CASES = [("Stats", {}), ("Stats", {"All": True}), (S, {}), (S, {"Name": True}), (S, {"Config": {"LineWidth": 0}}), (S, {"Config": {"LineWidth": 80}}),
         (S, {"Reverse": True, "Complement": True}), (S, {"UpperCase": True}),
         (G, {"Pattern": [PAT], "BySeq": True}), (G, {"Pattern": ["chr3"]}), (L, {"Pattern": [PAT]}),
         (SS, {"Region": "1:20"}), (SS, {"Region": "11:-11"}), (R, {"BySeq": True}), (R, {}), (T, {"Frame": ["1"]}), (T, {"Frame": ["6"]}),
         (ST, {"ByLength": True}), (ST, {})]
print("%s: %.2f GB, %d records" % (shape, t.numel() / 1e9, nrec_block * reps), flush=True)
for what, opts in CASES:
    name = what if isinstance(what, str) else what[0]
    if ONLY not in name + " " + json.dumps(opts): continue
    if shape == "protein" and name == "Translate": continue
    try:
        ms, ol = stats(opts) if name == "Stats" else run(what[0], what[1], opts)
        print("%-16s %-52s %9.2f ms  out %6.2f GB  %6.0f GB/s" % (name, json.dumps(opts), ms, ol / 1e9, (t.numel() + ol) / ms / 1e6), flush=True)
    except Exception as e:
        print("%-16s %-52s ERROR %s" % (name, json.dumps(opts), str(e)[:100]), flush=True)
