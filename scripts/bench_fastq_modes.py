"""Less common modes of the hot-path commands on FASTQ-150 (a survey for slow paths): HBM-resident, mean of 3 calls."""
import ctypes as C, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

GB = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
rb = lib.bsk_synth_record_bytes(0); n = int(GB * 1e9) // rb * rb
t = torch.empty(n, dtype=torch.uint8, device="cuda")
check(lib.bsk_synth_device(0, 42, 3, 0, C.c_void_p(t.data_ptr()), n, 0, None)); torch.cuda.synchronize()

def run(name, fn, opts, reps=3):
    out = _lib.Out()
    with bsk.Operator(name, json.dumps(opts), 0) as op:
        check(fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, 1, 0, None, C.byref(out)), op.ctx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            check(fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, 1, 0, None, C.byref(out)), op.ctx); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out.len

G, L, S, R, SS, T = ("Grep", lib.bsk_grep_run), ("Locate", lib.bsk_locate_run), ("SeqTransform", lib.bsk_seq_run), ("RmDup", lib.bsk_rmdup_run), ("SubseqTransform", lib.bsk_subseq_run), ("Translate", lib.bsk_translate_run)
P12 = "ACGTTGCAAGCT"
CASES = [(G, {"Pattern": [P12], "BySeq": True}), (G, {"Pattern": [P12], "BySeq": True, "IgnoreCase": True}),
         (G, {"Pattern": [P12], "BySeq": True, "InvertMatch": True}),
         (G, {"Pattern": ["ACGTTGCA"], "BySeq": True}),
         (G, {"Pattern": ["ACGTTGCANGCT"], "BySeq": True, "Degenerate": True}),
         (G, {"Pattern": [P12], "BySeq": True, "MaxMismatch": 1}),
         (G, {"Pattern": ["ACGT+GCAAGCT"], "BySeq": True, "UseRegexp": True}),
         (G, {"Pattern": ["^S00000001"], "UseRegexp": True}),
         (G, {"Pattern": [P12], "BySeq": True, "Region": "1:100"}),
         (L, {"Pattern": [P12]}), (L, {"Pattern": [P12], "IgnoreCase": True}), (L, {"Pattern": ["ACGTTGCANGCT"], "Degenerate": True}),
         (L, {"Pattern": [P12], "MaxMismatch": 1}), (L, {"Pattern": ["ACGT+GCAAGCT"], "UseRegexp": True}),
         (S, {}), (S, {"RemoveGaps": True}), (S, {"MinLen": 100}), (S, {"UpperCase": True}), (S, {"Dna2rna": True}),
         (S, {"Seq": True}), (S, {"Qual": True}), (S, {"MinQual": 20}), (S, {"ValidateSeq": True}),
         (R, {"ByName": True}), (R, {"BySeq": True, "IgnoreCase": True}),
         (SS, {"Region": "10:100"}), (T, {"Frame": ["6"]}), (T, {"Frame": ["1"], "Trim": True})]
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""   # substring of "<operator> <options>": run those cases only
for (name, fn), opts in CASES:
    if ONLY not in name + " " + json.dumps(opts): continue
    try:
        ms, ol = run(name, fn, opts)
        print("%-16s %-66s %9.2f ms  out %6.2f GB  %6.0f GB/s" % (name, json.dumps(opts), ms, ol / 1e9, (t.numel() + ol) / ms / 1e6), flush=True)
    except Exception as e:
        print("%-16s %-66s ERROR %s" % (name, json.dumps(opts), str(e)[:90]), flush=True)
