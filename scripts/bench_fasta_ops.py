"""The hot-path commands on FASTA (10 GB of 1 kb records wrapped at 60, and of 5 kb CDS records): a survey for slow paths;
HBM-resident, mean of 3 calls after a warm-up."""
import ctypes as C, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

def synth(kind, nbytes):
    rb = lib.bsk_synth_record_bytes(kind); n = int(nbytes) // rb * rb
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    check(lib.bsk_synth_device(kind, 42, 0, 0, C.c_void_p(t.data_ptr()), n, 0, None)); torch.cuda.synchronize()
    return t, n // rb

def run(name, fn, opts, t, reps=3):
    out = _lib.Out()
    with bsk.Operator(name, json.dumps(opts), 0) as op:
        check(fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, 0, 0, None, C.byref(out)), op.ctx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            check(fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, 0, 0, None, C.byref(out)), op.ctx); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out.len

GB = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""   # substring of "<layout> <operator> <options>": those cases only
CASES = [("Grep", lib.bsk_grep_run, {"Pattern": ["ACGTTGCAAGCT"], "BySeq": True}),
         ("Grep", lib.bsk_grep_run, {"Pattern": ["S0000000123"]}),
         ("Locate", lib.bsk_locate_run, {"Pattern": ["ACGTTGCAAGCT"]}),
         ("SubseqTransform", lib.bsk_subseq_run, {"Region": "1:50"}),
         ("SubseqTransform", lib.bsk_subseq_run, {"Region": "100:-100"}),
         ("RmDup", lib.bsk_rmdup_run, {"BySeq": True}),
         ("RmDup", lib.bsk_rmdup_run, {}),
         ("Translate", lib.bsk_translate_run, {"Frame": ["1"]}),
         ("Translate", lib.bsk_translate_run, {"Frame": ["6"]}),
         ("Translate", lib.bsk_translate_run, {"Frame": ["6"], "Trim": True}),
         ("Sort", lib.bsk_sort_run, {"ByLength": True}),
         ("Sort", lib.bsk_sort_run, {})]
for kind, name in ((1, "FASTA-1k"), (2, "FASTA-5k")):
    t, n = synth(kind, GB * 1e9)
    for op, fn, opts in CASES:
        if ONLY not in name + " " + op + " " + json.dumps(opts): continue
        try:
            ms, ol = run(op, fn, opts, t)
            print("%-9s %-16s %-52s %9.2f ms  out %6.2f GB  %6.0f GB/s" % (name, op, json.dumps(opts), ms, ol / 1e9, (t.numel() + ol) / ms / 1e6), flush=True)
        except Exception as e:
            print("%-9s %-16s %-52s ERROR %s" % (name, op, json.dumps(opts), str(e)[:80]), flush=True)
    del t
