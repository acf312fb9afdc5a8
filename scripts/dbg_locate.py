import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
n = int(12.5e9) // 317 * 317
for flags in (1, 0):
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    check(lib.bsk_synth_device(0, 42, flags, 0, C.c_void_p(t.data_ptr()), n, 0, None))
    torch.cuda.synchronize()
    for name, opts, fn in [("locate both", {"Pattern": ["ACGTTGCAAGCT"]}, lib.bsk_locate_run),
                           ("locate +only", {"Pattern": ["ACGTTGCAAGCT"], "OnlyPositiveStrand": True}, lib.bsk_locate_run),
                           ("locate 4-mer", {"Pattern": ["ACGT"], "OnlyPositiveStrand": True}, lib.bsk_locate_run),
                           ("grep both", {"Pattern": ["ACGTTGCAAGCT"], "BySeq": True}, lib.bsk_grep_run),
                           ("grep +only", {"Pattern": ["ACGTTGCAAGCT"], "BySeq": True, "OnlyPositiveStrand": True}, lib.bsk_grep_run)]:
        out = _lib.Out()
        with bsk.Operator("Locate" if "locate" in name else "Grep", json.dumps(opts), 0) as op:
            for _ in range(2):
                check(fn(op.ctx, C.c_void_p(t.data_ptr()), n, 1, 1, 0, None, C.byref(out)), op.ctx)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                check(fn(op.ctx, C.c_void_p(t.data_ptr()), n, 1, 1, 0, None, C.byref(out)), op.ctx)
            torch.cuda.synchronize()
            print("motif=%d %-14s %.2f ms  out=%d" % (flags, name, (time.perf_counter() - t0) / 3 * 1e3, out.len))
