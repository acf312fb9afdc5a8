#!/usr/bin/env python3
"""`translate -f 6` on the synthetic 5 kb CDS records that differ (bench.py's second C4 leg), HBM-resident: time per call and
per stage, for the values of BSK_* given in the environment.  Usage: bench_translate_var.py [GB] [calls]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = int(os.environ.get("KIND", str(_lib.SYNTH_FASTA5K_VAR)))
if kind == _lib.SYNTH_FASTA5K_VAR:
    lo, hi = 1, 1
    while lib.bsk_synth_offset(kind, hi) <= gb * 1e9: hi *= 2
    while lo + 1 < hi:
        mid = (lo + hi) // 2
        if lib.bsk_synth_offset(kind, mid) <= gb * 1e9: lo = mid
        else: hi = mid
    n = lib.bsk_synth_offset(kind, lo)
else:
    rb = lib.bsk_synth_record_bytes(kind); n = int(gb * 1e9) // rb * rb
t = torch.empty(n, dtype=torch.uint8, device="cuda")
check(lib.bsk_synth_device(kind, 42, 0, 0, C.c_void_p(t.data_ptr()), n, 0, None))
torch.cuda.synchronize()
out = _lib.Out()
with bsk.Operator("Translate", json.dumps({"Frame": ["6"]}), 0) as op:
    check(lib.bsk_translate_run(op.ctx, C.c_void_p(t.data_ptr()), n, 1, 0, 0, None, C.byref(out)), op.ctx)
    torch.cuda.synchronize()
    lib.bsk_profile_reset(op.ctx); lib.bsk_profile_enable(op.ctx, 1)
    t0 = time.perf_counter()
    for _ in range(calls):
        check(lib.bsk_translate_run(op.ctx, C.c_void_p(t.data_ptr()), n, 1, 0, 0, None, C.byref(out)), op.ctx)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / calls
    pb = C.create_string_buffer(4096); check(lib.bsk_profile_dump(op.ctx, pb, len(pb)), op.ctx)
print("%.1f GB in, %.1f GB out: %.2f ms per call  (%s)  %s" % (n / 1e9, out.len / 1e9, dt * 1e3,
      " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("BSK_")), pb.value.decode()))
