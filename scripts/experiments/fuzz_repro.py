#!/usr/bin/env python3
"""replay one case of tests/test_fuzz_gpu.py (JSON: opts, data) on the device: output, stages, and the oracle's answer"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
case = json.load(open(sys.argv[1]))
data, opts = case["data"].encode(), case["opts"]
want = oracle.translate(data, False, json.dumps(opts))
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
for sets in ((), ((b"translate_index", b"full"),), ((b"translate_index", b"light"),)):
    with bsk.Operator("Translate", json.dumps(opts), 0) as op:
        for k, v in sets:
            check(lib.bsk_ctx_set(op.ctx, k, v), op.ctx)
        lib.bsk_profile_reset(op.ctx); lib.bsk_profile_enable(op.ctx, 1)
        out = _lib.Out()
        check(lib.bsk_translate_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTA, 0, None, C.byref(out)), op.ctx)
        buf = C.create_string_buffer(max(1, out.len))
        check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
        pb = C.create_string_buffer(4096)
        check(lib.bsk_profile_dump(op.ctx, pb, len(pb)), op.ctx)
        got = buf.raw[:out.len]
        print(sets, "OK" if got == want else "DIFFERENT", pb.value.decode())
        if got != want:
            print("  got ", got.replace(b"\n", b" "))
            print("  want", want.replace(b"\n", b" "))
