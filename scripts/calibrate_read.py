#!/usr/bin/env python3
"""Streaming-read ceiling + FETCH_SIZE calibration for k_stats' access pattern.
Run plain for the timing, or under `rocprofv3 --pmc FETCH_SIZE` for the counter."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bigseqkit_amd._lib import lib, check
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
n = int(gb * 1e9) // 4096 * 4096
t = torch.empty(n, dtype=torch.uint8, device="cuda")
check(lib.bsk_synth_device(0, 42, 0, 0, C.c_void_p(t.data_ptr()), n, 0, None))
torch.cuda.synchronize()
out = {"bytes": n}
for bpc in (2, 4, 5, 6, 8):
    ms = C.c_float()
    check(lib.bsk_selftest_stream_read(C.c_void_p(t.data_ptr()), n, 5, bpc, C.byref(ms)))
    out[f"blocks_per_cu_{bpc}"] = {"ms": round(ms.value, 3), "GBps": round(n / ms.value / 1e6, 1)}
print(json.dumps(out))
