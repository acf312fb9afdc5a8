#!/usr/bin/env python3
"""What a plain device-to-device copy moves on this box (read + write bytes per second): torch's copy kernel and hipMemcpyDtoD
through libbsk's bsk_device_copy -- the number k_seg_copy / the slice gathers are compared with."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from bigseqkit_amd._lib import lib
n = 20_000_000_000
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.random_(0, 255)
for name, fn in (("torch copy_", lambda: b.copy_(a)), ("hipMemcpy D2D", lambda: lib.bsk_device_copy(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, 3))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%-14s %.2f ms for %.0f GB in + out = %.2f TB/s" % (name, dt * 1e3, 2 * n / 1e9, 2 * n / dt / 1e12))
a64 = a.view(torch.int64); b64 = b.view(torch.int64)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): b64.copy_(a64)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("torch copy_ i64 %.2f ms = %.2f TB/s" % (dt * 1e3, 2 * n / dt / 1e12))
# ... and a plain READ of the same bytes: torch's sum over int64 / int32 views (a vendor reduction kernel)
for name, v in (("torch sum i64", a64), ("torch sum i32", a.view(torch.int32))):
    v.sum(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): s = v.sum()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%-14s %.2f ms for %.0f GB = %.2f TB/s" % (name, dt * 1e3, n / 1e9, n / dt / 1e12))
