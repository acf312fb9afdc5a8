#!/usr/bin/env python3
"""Device phases of the multi-GPU rmdup on ONE GPU (the collectives in between are not timed here): one C5 shard
(25 GB FASTQ-150, 20 % duplicates) packed for a world of 8, then the owner phase over the tuples this rank would keep.
Usage: bench_rmdup_dist.py [GB] [world]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib, dist as bdist
from bigseqkit_amd._lib import lib, check

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(gb * 1e9) // 317 * 317
t = torch.empty(n, dtype=torch.uint8, device="cuda")
check(lib.bsk_synth_device(0, 42, _lib.SYNTH_FLAG_DUPS, 0, C.c_void_p(t.data_ptr()), n, 0, None))
torch.cuda.synchronize()
b = bdist.HipRmDupBackend(json.dumps({"BySeq": True}), 0)
res = {}
def timed(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    res[name + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 2)
    return r
nrec = timed("keys", lambda: b.keys(t, bsk.FORMAT_FASTQ))
send, counts = timed("pack", lambda: b.pack(0, world))
own = send[:counts[0]].contiguous()                      # bucket of owner 0 from this rank
recv = own.repeat(world, 1) if world > 1 else send       # stand-in for the world buckets an owner receives
keep = timed("resolve_owner", lambda: b.resolve(recv))
reply = torch.ones(nrec, dtype=torch.uint8, device="cuda")
out = timed("emit", lambda: b.emit(send, reply, 0), reps=1)
res.update(records=nrec, tuple_bytes_per_rank=int(nrec * 24), bucket_counts=counts[:8], survivors_bytes=len(out))
print(json.dumps(res))
b.close()
