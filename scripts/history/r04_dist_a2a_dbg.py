#!/usr/bin/env python3
"""the tuple exchange of the multi-GPU rmdup with ONE rank over RCCL: is what all_to_all_single delivers what was sent?"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
import torch
import torch.distributed as dist
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib, dist as bdist
from bigseqkit_amd._lib import lib, check

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
n = int(gb * 1e9) // 317 * 317
t = torch.empty(n, dtype=torch.uint8, device=dev)
check(lib.bsk_synth_device(0, 42, _lib.SYNTH_FLAG_DUPS, 0, C.c_void_p(t.data_ptr()), n, 0, None))
torch.cuda.synchronize()
b = bdist.HipRmDupBackend(json.dumps({"BySeq": True}), 0)
for it in range(4):
    nrec = b.keys(t, bsk.FORMAT_FASTQ)
    send, counts = b.pack(0, 1)
    recv = torch.empty((nrec, 3), dtype=torch.int64, device=dev)
    sync_before = it >= 2
    if sync_before:
        torch.cuda.synchronize()
    dist.all_to_all_single(recv, send, [nrec], [nrec])
    keep = b.resolve(recv)
    reply = torch.empty(nrec, dtype=torch.uint8, device=dev)
    dist.all_to_all_single(reply, keep, [nrec], [nrec])
    out = b.emit(send, reply, 0, to_host=False)
    torch.cuda.synchronize()
    print("iter", it, "synced" if sync_before else "", "recv==send", bool((recv == send).all()), "reply==keep", bool((reply == keep).all()),
          "kept", int(keep.sum()), "reply kept", int(reply.sum()), "records out", out.records, "expected", nrec - nrec // 5, flush=True)
b.close()
dist.destroy_process_group()
