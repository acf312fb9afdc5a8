#!/usr/bin/env python3
"""where all_to_all_single over RCCL (one rank, send to self) stops delivering what was sent: bytes per message"""
import os, sys
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29546")
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
for mb in (900, 1000, 1023, 1025, 1100, 1500, 2047, 2049, 3000, 4097):
    nb = mb << 20
    for dtype, w in ((torch.uint8, 1), (torch.int64, 8)):
        n = nb // w
        send = (torch.arange(n, device=dev) % 251).to(dtype)
        recv = torch.zeros_like(send)
        dist.all_to_all_single(recv, send, [n], [n])
        torch.cuda.synchronize()
        bad = (recv != send).nonzero().flatten()
        print(mb, "MiB", str(dtype), "ok" if bad.numel() == 0 else "WRONG from element %d (byte %d), %d wrong, last %d" % (int(bad[0]), int(bad[0]) * w, bad.numel(), int(bad[-1])), flush=True)
        del send, recv, bad
# the point-to-point form, to self
n = (1500 << 20) // 8
send = torch.arange(n, device=dev); recv = torch.zeros_like(send)
ops = [dist.P2POp(dist.isend, send, 0), dist.P2POp(dist.irecv, recv, 0)]
for w in dist.batch_isend_irecv(ops): w.wait()
torch.cuda.synchronize()
print("batch_isend_irecv 1500 MiB to self", bool((send == recv).all()))
dist.destroy_process_group()
