#!/usr/bin/env python3
"""all_to_all_single over RCCL with one rank: sizes, dtypes, with / without a device synchronisation behind it"""
import os, sys
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29545")
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
for rows in (1000, 10**6, 3 * 10**7, 78864353):
    for dtype, cols in ((torch.int64, 3), (torch.uint8, 1)):
        send = torch.randint(0, 127, (rows, cols), device=dev).to(dtype)
        for mode in ("plain", "sync_after", "async_wait"):
            recv = torch.zeros_like(send)
            torch.cuda.synchronize()
            if mode == "async_wait":
                w = dist.all_to_all_single(recv, send, [rows], [rows], async_op=True)
                w.wait()
            else:
                dist.all_to_all_single(recv, send, [rows], [rows])
            if mode == "sync_after":
                torch.cuda.synchronize()
            ok_now = bool((recv == send).all())
            torch.cuda.synchronize()
            ok_later = bool((recv == send).all())
            print(rows, str(dtype), mode, "equal at once:", ok_now, "after a device sync:", ok_later, flush=True)
# the equal-split form and all_reduce / all_gather for comparison
x = torch.arange(10**7, device=dev)
y = torch.zeros_like(x)
dist.all_to_all_single(y, x)
print("equal split", bool((x == y).all()))
z = x.clone(); dist.all_reduce(z); print("all_reduce", bool((x == z).all()))
dist.destroy_process_group()
