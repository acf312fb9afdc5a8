"""Multi-GPU host logic: one process per GPU, `torch.distributed` (backend "nccl" == RCCL on
ROCm; "gloo" in the CPU tests).  The path shards by records -- the reference's only strategy
(PlainFile partitions, /root/reference/bigseqkit/helper.go:148-178) -- so the only
collectives are the two small reductions the reference performs with IgnisHPC Reduce:
    StatsReduce      (bigseqkit/stats.go:91)   -> all_reduce(sum) of the dense stats vector
    GrepReduceCount  (bigseqkit/grep.go:175)   -> all_reduce(sum) of one int64
"""
import ctypes as C

from ._lib import lib, check


def shard_bounds(data, world, fmt):
    """Cut host-resident file text into `world` record-aligned byte ranges
    (PlainFileN + ReadFixer: every shard begins on a record).  Returns [(lo, hi)] * world."""
    n = len(data)
    arr = (C.c_char * max(1, n)).from_buffer_copy(data if n else b"\0")
    cuts = [0]
    for k in range(1, world):
        out = C.c_size_t()
        check(lib.bsk_find_record_start(C.cast(arr, C.c_void_p), n, n * k // world, fmt, C.byref(out)))
        cuts.append(max(cuts[-1], out.value))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def all_reduce_stats_vector(vec):
    """StatsReduce across ranks: ONE sum all-reduce of the stats vector (int64 tensor on the
    rank's device).  512 KB at hist_cap = 65536: latency-bound, not bandwidth-bound."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return vec


def all_reduce_count(count, device="cpu"):
    """GrepReduceCount across ranks."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(count)], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
