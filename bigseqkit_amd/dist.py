"""Multi-GPU host logic: one process per GPU, `torch.distributed` (backend "nccl" == RCCL on
ROCm; "gloo" in the CPU tests).  The path shards by records -- the reference's only strategy
(PlainFile partitions, /root/reference/bigseqkit/helper.go:148-178) -- so the only
collectives are the two small reductions the reference performs with IgnisHPC Reduce:
    StatsReduce      (bigseqkit/stats.go:91)   -> all_reduce(sum) of the dense stats vector
    GrepReduceCount  (bigseqkit/grep.go:175)   -> all_reduce(sum) of one int64
plus the exchange of rmdup (GroupByKey) and the two prefix sums that MapWithIndex / FaidxOffset imply:
    Range / Head     (bigseqkit/range.go:69-103) -> all_gather of the record counts
    Faidx            (bigseqkit/faidx.go:69-80)  -> all_gather of the shard sizes
"""
import os
import ctypes as C

from ._lib import lib, check


def shard_bounds(data, world, fmt):
    """Cut host-resident file text into `world` record-aligned byte ranges
    (PlainFileN + ReadFixer: every shard begins on a record).  Returns [(lo, hi)] * world."""
    n = len(data)
    # the text is read in place (a file of 100 GB is not copied to be cut): bytes through c_char_p, writable buffers
    # (bytearray, mmap, numpy) through from_buffer; only a read-only view that is not `bytes` costs a copy
    if n == 0:
        keep = C.create_string_buffer(1)
        ptr = C.cast(keep, C.c_void_p)
    elif isinstance(data, bytes):
        keep = C.c_char_p(data)
        ptr = C.cast(keep, C.c_void_p)
    else:
        try:
            keep = (C.c_char * n).from_buffer(data)
            ptr = C.cast(keep, C.c_void_p)
        except (TypeError, ValueError):
            # a read-only buffer (an mmap opened ACCESS_READ): its address through numpy, as run.cut_points does -- no copy
            import numpy as np
            keep = np.frombuffer(data, dtype=np.uint8)
            ptr = C.c_void_p(keep.ctypes.data)
    cuts = [0]
    for k in range(1, world):
        out = C.c_size_t()
        check(lib.bsk_find_record_start(ptr, n, n * k // world, fmt, C.byref(out)))
        cuts.append(max(cuts[-1], out.value))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _active(group=None):
    """a process group whose collectives are to be run: more than one rank -- or ONE rank when
    BSK_DIST_SINGLE_RANK_COLLECTIVES=1 asks for it (bench.py BSK_BENCH_DIST_SINGLE: how a one-GPU box executes the RCCL calls
    of the N-rank path at all)"""
    import torch.distributed as dist
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("BSK_DIST_SINGLE_RANK_COLLECTIVES") == "1")


def coll_device(device, group=None):
    """The device a collective's tensors must live on: the rank's GPU under RCCL ("nccl"), the host under gloo (CPU
    tests; two ranks sharing one GPU in bench.py's functional check)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_backend(group) == "gloo":
        return torch.device("cpu")
    return torch.device(device)


# ---- the only places where this module hands tensors to torch.distributed -------------------------------------------
# RCCL moves device memory and nothing else: a host tensor that reaches an "nccl" collective is an error that shows up
# as a hang or a crash on the first multi-GPU node.  Every collective below goes through _checked(), which refuses
# (ValueError) any tensor that does not live where the backend needs it -- the GPU under nccl, the host under gloo --
# BEFORE the call, so that the mistake is a test failure on one box, not a dead rank on eight.
def _checked(tensors, group=None):
    import torch.distributed as dist
    backend = dist.get_backend(group)
    for t in tensors:
        on_host = t.device.type == "cpu"
        if backend == "nccl" and on_host:
            raise ValueError("bigseqkit_amd.dist: a host tensor was handed to an RCCL collective (backend nccl); "
                             "build it on coll_device(<rank's GPU>)")
        if backend == "gloo" and not on_host:
            raise ValueError("bigseqkit_amd.dist: a %s tensor was handed to a gloo collective; stage it through "
                             "coll_device()" % t.device.type)
    return tensors


def _all_reduce(t, op=None, group=None):
    import torch.distributed as dist
    _checked([t], group)
    dist.all_reduce(t, op=op if op is not None else dist.ReduceOp.SUM, group=group)
    return t


def _all_gather(parts, t, group=None):
    import torch.distributed as dist
    _checked(list(parts) + [t], group)
    dist.all_gather(parts, t, group=group)
    return parts


def barrier(group=None):
    import torch.distributed as dist
    if _active(group):
        dist.barrier(group=group)


def all_reduce_stats_vector(vec, group=None):
    """StatsReduce across ranks: ONE sum all-reduce of the stats vector (int64 tensor on the
    rank's device).  512 KB at hist_cap = 65536: latency-bound, not bandwidth-bound."""
    import torch.distributed as dist
    if _active(group):
        cd = coll_device(vec.device, group)
        if cd != vec.device:  # gloo: through the host
            tmp = vec.to(cd)
            _all_reduce(tmp, group=group)
            vec.copy_(tmp)
        else:
            _all_reduce(vec, group=group)
    return vec


def exchange_stats_overflow(op, vec, group=None, total=None):
    """After the all-reduce, slot [5] of the stats vector is the number of sequence lengths >= hist_cap over ALL ranks,
    but the lengths themselves sit in per-context lists (include/bsk.h).  Every rank hands its list to every other
    (one all_gather of the counts, one of the padded lists) so that bsk_stats_collect on any rank sees all of them --
    without this a chromosome that another rank parsed would vanish from num_seqs / sum_len / N50.
    `total` = slot [5] as every rank already knows it (bsk_stats_overflow_total after a collect: no device round trip);
    None reads it from the device (one synchronising copy)."""
    import torch
    import torch.distributed as dist
    if not _active(group):
        return 0
    if total is None:
        total = int(vec[5].item())
    if int(total) == 0:              # nothing over the dense histogram anywhere (short reads): no exchange
        return 0
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = C.c_size_t()
    check(lib.bsk_stats_overflow_get(op.ctx, None, 0, C.byref(n)), op.ctx)
    mine = (C.c_uint64 * max(1, n.value))()
    check(lib.bsk_stats_overflow_get(op.ctx, mine, n.value, C.byref(n)), op.ctx)
    cdev = coll_device(vec.device, group)
    counts, _ = _all_gather_int(n.value, vec.device, group)
    width = max(counts)
    pad = torch.zeros(max(1, width), dtype=torch.int64, device=cdev)
    if n.value:
        pad[:n.value] = torch.tensor([int(x) for x in mine[:n.value]], dtype=torch.int64, device=cdev)
    parts = [torch.zeros_like(pad) for _ in range(world)]
    _all_gather(parts, pad, group)
    added = 0
    for r in range(world):
        if r == rank or counts[r] == 0:
            continue
        vals = [int(x) for x in parts[r][:counts[r]].tolist()]
        arr = (C.c_uint64 * len(vals))(*vals)
        check(lib.bsk_stats_overflow_add(op.ctx, arr, len(vals)), op.ctx)
        added += len(vals)
    return added


def collect_reduced(op, vec, group=None, reduce=True):
    """StatsReduce + the driver's collect for one step, WITHOUT a device round trip of its own: the sum all-reduce of the
    stats vector, then bsk_stats_collect (the one synchronising copy of the step).  The collect records slot [5] -- the
    number of lengths >= hist_cap over all ranks, the same on every rank -- and only when it is non-zero do the ranks
    exchange their overflow lists and collect again (chromosome-sized records; never for reads).  Returns the map."""
    import torch.distributed as dist
    from . import _lib
    from .api import _collect_map
    if reduce:  # (False: the caller has issued all_reduce_stats_vector itself, e.g. between timing events)
        all_reduce_stats_vector(vec, group)
    multi = _active(group)
    dvec = C.c_void_p(vec.data_ptr())
    try:
        m = _collect_map(op, dvec)
        if not multi:
            return m
    except _lib.BskError as e:
        if not (multi and e.code == _lib.BSK_ERR_OVERFLOW_EXCHANGE):
            raise
        m = None
    total = C.c_uint64()
    check(lib.bsk_stats_overflow_total(op.ctx, C.byref(total)), op.ctx)
    if total.value == 0:
        return m
    exchange_stats_overflow(op, vec, group, total=total.value)   # (every rank: slot [5] is the reduced value)
    return _collect_map(op, dvec)


def all_reduce_count(count, device="cpu", group=None):
    """GrepReduceCount across ranks."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(count)], dtype=torch.int64, device=coll_device(device, group))
    if _active(group):
        _all_reduce(t, group=group)
    return int(t.item())


def all_reduce_max_float(value, device, group=None):
    """max over ranks of one float (bench.py: the step time is the slowest rank's)"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=coll_device(device, group))
    if _active(group):
        _all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def all_gather_floats(values, device, group=None):
    """[[values of rank 0], ..., [values of rank world-1]] (bench.py: per-rank kernel times, barrier skew)"""
    import torch
    import torch.distributed as dist
    if not _active(group):
        return [[float(v) for v in values]]
    cd = coll_device(device, group)
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=cd)
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    _all_gather(parts, mine, group)
    return [[float(x) for x in p.tolist()] for p in parts]


class DeviceText:
    """What an operator left in its context's output buffer (`bsk_out`): `len` bytes of record text in HBM, owned by the
    context until its next run.  At BASELINE sizes a rank's survivors are 20 GB: they stay where they are (a writer drains
    them with bsk_store_put, another operator reads them in place); `bytes()` is the host copy the tests compare."""

    def __init__(self, op, out, device):
        self.op, self.ptr, self.len, self.records, self.device = op, int(out.d_data or 0), int(out.len), int(out.records), device
        self.out = out   # (with the switch "out" = "slices" the text may still be a list of slices: include/bsk.h bsk_out.d_seg_*)

    def __len__(self):
        return self.len

    def tensor(self):
        """torch uint8 view of the bytes (no copy; valid until the context's next run).  A result that is still a list of
        slices is made one block first (bsk_out_materialize)."""
        import torch
        if self.len == 0:
            return torch.empty(0, dtype=torch.uint8, device=self.device)
        if self.out.n_segments:
            check(lib.bsk_out_materialize(self.op.ctx, C.byref(self.out), None), self.op.ctx)
            self.ptr = int(self.out.d_data or 0)

        class _Arr:  # __cuda_array_interface__ works for HIP pointers in torch-rocm
            pass
        a = _Arr()
        a.__cuda_array_interface__ = {"shape": (self.len,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}
        return torch.as_tensor(a, device=self.device)

    def __bytes__(self):
        buf = C.create_string_buffer(max(1, self.len))
        check(lib.bsk_out_to_host(self.op.ctx, C.byref(self.out), buf, self.len), self.op.ctx)
        return buf.raw[:self.len]


# ---------------------------------------------------------------------------
# rmdup across ranks: duplicates are global, so this is the one command with a real exchange step.
# The reference shuffles whole records (GroupByKey, bigseqkit/rmdup.go:97); here 24-byte tuples travel to
# owner = key % world and one keep byte per tuple travels back (include/bsk.h, "rmdup across ranks").
# ---------------------------------------------------------------------------
def _device_view(ptr, shape, typestr, device):
    """torch view of context-owned device memory (no copy; valid until the context's next phase)"""
    import torch

    class _Arr:  # __cuda_array_interface__ works for HIP pointers in torch-rocm
        pass
    a = _Arr()
    a.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(a, device=device)


class HipRmDupBackend:
    """The device phases of libbsk for one rank's HBM-resident shard (a torch uint8 CUDA tensor)."""

    def __init__(self, opts_json, device=0):
        from .api import Operator
        self.op = Operator("RmDup", opts_json, device)
        self.device = device
        self.n = 0

    def close(self):
        self.op.close()

    def keys(self, shard, fmt):
        n = C.c_uint64()
        self._keep = shard
        check(lib.bsk_rmdup_dist_keys(self.op.ctx, C.c_void_p(shard.data_ptr()), shard.numel(), fmt, None, C.byref(n)),
              self.op.ctx)
        self.n = n.value
        return self.n

    def pack(self, base, world):
        import torch
        send = torch.empty((self.n, 3), dtype=torch.int64, device=self._keep.device)
        counts = (C.c_uint64 * world)()
        check(lib.bsk_rmdup_dist_pack(self.op.ctx, base, world, C.c_void_p(send.data_ptr()), counts, None), self.op.ctx)
        return send, [int(x) for x in counts]

    def resolve(self, tuples):
        import torch
        keep = torch.empty(tuples.shape[0], dtype=torch.uint8, device=tuples.device)
        check(lib.bsk_rmdup_dist_resolve(self.op.ctx, C.c_void_p(tuples.data_ptr()), tuples.shape[0],
                                         C.c_void_p(keep.data_ptr()), None), self.op.ctx)
        return keep

    def resolve_ex(self, tuples):
        """resolve + the global index of every tuple's survivor (bsk_rmdup_dist_resolve_ex)"""
        import torch
        keep = torch.empty(tuples.shape[0], dtype=torch.uint8, device=tuples.device)
        surv = torch.empty(tuples.shape[0], dtype=torch.int64, device=tuples.device)
        check(lib.bsk_rmdup_dist_resolve_ex(self.op.ctx, C.c_void_p(tuples.data_ptr()), tuples.shape[0], C.c_void_p(keep.data_ptr()),
                                            C.c_void_p(surv.data_ptr()), None), self.op.ctx)
        return keep, surv

    # ---- round 6: RmDupCheck's text comparison for the duplicates whose survivor lives on another rank (include/bsk.h)
    def xpack(self, send, reply, surv_reply, base, rank_base):
        """-> (requests [m, 3] int64, text uint8, requests per destination, text bytes per destination); the tensors are
        views of the context's send buffers"""
        import torch
        world = len(rank_base) - 1
        rb = (C.c_uint64 * (world + 1))(*[int(x) for x in rank_base])
        rc_, bc_ = (C.c_uint64 * world)(), (C.c_uint64 * world)()
        d_req, d_text = C.c_void_p(), C.c_void_p()
        check(lib.bsk_rmdup_dist_xpack(self.op.ctx, C.c_void_p(send.data_ptr()), C.c_void_p(reply.data_ptr()), C.c_void_p(surv_reply.data_ptr()),
                                       base, rb, world, rc_, bc_, C.byref(d_req), C.byref(d_text), None), self.op.ctx)
        req_cnt, byte_cnt = [int(x) for x in rc_], [int(x) for x in bc_]
        dev = self._keep.device
        m, nb = sum(req_cnt), sum(byte_cnt)
        req = _device_view(d_req.value, (m, 3), "<i8", dev) if m else torch.empty((0, 3), dtype=torch.int64, device=dev)
        text = _device_view(d_text.value, (nb,), "|u1", dev) if nb else torch.empty(0, dtype=torch.uint8, device=dev)
        return req, text, req_cnt, byte_cnt

    def xcompare(self, req_in, req_from, text_in, bytes_from):
        import torch
        world = len(req_from)
        verdict = torch.empty(req_in.shape[0], dtype=torch.uint8, device=req_in.device)
        check(lib.bsk_rmdup_dist_xcompare(self.op.ctx, C.c_void_p(req_in.data_ptr()), (C.c_uint64 * world)(*req_from), C.c_void_p(text_in.data_ptr()),
                                          (C.c_uint64 * world)(*bytes_from), world, C.c_void_p(verdict.data_ptr()), None), self.op.ctx)
        return verdict

    def xapply(self, verdict_back):
        """-> number of records of this shard whose text differs from their survivor's; self.pairs_compared"""
        nf, pairs = C.c_uint64(), C.c_uint64()
        check(lib.bsk_rmdup_dist_xapply(self.op.ctx, C.c_void_p(verdict_back.data_ptr()), C.byref(nf), C.byref(pairs), None), self.op.ctx)
        self.pairs_compared = pairs.value
        return nf.value

    def flagged(self):
        need = C.c_size_t()
        check(lib.bsk_rmdup_dist_flagged_get(self.op.ctx, None, 0, C.byref(need)), self.op.ctx)
        buf = C.create_string_buffer(max(1, need.value))
        check(lib.bsk_rmdup_dist_flagged_get(self.op.ctx, buf, need.value, C.byref(need)), self.op.ctx)
        return buf.raw[:need.value]

    def settle(self, all_lists):
        check(lib.bsk_rmdup_dist_flagged_settle(self.op.ctx, all_lists, len(all_lists)), self.op.ctx)

    def pair_stats(self):
        """(pairs compared inside the shard, pairs whose survivor lives on another rank, flagged records) of the last exchange"""
        a, b, f = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib.bsk_rmdup_dist_stats(self.op.ctx, C.byref(a), C.byref(b), C.byref(f)), self.op.ctx)
        return a.value, b.value, f.value

    def emit(self, send, reply, base, to_host=True, surv_reply=None):
        """survivors of this rank's shard: host bytes (tests), or with to_host=False a DeviceText -- the text stays in the
        context's output buffer in HBM (20 GB per rank at C5 do not belong on the host).  surv_reply (the survivors' global
        indices, routed back like the keep bytes): without the cross-rank check (xpack .. xapply) before it, the duplicates
        whose survivor lives in this shard are byte-compared with it here (bsk_rmdup_dist_emit_ex); self.local_pairs = how
        many pairs inside the shard were compared"""
        from . import _lib
        out = _lib.Out()
        if surv_reply is None:
            check(lib.bsk_rmdup_dist_emit(self.op.ctx, C.c_void_p(send.data_ptr()), C.c_void_p(reply.data_ptr()), base, None,
                                          C.byref(out)), self.op.ctx)
        else:
            n = C.c_uint64()
            check(lib.bsk_rmdup_dist_emit_ex(self.op.ctx, C.c_void_p(send.data_ptr()), C.c_void_p(reply.data_ptr()),
                                             C.c_void_p(surv_reply.data_ptr()), base, None, C.byref(out), C.byref(n)), self.op.ctx)
            self.local_pairs = n.value
        text = DeviceText(self.op, out, self._keep.device)
        return bytes(text) if to_host else text


class _Phases:
    """wall clock per phase of rmdup_distributed, each one closed by a device synchronisation (only when the caller asks
    for them: bench.py's breakdown call; the synchronisations are not part of the product path)"""

    def __init__(self, sink, device):
        self.sink, self.device, self.t = sink, device, None

    def mark(self, name=None):
        if self.sink is None:
            return
        import time
        import torch
        if getattr(self.device, "type", "cpu") == "cuda":
            torch.cuda.synchronize(self.device)
        now = time.perf_counter()
        if name is not None:
            self.sink[name] = self.sink.get(name, 0.0) + (now - self.t) * 1e3
        self.t = now


def _xcheck(backend, send, reply, surv_reply, base, counts_all, world, rank, dev, group, multi, ph, phases):
    """Round 6: RmDupCheck's text comparison (bigseqkit-lib/rmdup.go:193-211) for EVERY duplicate.  The duplicates whose
    survivor lives on another rank send their subject there (24-byte requests + the text, two all_to_alls), the survivor's
    rank answers one byte per request (one all_to_all back); the pairs inside a shard are compared where they are.  Records
    whose text differs from their survivor's (two subjects under one pair of keys; the tests mask the keys) are handed to
    every rank -- only when there are any -- and regrouped by text."""
    import torch
    rank_base = [0]
    for c in counts_all:
        rank_base.append(rank_base[-1] + int(c))
    req, text, req_cnt, byte_cnt = backend.xpack(send, reply, surv_reply, base, rank_base)
    ph.mark("xpack")
    if phases is not None:
        phases["xcheck_requests"] = int(sum(req_cnt))
        phases["xcheck_text_bytes_sent"] = int(sum(byte_cnt))
    if multi:
        t_in = torch.tensor([v for p in range(world) for v in (req_cnt[p], byte_cnt[p])], dtype=torch.int64, device=dev)
        t_out = torch.empty(2 * world, dtype=torch.int64, device=dev)
        _all_to_all_single(t_out, t_in, None, None, group)
        got = [int(x) for x in t_out.tolist()]
        req_from, bytes_from = got[0::2], got[1::2]
        req_in = torch.empty((sum(req_from), 3), dtype=torch.int64, device=dev)
        _all_to_all_single(req_in, req, req_from, req_cnt, group)
        text_in = torch.empty(sum(bytes_from), dtype=torch.uint8, device=dev)
        _all_to_all_single(text_in, text, bytes_from, byte_cnt, group)
        ph.mark("xchange")
        verdict = backend.xcompare(req_in, req_from, text_in, bytes_from)
        ph.mark("xcompare")
        vback = torch.empty(sum(req_cnt), dtype=torch.uint8, device=dev)
        _all_to_all_single(vback, verdict, req_cnt, req_from, group)
        ph.mark("xreply")
    else:
        vback = torch.empty(0, dtype=torch.uint8, device=dev)
    n_flagged = backend.xapply(vback)
    blob = backend.flagged() if n_flagged else b""
    if multi:
        sizes, _ = _all_gather_int(len(blob), dev, group)
    else:
        sizes = [len(blob)]
    if sum(sizes):
        if multi:
            cd = coll_device(dev, group)
            width = max(sizes)
            pad = torch.zeros(width, dtype=torch.uint8, device=cd)
            if blob:
                pad[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(cd)
            parts = [torch.zeros_like(pad) for _ in range(world)]
            _all_gather(parts, pad, group)
            blob = b"".join(bytes(parts[r][:sizes[r]].cpu().numpy().tobytes()) for r in range(world))
        backend.settle(blob)
    ph.mark("xapply")
    if phases is not None:
        phases["xcheck_flagged"] = int(sum(sizes) != 0)


def rmdup_distributed(shard, fmt, backend, group=None, to_host=True, phases=None):
    """RmDup over the shards of all ranks of `group`; returns the survivors of THIS rank's shard (file order), so that the
    concatenation over ranks equals the single-GPU output: host bytes, or with to_host=False whatever the backend's emit
    leaves on the device (HipRmDupBackend: a DeviceText).  Collectives: one all_gather of the record counts, one
    all_to_all of split sizes, one all_to_all of tuples, one all_to_all of keep bytes and one of survivor indices, then the
    text comparison of the duplicates whose survivor lives on another rank (_xcheck; BSK_RMDUP_XCHECK=off leaves them to
    their two keys, as round 5 did).
    `phases` (a dict) receives milliseconds per phase -- keys / pack / all_to_all / resolve / reply / x* / emit -- and
    `tuple_bytes_sent` / `tuple_bytes_sent_off_rank` / `xcheck_*`."""
    import torch
    import torch.distributed as dist
    # (BSK_DIST_SINGLE_RANK_COLLECTIVES=1: a process group of ONE rank still takes the exchange -- every collective of the
    # N-rank path then runs over the backend, which is how a one-GPU box executes the RCCL calls at all: bench.py
    # BSK_BENCH_DIST_SINGLE)
    multi = _active(group)
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    dev = shard.device
    ph = _Phases(phases, dev)
    ph.mark()
    n = backend.keys(shard, fmt)
    ph.mark("keys")
    if multi:
        counts_all, _ = _all_gather_int(n, dev, group)
        base = int(sum(counts_all[:rank]))
    else:
        base = 0
    send, in_splits = backend.pack(base, world)
    ph.mark("pack")
    if phases is not None:
        phases["tuple_bytes_sent"] = 24 * int(sum(in_splits))
        phases["tuple_bytes_sent_off_rank"] = 24 * int(sum(c for r, c in enumerate(in_splits) if r != rank))
    ex = hasattr(backend, "resolve_ex")   # (round 5: the owner also names the survivor, local pairs are byte-compared)
    xc = ex and hasattr(backend, "xpack") and os.environ.get("BSK_RMDUP_XCHECK") != "off"
    if not multi:
        keep, surv = backend.resolve_ex(send) if ex else (backend.resolve(send), None)
        ph.mark("resolve")
        if xc:
            _xcheck(backend, send, keep, surv, base, [n], 1, 0, dev, group, False, ph, phases)
        kw = {"surv_reply": surv} if ex else {}
        out = backend.emit(send, keep, base, **kw) if to_host else backend.emit(send, keep, base, to_host=False, **kw)
        ph.mark("emit")
        return out
    t_in = torch.tensor(in_splits, dtype=torch.int64, device=dev)
    t_out = torch.empty(world, dtype=torch.int64, device=dev)
    _all_to_all_single(t_out, t_in, None, None, group)
    out_splits = [int(x) for x in t_out.tolist()]
    recv = torch.empty((sum(out_splits), 3), dtype=torch.int64, device=dev)
    _all_to_all_single(recv, send, out_splits, in_splits, group)
    ph.mark("all_to_all")
    keep, surv = backend.resolve_ex(recv) if ex else (backend.resolve(recv), None)
    ph.mark("resolve")
    reply = torch.empty(n, dtype=torch.uint8, device=dev)
    _all_to_all_single(reply, keep, in_splits, out_splits, group)
    kw = {}
    if ex:
        surv_reply = torch.empty(n, dtype=torch.int64, device=dev)
        _all_to_all_single(surv_reply, surv, in_splits, out_splits, group)
        kw = {"surv_reply": surv_reply}
    ph.mark("reply")
    if xc:
        _xcheck(backend, send, reply, surv_reply, base, counts_all, world, rank, dev, group, True, ph, phases)
    out = backend.emit(send, reply, base, **kw) if to_host else backend.emit(send, reply, base, to_host=False, **kw)
    ph.mark("emit")
    return out


# No single message of an all-to-all may exceed this many bytes.  Measured on this image (RCCL 2.26.6, torch 2.10, one rank
# sending to itself: scripts/history/r04_a2a_probe2.py): all_to_all_single delivers a message of up to 1 GiB whole and of a
# larger one only the FIRST HALF -- silently.  A C5 rank sends 1.9 GB of tuples; with 8 ranks a message is 236 MB, with 2
# ranks 946 MB, with one 1.9 GB.  Larger exchanges go in rounds of at most this size per message.
A2A_MAX_BYTES = 512 << 20


def _all_to_all_single(out, inp, out_splits, in_splits, group=None):
    """dist.all_to_all_single on the rank's device under RCCL; staged through the host under gloo (CPU tests, ranks that
    share a GPU).  With split lists (rows of dim 0 per peer) the exchange runs in as many rounds as the largest message
    anywhere needs to stay under A2A_MAX_BYTES (BSK_A2A_MAX_BYTES overrides: the tests force the rounds)."""
    import torch
    import torch.distributed as dist
    cd = coll_device(out.device, group)
    if cd != out.device:
        o = torch.empty(out.shape, dtype=out.dtype, device=cd)
        i = inp.to(cd)
        _all_to_all_single(o, i, out_splits, in_splits, group)
        out.copy_(o)
        return
    _checked([out, inp], group)
    if out_splits is None:  # equal split (the few words of the split sizes themselves)
        dist.all_to_all_single(out, inp, None, None, group=group)
        return
    # (the row size comes from the SHAPE: a rank whose shard holds no record sends a (0, 3) tensor, and inp[0] of that
    # raises before the all-reduce below -- the other ranks would wait in it for ever)
    row_elems = 1
    for d in tuple(inp.shape[1:]):
        row_elems *= int(d)
    row_bytes = max(1, inp.element_size() * row_elems)
    limit = max(1, int(os.environ.get("BSK_A2A_MAX_BYTES", A2A_MAX_BYTES)) // row_bytes)  # rows per message
    biggest = torch.tensor([max(list(out_splits) + list(in_splits) + [0])], dtype=torch.int64, device=cd)
    _all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)  # (every rank must take the same number of rounds)
    rounds = max(1, -(-int(biggest.item()) // limit))
    if rounds == 1:
        dist.all_to_all_single(out, inp, list(out_splits), list(in_splits), group=group)
        return
    def offsets(splits):
        acc, r = 0, []
        for c in splits:
            r.append(acc)
            acc += c
        return r
    o_off, i_off = offsets(out_splits), offsets(in_splits)
    for r in range(rounds):
        piece = lambda c: max(0, min(limit, c - r * limit))
        o_cnt, i_cnt = [piece(c) for c in out_splits], [piece(c) for c in in_splits]
        send = torch.cat([inp[i_off[p] + r * limit: i_off[p] + r * limit + i_cnt[p]] for p in range(len(in_splits))])
        recv = torch.empty((sum(o_cnt),) + tuple(out.shape[1:]), dtype=out.dtype, device=cd)
        dist.all_to_all_single(recv, send, o_cnt, i_cnt, group=group)
        at = 0
        for p in range(len(out_splits)):
            out[o_off[p] + r * limit: o_off[p] + r * limit + o_cnt[p]].copy_(recv[at: at + o_cnt[p]])
            at += o_cnt[p]


def _all_gather_int(value, device, group=None):
    """[value of rank 0, ..., value of rank world-1] (a list of one element without a process group)"""
    import torch
    import torch.distributed as dist
    if not _active(group):
        return [int(value)], 0
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    device = coll_device(device, group)
    parts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    _all_gather(parts, torch.tensor([int(value)], dtype=torch.int64, device=device), group)
    # ONE device -> host copy for all counts (a .item() per part would be `world` synchronising copies under RCCL)
    return [int(x) for x in torch.cat(parts).tolist()], rank


class HipRangeBackend:
    """Range / Head for one rank's HBM-resident shard: count (record table), then the run with the global index base."""

    def __init__(self, op_name, opts_json, device=0):
        from .api import Operator
        self.op = Operator(op_name, opts_json, device)

    def close(self):
        self.op.close()

    def count(self, shard, fmt):
        n = C.c_uint64()
        self._shard, self._fmt = shard, fmt
        check(lib.bsk_index_build(self.op.ctx, C.c_void_p(shard.data_ptr()), shard.numel(), 1, fmt, None, C.byref(n)), self.op.ctx)
        return n.value

    def run(self, first_record, total, to_host=True):
        from . import _lib
        needs = C.c_int()
        check(lib.bsk_range_needs_count(self.op.ctx, C.byref(needs)), self.op.ctx)
        if needs.value:
            check(lib.bsk_range_set_count(self.op.ctx, total), self.op.ctx)
        out = _lib.Out()
        s = self._shard
        check(lib.bsk_range_run(self.op.ctx, C.c_void_p(s.data_ptr()), s.numel(), 1, self._fmt, 0, first_record, None,
                                C.byref(out)), self.op.ctx)
        text = DeviceText(self.op, out, s.device)
        return bytes(text) if to_host else text


def range_distributed(shard, fmt, backend, group=None, to_host=True):
    """Range / Head over the shards of all ranks: the record index of MapWithIndex is global, so every rank learns the
    number of records before its shard (and the total, for negative positions) from ONE all_gather of the counts.
    Returns this rank's selected records (host bytes, or with to_host=False what the backend leaves on the device); the
    concatenation over ranks equals the single-GPU output."""
    counts, rank = _all_gather_int(backend.count(shard, fmt), shard.device, group)
    if to_host:
        return backend.run(sum(counts[:rank]), sum(counts))
    return backend.run(sum(counts[:rank]), sum(counts), to_host=False)


def faidx_distributed(shard, fmt, run, group=None):
    """faidx index rows over the shards of all ranks: the FaidxOffset pass (bigseqkit/faidx.go:69-80) is ONE all_gather
    of the shard sizes; `run(base_offset)` produces this rank's rows (HIP: bsk_faidx_run)."""
    sizes, rank = _all_gather_int(shard.numel(), shard.device, group)
    return run(sum(sizes[:rank]))


def store_fastx(path, payload, group=None, device=None):
    """StoreFASTX (`--merge`, one output file) across ranks.  The reference's FileStore passes an MPI token from
    executor to executor so that partitions append in order (bigseqkit-lib/helper.go:378-460); here every rank learns
    its byte offset from one all_gather of the payload sizes and writes its part with a single pwrite -- all ranks
    write concurrently, the file equals the single-GPU output (rank order == file order).  The size tensors live on
    `device` (default: the current GPU under the nccl backend, the host under gloo)."""
    import os
    import torch
    import torch.distributed as dist
    multi = _active(group)
    if not multi:
        with open(path, "wb") as f:
            f.write(payload)
        return len(payload)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if device is None:  # RCCL ("nccl") only moves device tensors; gloo takes host tensors
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    sizes, _ = _all_gather_int(len(payload), device, group)
    total, off = sum(sizes), sum(sizes[:rank])
    if rank == 0:
        with open(path, "wb") as f:
            f.truncate(total)
    barrier(group)
    fd = os.open(path, os.O_WRONLY)
    try:
        done, view = 0, memoryview(payload)
        while done < len(view):
            done += os.pwrite(fd, view[done:], off + done)
    finally:
        os.close(fd)
    barrier(group)
    return total
