"""Worker of tests/test_multirank_gpu.py: one of N ranks (python -m torch.distributed.run), every rank with a
record-aligned shard of the same text on the GPU it is given (ranks share GPUs when there are fewer than ranks; the
collectives then run over gloo through the host, bigseqkit_amd/dist.py coll_device).  Writes what the rank produced to
<outdir>/<name>.<rank>."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C

import torch
import torch.distributed as dist

import bigseqkit_amd as bsk
from bigseqkit_amd import _lib, dist as bdist
from bigseqkit_amd._lib import lib, check


def main():
    outdir, data_path, fmt = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    ngpu = torch.cuda.device_count()
    backend = "nccl" if ngpu >= world else "gloo"
    devi = local % ngpu
    torch.cuda.set_device(devi)
    dev = torch.device("cuda", devi)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=dev)
    else:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    data = open(data_path, "rb").read()
    lo, hi = bdist.shard_bounds(data, world, fmt)[rank]
    shard = torch.frombuffer(bytearray(data[lo:hi]), dtype=torch.uint8).to(dev) if hi > lo else torch.empty(0, dtype=torch.uint8, device=dev)

    def put(name, payload):
        with open(os.path.join(outdir, "%s.%d" % (name, rank)), "wb") as f:
            f.write(payload)

    # stats (+ -a): one all-reduce of the stats vector
    o = bsk.SeqKitStatsOptions().Tabular(True).All(True)
    with bsk.Operator("Stats", o.to_json(), devi) as op:
        vlen = lib.bsk_stats_vector_len(op.ctx)
        vec = torch.zeros(vlen, dtype=torch.int64, device=dev)
        if shard.numel():
            check(lib.bsk_stats_run(op.ctx, C.c_void_p(shard.data_ptr()), shard.numel(), 1, fmt, rank, C.c_void_p(vec.data_ptr()), None), op.ctx)
        torch.cuda.synchronize()
        m = bdist.collect_reduced(op, vec)   # all-reduce + collect (+ the overflow exchange when the reduced vector asks)
        put("stats", json.dumps(sorted(m.items())).encode())
    # grep -C: one all-reduce of a count
    g = {"Pattern": ["ACG"], "BySeq": True, "Count": True}
    with bsk.Operator("Grep", json.dumps(g), devi) as op:
        out = _lib.Out()
        cnt = C.c_uint64(0)
        if shard.numel():
            check(lib.bsk_grep_run(op.ctx, C.c_void_p(shard.data_ptr()), shard.numel(), 1, fmt, rank, None, C.byref(out)), op.ctx)
            check(lib.bsk_grep_last_count(op.ctx, C.byref(cnt)), op.ctx)
        put("grepc", str(bdist.all_reduce_count(cnt.value, dev)).encode())
    # rmdup: tuple exchange
    be = bdist.HipRmDupBackend(json.dumps({"BySeq": True}), devi)
    put("rmdup", bdist.rmdup_distributed(shard, fmt, be))
    put("rmdup_pairs", json.dumps(list(be.pair_stats()) + [be.n]).encode())   # (local pairs, cross-rank pairs, flagged, records)
    be.close()
    # the same with 16 bits of k1 and 2 bits of k2: DIFFERENT sequences share a pair of keys, also across ranks -- the text
    # comparison of round 6 (bsk_rmdup_dist_x*) flags them and the settlement by text lets the first of every text survive
    be = bdist.HipRmDupBackend(json.dumps({"BySeq": True}), devi)
    check(lib.bsk_ctx_set(be.op.ctx, b"rmdup_k1_bits", b"16"), be.op.ctx)
    check(lib.bsk_ctx_set(be.op.ctx, b"rmdup_k2_bits", b"2"), be.op.ctx)
    put("rmdupx", bdist.rmdup_distributed(shard, fmt, be))
    put("rmdupx_pairs", json.dumps(list(be.pair_stats()) + [be.n]).encode())
    be.close()
    # range with negative positions: needs the global count
    rb = bdist.HipRangeBackend("Range", json.dumps({"Range": "3:-3"}), devi)
    put("range", bdist.range_distributed(shard, fmt, rb))
    rb.close()
    # --merge store: every rank writes its part at its offset
    with bsk.Operator("SeqTransform", json.dumps({"Reverse": True}), devi) as op:
        out = _lib.Out()
        payload = b""
        if shard.numel():
            check(lib.bsk_seq_run(op.ctx, C.c_void_p(shard.data_ptr()), shard.numel(), 1, fmt, rank, None, C.byref(out)), op.ctx)
            buf = C.create_string_buffer(max(1, out.len))
            check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
            payload = buf.raw[:out.len]
        bdist.store_fastx(os.path.join(outdir, "merged.fq"), payload)
    put("backend", backend.encode())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
