"""The N>1 path on CPU: world_size 2 over gloo.  Record-aligned shard cutting and the two
reductions of the hot path (StatsReduce as one all-reduce of the dense stats vector,
GrepReduceCount as one int64).  The per-shard maps come from the oracle here (no GPU in this
container); on the GPU box the same vectors come out of k_stats (tests -m gpu, bench.py)."""
import ctypes as C
import json
import os
import random
import socket

import pytest

import oracle
import seqgen
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib, dist as bdist
from bigseqkit_amd._lib import lib


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, data, fastq, opts, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
        lo, hi = bdist.shard_bounds(data, world, fmt)[rank]
        shard = data[lo:hi]
        op = bsk.Operator("Stats", json.dumps(opts), -1)
        vlen = lib.bsk_stats_vector_len(op.ctx)
        # dense stats vector of this rank's shard (layout of include/bsk.h)
        m = oracle.stats_map(shard, fastq, json.dumps(opts))
        vec = torch.zeros(vlen, dtype=torch.int64)
        nrec = 0
        for k, v in m.items():
            if k >= 0:
                vec[_lib.STATS_HDR + k] = v
                nrec += v
        vec[0], vec[1], vec[2], vec[3] = m.get(-1, 0), m.get(-2, 0), m.get(-3, 0), nrec
        bdist.all_reduce_stats_vector(vec)
        # grep -C on the same shards
        gopts = {"BySeq": True, "Pattern": ["ACG"], "Count": True}
        cnt = int(oracle.grep(shard, fastq, json.dumps(gopts))) if len(shard) else 0
        total = bdist.all_reduce_count(cnt)
        if rank == 0:
            first = oracle.record_spans(data, fastq)
            fr = data[first[0][0]:first[0][0] + first[0][1]] if first else b""
            h = (C.c_uint64 * vlen)(*[int(x) for x in vec.tolist()])
            keys, vals, n = (C.c_int64 * 4096)(), (C.c_int64 * 4096)(), C.c_size_t()
            rc = lib.bsk_stats_collect_host(op.ctx, h, vlen, fr, len(fr), fmt, keys, vals, 4096, C.byref(n))
            assert rc == 0, lib.bsk_last_error(op.ctx)
            got = dict(zip(keys[:n.value], vals[:n.value]))
            info = _lib.StatInfo()
            assert lib.bsk_stats_finalize(op.ctx, keys, vals, n.value, C.byref(info)) == 0
            out = C.create_string_buffer(4096)
            assert lib.bsk_stats_string(op.ctx, b"input0", b"N/A", C.byref(info), out, 4096) == 0
            q.put((got, out.value.decode(), total, (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fastq", [True, False])
def test_two_ranks_gloo_stats_and_grep_count(fastq):
    import torch.multiprocessing as mp
    rng = random.Random(21 + fastq)
    data = seqgen.random_fastq(rng, 500, 0, 120) if fastq else seqgen.random_fasta(rng, 300, 0, 400, gt_in_header=True)
    opts = {"All": True, "Tabular": True}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, data, fastq, opts, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, text, total, (lo, hi) = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert 0 < hi < len(data)  # a real cut
    assert got == oracle.stats_map(data, fastq, json.dumps(opts))
    assert text == oracle.stats_string(data, fastq, json.dumps(opts))
    assert total == int(oracle.grep(data, fastq, json.dumps({"BySeq": True, "Pattern": ["ACG"], "Count": True})))


def test_shard_bounds_are_record_aligned_and_cover_the_file():
    rng = random.Random(5)
    for fastq in (True, False):
        data = seqgen.random_fastq(rng, 300, 0, 90) if fastq else seqgen.random_fasta(rng, 200, 0, 300)
        starts = {s for s, _ in oracle.record_spans(data, fastq)} | {len(data)}
        for world in (1, 2, 3, 8):
            b = bdist.shard_bounds(data, world, int(fastq))
            assert b[0][0] == 0 and b[-1][1] == len(data)
            assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert all(lo in starts for lo, _ in b)
            # per-shard results reduce to the whole (StatsReduce sums, PARITY.md Q2)
            whole = oracle.stats_map(data, fastq, '{"All": true}')
            acc = {}
            for lo, hi in b:
                for k, v in oracle.stats_map(data[lo:hi], fastq, '{"All": true}').items():
                    if k != -4:
                        acc[k] = acc.get(k, 0) + v
            acc[-4] = whole[-4]
            if -3 not in whole:
                acc.pop(-3, None)
            assert {k: v for k, v in acc.items() if v or k in whole} == whole


# ---------------------------------------------------------------- rmdup: the one command with an exchange step
def _rmdup_worker(rank, world, port, data, opts, q, store=None, key_bits=64):
    import torch
    import torch.distributed as dist
    from rmdup_cpu_backend import OracleRmDupBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = bdist.shard_bounds(data, world, bsk.FORMAT_FASTQ)[rank]
        shard = torch.frombuffer(bytearray(data[lo:hi]), dtype=torch.uint8) if hi > lo else torch.empty(0, dtype=torch.uint8)
        out = bdist.rmdup_distributed(shard, bsk.FORMAT_FASTQ, OracleRmDupBackend(opts, key_bits))
        # the resident form (bench.py's N > 1 leg: survivors stay with the backend) and the per-phase clock
        phases = {}
        res = bdist.rmdup_distributed(shard, bsk.FORMAT_FASTQ, OracleRmDupBackend(opts, key_bits), to_host=False, phases=phases)
        assert bytes(res) == out and res.len == len(out) and res.records == out.count(b"\n") // 4
        assert set(phases) >= {"keys", "pack", "all_to_all", "resolve", "reply", "emit", "tuple_bytes_sent", "xpack", "xapply", "xcheck_requests"}
        assert phases["xcheck_flagged"] == (1 if key_bits < 64 else 0)   # (masked keys: different subjects under one pair -- settled by text)
        assert phases["tuple_bytes_sent"] == 24 * (shard.numpy().tobytes().count(b"\n") // 4)
        assert 0 <= phases["tuple_bytes_sent_off_rank"] <= phases["tuple_bytes_sent"]
        if store:
            bdist.store_fastx(store, out)   # scan of sizes + one pwrite per rank (FileStore's ordered single-file merge)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("key_bits", [64, 5])
@pytest.mark.parametrize("a2a_max_bytes", [None, 240])
@pytest.mark.parametrize("opts", [{"BySeq": True}, {"BySeq": True, "IgnoreCase": True}, {}])
def test_two_ranks_gloo_rmdup_exchange(opts, a2a_max_bytes, key_bits, tmp_path, monkeypatch):
    """all_gather(counts) + all_to_all(tuples) + all_to_all(keep bytes): the concatenated per-rank survivors equal the
    single-shard result, i.e. the first occurrence in FILE order survives even when it lives on the other rank.
    Round 6: the subject text of every duplicate whose survivor lives on the other rank travels there and is compared
    (dist._xcheck); key_bits = 5 keeps five bits of either key, so that DIFFERENT subjects share a pair of keys on both
    ranks -- the comparison flags them and the settlement by text lets every first-of-its-text survive, as the oracle does.
    a2a_max_bytes = 240: ten tuples per message, the exchange runs in rounds (dist._all_to_all_single: RCCL on this image
    delivers only the first half of a message beyond 1 GiB, so large exchanges are cut)."""
    import torch.multiprocessing as mp
    if a2a_max_bytes:
        monkeypatch.setenv("BSK_A2A_MAX_BYTES", str(a2a_max_bytes))  # (the spawned ranks inherit the environment)
    rng = random.Random(77)
    seqs, recs = [], []
    for i in range(600):
        s = seqs[rng.randrange(len(seqs))] if (i > 5 and rng.random() < 0.35) else "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 60)))
        if rng.random() < 0.2:
            s = s.lower()
        seqs.append(s)
        recs.append(f"@r{i % 400} c\n{s}\n+\n{'I' * len(s)}\n")
    data = "".join(recs).encode()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    merged = str(tmp_path / "merged.fq")
    procs = [ctx.Process(target=_rmdup_worker, args=(r, 2, port, data, opts, q, merged, key_bits)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = oracle.rmdup(data, True, json.dumps(opts))
    assert outs[0] + outs[1] == want
    assert 0 < len(want) < len(data)
    assert open(merged, "rb").read() == want


@pytest.mark.parametrize("a2a_max_bytes", [None, 240])
def test_two_ranks_gloo_rmdup_with_an_empty_rank(a2a_max_bytes, monkeypatch):
    """A file with fewer records than ranks: one rank's shard is EMPTY, it sends a (0, 3) tuple tensor.  ADVICE r04: the
    chunked all-to-all took its row size from inp[0] and raised IndexError on that rank before the all-reduce of the round
    count -- the other rank then waited in the collective until the launcher killed it."""
    import torch.multiprocessing as mp
    if a2a_max_bytes:
        monkeypatch.setenv("BSK_A2A_MAX_BYTES", str(a2a_max_bytes))
    data = b"@only one\nACGTACGT\n+\nIIIIIIII\n"
    assert [hi - lo for lo, hi in bdist.shard_bounds(data, 2, bsk.FORMAT_FASTQ)].count(0) == 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rmdup_worker, args=(r, 2, port, data, {"BySeq": True}, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[0] + outs[1] == data


def test_shard_bounds_reads_a_read_only_mapping_in_place(tmp_path):
    """a read-only mmap fell back to from_buffer_copy -- the file it says is 'not copied to be cut' was copied
    in full.  The address now comes through numpy, as run.cut_points takes it."""
    import mmap
    rng = random.Random(5)
    data = "".join(f"@r{i}\n{''.join(rng.choice('ACGT') for _ in range(40))}\n+\n{'I' * 40}\n" for i in range(300)).encode()
    f = tmp_path / "x.fq"
    f.write_bytes(data)
    with open(f, "rb") as fh:
        m = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            assert bdist.shard_bounds(m, 3, bsk.FORMAT_FASTQ) == bdist.shard_bounds(data, 3, bsk.FORMAT_FASTQ)
        finally:
            m.close()


# ---------------------------------------------------------------- range / head / faidx: one all_gather each
class _CpuRangeBackend:
    """TEST stand-in for HipRangeBackend: libbsk's host-side range arithmetic (bsk_create on device -1,
    bsk_range_set_count, bsk_range_bounds) + a record split by the oracle's rule."""

    def __init__(self, op_name, opts_json):
        self.op = bsk.Operator(op_name, opts_json, -1)

    def count(self, shard, fmt):
        data = bytes(shard.numpy().tobytes())
        b = bdist.shard_bounds(data, 1, fmt)  # (exercises the anchor search; one shard)
        assert b == [(0, len(data))]
        # records of this shard: cut at every record start the library finds
        self.records, pos = [], 0
        import ctypes as C
        arr = (C.c_char * max(1, len(data))).from_buffer_copy(data if data else b"\0")
        while pos < len(data):
            out = C.c_size_t()
            bsk.lib.bsk_find_record_start(C.cast(arr, C.c_void_p), len(data), pos + 1, fmt, C.byref(out))
            nxt = min(out.value, len(data)) if out.value > pos else len(data)
            self.records.append(data[pos:nxt])
            pos = nxt
        return len(self.records)

    def run(self, first_record, total):
        import ctypes as C
        needs = C.c_int()
        assert bsk.lib.bsk_range_needs_count(self.op.ctx, C.byref(needs)) == 0
        if needs.value:
            assert bsk.lib.bsk_range_set_count(self.op.ctx, total) == 0
        lo, hi = C.c_int64(), C.c_int64()
        assert bsk.lib.bsk_range_bounds(self.op.ctx, C.byref(lo), C.byref(hi)) == 0
        keep = [r for k, r in enumerate(self.records) if lo.value <= first_record + k < hi.value]
        return b"".join(r if r.endswith(b"\n") else r + b"\n" for r in keep)


def _range_worker(rank, world, port, data, fmt, op_name, opts, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = bdist.shard_bounds(data, world, fmt)[rank]
        shard = torch.frombuffer(bytearray(data[lo:hi]), dtype=torch.uint8) if hi > lo else torch.empty(0, dtype=torch.uint8)
        got = bdist.range_distributed(shard, fmt, _CpuRangeBackend(op_name, json.dumps(opts)))
        # faidx: rows of this shard from the oracle, shifted by the base offset the collective delivers
        def rows(base):
            out = []
            for line in oracle.faidx(data[lo:hi], fmt == bsk.FORMAT_FASTQ).decode().splitlines():
                f = line.split("\t")
                f[2] = str(int(f[2]) + base)
                if len(f) > 5:
                    f[5] = str(int(f[5]) + base)
                out.append("\t".join(f))
            return ("\n".join(out) + "\n").encode() if out else b""
        fai = bdist.faidx_distributed(shard, fmt, rows)
        q.put((rank, got, fai))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fastq", [True, False])
@pytest.mark.parametrize("op_name,opts", [("Range", {"Range": "120:480"}), ("Range", {"Range": "-90:-1"}), ("Head", {"N": 333}),
                                          ("Range", {"Range": "2:5"})])
def test_two_ranks_gloo_range_head_and_faidx(fastq, op_name, opts):
    """the record index is global (all_gather of the counts), the .fai offsets are file offsets (all_gather of the
    shard sizes): the concatenated per-rank results equal the single-shard oracle result"""
    import torch.multiprocessing as mp
    rng = random.Random(5)
    data = seqgen.random_fastq(rng, 600, 1, 80) if fastq else seqgen.random_fasta(rng, 600, 1, 200, width=60)
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_range_worker, args=(r, 2, port, data, fmt, op_name, opts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, got, fai = q.get(timeout=120)
        res[rank] = (got, fai)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = oracle.head(data, fastq, json.dumps(opts)) if op_name == "Head" else oracle.range_(data, fastq, json.dumps(opts))
    assert res[0][0] + res[1][0] == want and len(want) > 0
    assert res[0][1] + res[1][1] == oracle.faidx(data, fastq)


def test_bench_self_launches_the_ranks_it_is_asked_for():
    """`python bench.py --gpus 2` (no torchrun, no WORLD_SIZE) must start TWO ranks that reduce together -- the
    launcher path the driver's N>1 runs take when they call bench.py directly.  Here with gloo and the launch check
    only (no GPU in this container); the timed path asserts the same world size before it touches a device."""
    import subprocess
    import sys
    env = dict(os.environ, BSK_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    assert got == {"launch_check": True, "n_gpus": 2, "backend": "gloo", "ranks_reduced": 2}


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", BSK_BENCH_BACKEND="gloo")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_without_a_gpu_fails_loudly():
    """No CPU fallback: on a box without a HIP device the bench exits non-zero instead of printing a line."""
    import subprocess
    import sys
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is visible")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gb", "0.01", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "no HIP device" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_global_operators_see_several_shards_as_one():
    """api._one_shard: RmDup / Rename / Sort / grep --delete-matched are global over the dataframe (GroupByKey,
    bigseqkit/rmdup.go:97; SortByKey, sort.go) -- the shards of ReadFASTQN are joined, a missing final newline added."""
    from bigseqkit_amd import api
    a, b, c = b"@r1\nAC\n+\nII\n", b"@r2\nGT\n+\nII", b"@r1\nAC\n+\n##\n"
    f = api.SeqFrame(bsk.FORMAT_FASTQ, [a, b, c])
    one = api._one_shard(f)
    assert len(one.shards) == 1 and bytes(one.shards[0]) == a + b + b"\n" + c
    assert api._one_shard(api.SeqFrame(bsk.FORMAT_FASTQ, [a])).shards == [a]
    # and the joined text is what the oracle dedups globally: r1 appears twice, once per shard
    want = oracle.rmdup(bytes(one.shards[0]), True, json.dumps({"BySeq": True}))
    assert want.count(b"@r1") == 1


# ---------------------------------------------------------------- the RCCL ("nccl") branches, as far as a box without GPUs can hold them
class _FakeDist:
    """torch.distributed with a chosen backend name and collectives that only RECORD what they were handed: what the
    nccl branches of dist.py would give RCCL on a multi-GPU node (that code had never run)."""

    def __init__(self, backend, world=2, rank=0):
        import torch.distributed as real
        self.ReduceOp = real.ReduceOp
        self.backend, self.world, self.rank = backend, world, rank
        self.calls = []

    def is_initialized(self):
        return True

    def get_backend(self, group=None):
        return self.backend

    def get_world_size(self, group=None):
        return self.world

    def get_rank(self, group=None):
        return self.rank

    def all_reduce(self, t, op=None, group=None):
        self.calls.append(("all_reduce", [t.device.type]))

    def all_gather(self, parts, t, group=None):
        self.calls.append(("all_gather", [p.device.type for p in parts] + [t.device.type]))
        for p in parts:
            p.copy_(t)

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None, group=None):
        self.calls.append(("all_to_all_single", [out.device.type, inp.device.type]))

    def barrier(self, group=None):
        self.calls.append(("barrier", []))


def test_no_host_tensor_reaches_an_rccl_collective(monkeypatch):
    """Under backend "nccl" every tensor dist.py hands to a collective must live on the rank's GPU.  Without a GPU here:
    (1) with host tensors every helper must REFUSE before the collective is called (ValueError, nothing recorded);
    (2) with tensors on a non-host device (torch's "meta" device stands in for the GPU) the helpers that need no values
    back reach the collective, and everything they hand over is on that device."""
    import torch
    import torch.distributed as real
    fake = _FakeDist("nccl")
    for name in ("is_initialized", "get_backend", "get_world_size", "get_rank", "all_reduce", "all_gather",
                 "all_to_all_single", "barrier"):
        monkeypatch.setattr(real, name, getattr(fake, name))
    host = torch.zeros(16, dtype=torch.int64)
    with pytest.raises(ValueError, match="host tensor"):
        bdist.all_reduce_stats_vector(host)
    with pytest.raises(ValueError, match="host tensor"):
        bdist.all_reduce_count(3, "cpu")
    with pytest.raises(ValueError, match="host tensor"):
        bdist.all_reduce_max_float(1.0, "cpu")
    with pytest.raises(ValueError, match="host tensor"):
        bdist.all_gather_floats([1.0, 2.0], "cpu")
    with pytest.raises(ValueError, match="host tensor"):
        bdist._all_gather_int(5, torch.device("cpu"))
    with pytest.raises(ValueError, match="host tensor"):
        bdist._all_to_all_single(torch.zeros(4, dtype=torch.int64), torch.zeros(4, dtype=torch.int64), None, None)
    assert fake.calls == []   # nothing got through
    # (2) device-resident tensors pass, and stay on the device all the way into the collective
    meta = torch.device("meta")
    bdist.all_reduce_stats_vector(torch.zeros(16, dtype=torch.int64, device=meta))
    bdist._all_to_all_single(torch.zeros(4, dtype=torch.int64, device=meta), torch.zeros(4, dtype=torch.int64, device=meta), None, None)
    bdist.barrier()
    assert [c[0] for c in fake.calls] == ["all_reduce", "all_to_all_single", "barrier"]
    assert all(d == "meta" for _, devs in fake.calls for d in devs)
    assert bdist.coll_device("cuda:3") == torch.device("cuda", 3)


def test_gloo_collectives_refuse_device_tensors_and_stage_through_the_host(monkeypatch):
    import torch
    import torch.distributed as real
    fake = _FakeDist("gloo")
    for name in ("is_initialized", "get_backend", "get_world_size", "get_rank", "all_reduce", "all_gather",
                 "all_to_all_single", "barrier"):
        monkeypatch.setattr(real, name, getattr(fake, name))
    assert bdist.coll_device("cuda:0") == torch.device("cpu")
    with pytest.raises(ValueError, match="gloo"):
        bdist._checked([torch.zeros(1, device="meta")])
    assert bdist.all_reduce_count(7, "cuda:0") == 7          # built on the host although the rank's device is a GPU
    counts, rank = bdist._all_gather_int(5, torch.device("cuda", 0))
    assert counts == [5, 5] and rank == 0
    assert all(d == "cpu" for _, devs in fake.calls for d in devs)


def test_overflow_exchange_is_decided_without_a_device_round_trip(monkeypatch):
    """collect_reduced: the all-reduce, ONE collect, and an exchange of the overflow lists only when the collect saw a
    non-zero slot [5] -- decided from bsk_stats_overflow_total (host state of the context), never from vec[5].item()."""
    import torch
    import torch.distributed as real
    fake = _FakeDist("gloo")
    for name in ("is_initialized", "get_backend", "get_world_size", "get_rank", "all_reduce", "all_gather",
                 "all_to_all_single", "barrier"):
        monkeypatch.setattr(real, name, getattr(fake, name))
    from bigseqkit_amd import api
    events = []

    class Vec:  # a stats vector whose VALUES must not be looked at by the protocol
        device = torch.device("cpu")

        def data_ptr(self):
            return 0

        def to(self, d):
            return torch.zeros(8, dtype=torch.int64)

        def copy_(self, t):
            return self

        def __getitem__(self, i):
            raise AssertionError("collect_reduced read the vector on its own (a synchronising device copy per step)")

    class Op:
        ctx = None

    state = {"total": 0, "collects": 0}

    def fake_collect(op, d_vec=None):
        state["collects"] += 1
        events.append("collect")
        if state["total"] and state["collects"] == 1:
            raise _lib.BskError(_lib.BSK_ERR_OVERFLOW_EXCHANGE, "exchange the overflow lists")
        return {150: 7}

    monkeypatch.setattr(api, "_collect_map", fake_collect)
    monkeypatch.setattr(lib, "bsk_stats_overflow_total", lambda ctx, p: (setattr(p._obj, "value", state["total"]), 0)[1])
    monkeypatch.setattr(bdist, "exchange_stats_overflow", lambda op, vec, group=None, total=None: events.append(("exchange", total)))
    assert bdist.collect_reduced(Op(), Vec()) == {150: 7}
    assert events == ["collect"]                       # reads: one collect, no exchange
    events.clear()
    state.update(total=3, collects=0)
    assert bdist.collect_reduced(Op(), Vec()) == {150: 7}
    assert events == ["collect", ("exchange", 3), "collect"]
