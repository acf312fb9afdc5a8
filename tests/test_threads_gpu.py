"""The threading contract of the boundary (include/bsk.h "THREADS"): the reference runs Call() of one operator struct from
Threads() goroutines at once (/root/reference/bigseqkit-lib/helper.go:413-416, rmdup.go:100,224); here every caller thread
owns a context, a FileStore is shared, and a second call on a busy context is refused, not raced."""
import ctypes as C
import json
import random
import threading

import pytest

import oracle
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

pytestmark = pytest.mark.gpu


def fastq(rng, nrec, tag):
    recs = []
    for i in range(nrec):
        L = rng.randint(30, 200)
        s = "".join(rng.choice("ACGT") for _ in range(L))
        if i % 7 == 3 and i > 10:
            s = recs[rng.randrange(len(recs))][1]                     # duplicates for rmdup
        recs.append(("@%s%d x" % (tag, i), s))
    return "".join("%s\n%s\n+\n%s\n" % (h, s, "I" * len(s)) for h, s in recs).encode()


JOBS = [("SeqTransform", lib.bsk_seq_run, {"Reverse": True, "Complement": True, "Config": {"SeqType": "dna", "Quiet": True}}, oracle.seq),
        ("Grep", lib.bsk_grep_run, {"BySeq": True, "Pattern": ["ACGTA", "TTGCA"]}, oracle.grep),
        ("SubseqTransform", lib.bsk_subseq_run, {"Region": "3:-3"}, oracle.subseq),
        ("RmDup", lib.bsk_rmdup_run, {"BySeq": True}, lambda d, fq, o: oracle.rmdup(d, fq, o)),
        ("SeqTransform", lib.bsk_seq_run, {"Name": True}, oracle.seq),
        ("Translate", lib.bsk_translate_run, {"Frame": ["6"], "AllowUnknownCodon": True}, oracle.translate),
        ("Grep", lib.bsk_grep_run, {"BySeq": True, "Pattern": ["GATTACA"], "InvertMatch": True}, oracle.grep),
        ("SeqTransform", lib.bsk_seq_run, {"MinLen": 100, "Config": {"Quiet": True}}, oracle.seq)]


def test_eight_threads_eight_contexts_one_device_one_store(tmp_path):
    """8 host threads x their own context x one device, mixed operators, several shards each, every output put into ONE
    merged FileStore as its own part: the file == the oracle's outputs in part order, whatever the threads' timing"""
    import torch
    rng = random.Random(8)
    rounds = 3
    shards = [[fastq(rng, 1500 + 40 * (t + r), "t%dr%d_" % (t, r)) for r in range(rounds)] for t in range(len(JOBS))]
    want = {}
    for t, (name, fn, opts, orc) in enumerate(JOBS):
        for r in range(rounds):
            want[t * rounds + r] = orc(shards[t][r], True, json.dumps(opts))
    dev = [[torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda() for s in row] for row in shards]
    torch.cuda.synchronize()
    path = str(tmp_path / "merged.out")
    st = C.c_void_p()
    assert lib.bsk_store_open(path.encode(), 1, C.byref(st)) == 0
    errors = []
    start = threading.Barrier(len(JOBS))

    def worker(t):
        name, fn, opts, _ = JOBS[t]
        try:
            with bsk.Operator(name, json.dumps(opts), 0) as op:
                start.wait()
                for r in range(rounds):
                    out = _lib.Out()
                    d = dev[t][r]
                    check(fn(op.ctx, C.c_void_p(d.data_ptr()), d.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out)), op.ctx)
                    check(lib.bsk_store_put(st, op.ctx, t * rounds + r, C.byref(out)), op.ctx)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(JOBS))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    tot = C.c_uint64()
    assert lib.bsk_store_close(st, C.byref(tot)) == 0
    got = open(path, "rb").read()
    assert got == b"".join(want[k] for k in sorted(want))
    assert tot.value == len(got)


def test_second_call_on_a_busy_context_is_refused():
    """one context, two threads: while a long run is under way every other call on that context comes back with
    BSK_ERR_INVALID_ARG and the text "context busy" -- and the long run's result is untouched"""
    import torch
    rng = random.Random(3)
    big = fastq(rng, 120000, "b")
    small = fastq(rng, 50, "s")
    want = oracle.rmdup(big, True, json.dumps({"BySeq": True}))
    d_big = torch.frombuffer(bytearray(big), dtype=torch.uint8).cuda()
    d_small = torch.frombuffer(bytearray(small), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    with bsk.Operator("RmDup", json.dumps({"BySeq": True}), 0) as op:
        out_big = _lib.Out()
        done = threading.Event()
        res = {}

        def long_run():
            good = 0
            while good < 6:      # (several runs back to back: the window the other thread has to hit; it may win a slot too)
                rc = lib.bsk_rmdup_run(op.ctx, C.c_void_p(d_big.data_ptr()), d_big.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out_big))
                if rc == 0:
                    good += 1
                elif rc != _lib.BSK_ERR_INVALID_ARG:
                    res["rc"] = rc
                    break
            done.set()

        th = threading.Thread(target=long_run)
        th.start()
        busy, ok = 0, 0
        while not done.is_set():
            out = _lib.Out()
            rc = lib.bsk_rmdup_run(op.ctx, C.c_void_p(d_small.data_ptr()), d_small.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out))
            if rc == _lib.BSK_ERR_INVALID_ARG:
                assert b"context busy" in lib.bsk_global_error()   # (the refused caller's thread-local text: the context's own belongs to the running call)
                busy += 1
            else:
                assert rc == 0
                ok += 1
        th.join()
        assert "rc" not in res, res
        assert busy > 0, "the second thread never met the running call (ok=%d)" % ok
        # the context still works, and a run without company gives the oracle's answer
        out = _lib.Out()
        check(lib.bsk_rmdup_run(op.ctx, C.c_void_p(d_big.data_ptr()), d_big.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out)), op.ctx)
        buf = C.create_string_buffer(max(1, out.len))
        check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
        assert buf.raw[:out.len] == want


def test_stats_contexts_of_several_threads_add_up():
    """Stats from 4 threads, a context each, partial maps merged as StatsReduce does (sum): == the oracle on the whole"""
    import torch
    rng = random.Random(5)
    parts = [fastq(rng, 3000, "p%d_" % k) for k in range(4)]
    whole = b"".join(parts)
    want = oracle.stats_map(whole, True, '{"All": true}')
    maps, errors = [None] * 4, []

    def worker(k):
        try:
            t = torch.frombuffer(bytearray(parts[k]), dtype=torch.uint8).cuda()
            o = bsk.SeqKitStatsOptions().All(True)
            with bsk.Operator("Stats", o.to_json(), 0) as op:
                check(lib.bsk_stats_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTQ, k, None, None), op.ctx)
                maps[k] = bsk.api._collect_map(op, None)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=120)
    assert not errors, errors
    acc = {}
    for m in maps:
        for key, v in m.items():
            if key != -4:
                acc[key] = acc.get(key, 0) + v
    acc[-4] = want[-4]
    assert {k: v for k, v in acc.items() if v or k in want} == want
