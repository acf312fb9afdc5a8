"""Round 6: results as ORDERED SLICES.  The reference's Call returns []string whose elements share the
bytes of the partition (/root/reference/bigseqkit-lib/rmdup.go:200-222 appends the strings it was handed; subseq.go:167-225,
seq.go:81-269 build one string per record) -- nothing is moved into one block.  With the switch "out" = "slices" the operators
whose text already sits in HBM in output order return that (include/bsk.h bsk_out.d_seg_*): `rmdup -s` on FASTQ (segments of
the shard), `seq -n [-i]` and `subseq -r` on FASTQ (the per-range buffers of the streaming passes).  Every consumer must see
the same bytes as with the one block: bsk_out_to_host, bsk_store_put (gathered in 32 MiB pieces), bsk_out_materialize."""
import ctypes as C
import json
import os
import random

import pytest

import oracle
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

pytestmark = pytest.mark.gpu

RUN = {"RmDup": lib.bsk_rmdup_run, "SeqTransform": lib.bsk_seq_run, "SubseqTransform": lib.bsk_subseq_run, "Grep": lib.bsk_grep_run}
# (the last four: records an operator KEEPS verbatim are segments of the shard too -- `seq` with its filters, `grep`)
CASES = [("RmDup", {"BySeq": True}, oracle.rmdup), ("RmDup", {"BySeq": True, "IgnoreCase": True}, oracle.rmdup),
         ("SeqTransform", {"Name": True}, oracle.seq), ("SeqTransform", {"Name": True, "OnlyId": True}, oracle.seq),
         ("SubseqTransform", {"Region": "1:50"}, oracle.subseq), ("SubseqTransform", {"Region": "-30:-2"}, oracle.subseq),
         ("SeqTransform", {"MinLen": 100}, oracle.seq), ("SeqTransform", {"MinQual": 19.5, "MaxLen": 150}, oracle.seq),
         ("Grep", {"Pattern": ["ACG"], "BySeq": True, "OnlyPositiveStrand": True}, oracle.grep),
         ("Grep", {"Pattern": ["lane=3"], "ByName": True, "UseRegexp": True, "InvertMatch": True}, oracle.grep)]


def fastq(seed, n, dup=0.3):
    rng = random.Random(seed)
    pool, recs = [], []
    for i in range(n):
        if pool and rng.random() < dup:
            s = rng.choice(pool)
        else:
            s = "".join(rng.choice("ACGTacgtN") for _ in range(rng.choice((150, 150, 151, 36, 75))))
            pool.append(s)
        recs.append("@read%d/1 lane=%d\n%s\n+\n%s\n" % (i, i % 8, s, "".join(chr(rng.randint(35, 73)) for _ in s)))
    return "".join(recs).encode()


def run(name, opts, t, mode):
    op = bsk.Operator(name, json.dumps(opts), 0)
    check(lib.bsk_ctx_set(op.ctx, b"out", mode.encode()), op.ctx)
    out = _lib.Out()
    check(RUN[name](op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out)), op.ctx)
    return op, out


def host(op, out):
    buf = C.create_string_buffer(max(1, out.len))
    check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
    return buf.raw[:out.len]


@pytest.mark.parametrize("case", CASES, ids=["%s-%s" % (c[0], "-".join(c[1])) for c in CASES])
@pytest.mark.parametrize("n", [1, 50, 20000])
def test_slices_hold_the_same_text(case, n, tmp_path):
    import torch
    name, opts, ofn = case
    data = fastq(n, n)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    want = ofn(data, True, json.dumps(opts))
    op_c, out_c = run(name, opts, t, "contiguous")
    assert out_c.n_segments == 0 and host(op_c, out_c) == want
    op, out = run(name, opts, t, "slices")
    assert out.len == len(want) and out.records == out_c.records
    if want:   # (a filter that keeps nothing has nothing to leave in place)
        assert out.n_segments > 0 and not out.d_data and out.d_seg_src and out.d_seg_off     # nothing was gathered
    # 1. to the host, the context keeps the slices ...
    assert host(op, out) == want
    # 2. ... the writer drains them ...
    st = C.c_void_p()
    path = str(tmp_path / "o.fq")
    check(lib.bsk_store_open(path.encode(), 1, C.byref(st)))
    check(lib.bsk_store_put(st, op.ctx, 0, C.byref(out)), op.ctx)
    tot = C.c_uint64()
    check(lib.bsk_store_close(st, C.byref(tot)))
    assert tot.value == len(want) and open(path, "rb").read() == want
    # 3. ... and one block is made when somebody asks for it (the next operator of a pipe)
    check(lib.bsk_out_materialize(op.ctx, C.byref(out), None), op.ctx)
    assert out.n_segments == 0 and (out.d_data or not want) and host(op, out) == want
    got = torch.empty(out.len, dtype=torch.uint8, device="cuda")
    check(lib.bsk_device_copy(C.c_void_p(got.data_ptr()), C.c_void_p(out.d_data), out.len, 3))
    torch.cuda.synchronize()
    assert bytes(got.cpu().numpy().tobytes()) == want
    op.close(); op_c.close()


@pytest.mark.parametrize("name,opts,nrec", [("RmDup", {"BySeq": True}, 1300000), ("SeqTransform", {"Name": True}, 6000000),
                                            ("SubseqTransform", {"Region": "1:140"}, 1300000)])
def test_pieces_of_a_large_result(name, opts, nrec, tmp_path):
    """results beyond the 32 MiB pieces of the drain and the 64 MiB pieces of bsk_out_to_host: 0.4 GB (names: 1.9 GB) of
    synthetic reads with 20 % duplicates; the slices against the one block of the same call"""
    import torch
    n = 317 * nrec
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.bsk_synth_device(0, 11, 2, 0, C.c_void_p(t.data_ptr()), n, 0, None) == 0
    torch.cuda.synchronize()
    op_c, out_c = run(name, opts, t, "contiguous")
    want = host(op_c, out_c)
    op_c.close()
    op, out = run(name, opts, t, "slices")
    assert out.n_segments > 0 and out.len == len(want) > (64 << 20)
    assert host(op, out) == want
    st = C.c_void_p()
    path = str(tmp_path / "big.out")
    check(lib.bsk_store_open(path.encode(), 1, C.byref(st)))
    check(lib.bsk_store_put(st, op.ctx, 0, C.byref(out)), op.ctx)
    check(lib.bsk_store_close(st, None))
    with open(path, "rb") as f:
        assert f.read() == want
    os.unlink(path)
    op.close()


def test_a_result_dies_with_the_next_run_and_exceptions_keep_one_block():
    import torch
    data = fastq(5, 3000)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    op, out = run("RmDup", {"BySeq": True}, t, "slices")
    first = _lib.Out.from_buffer_copy(out)
    out2 = _lib.Out()
    t2 = torch.frombuffer(bytearray(fastq(6, 10)), dtype=torch.uint8).cuda()
    check(lib.bsk_rmdup_run(op.ctx, C.c_void_p(t2.data_ptr()), t2.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out2)), op.ctx)
    buf = C.create_string_buffer(max(1, first.len))
    assert lib.bsk_out_to_host(op.ctx, C.byref(first), buf, first.len) != 0          # refused, not answered with other bytes
    assert b"slices" in lib.bsk_last_error(op.ctx)
    op.close()
    # -d / -D read the scratch arrays again, '+' lines that repeat the name are re-formatted: one block, whatever the switch says
    plus = data.replace(b"\n+\n", b"\n+read\n", 1)
    for d, o in ((data, {"BySeq": True, "DupNumFile": "/tmp/bsk_slices_dupnum"}), (plus, {"BySeq": True})):
        tt = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
        op, out = run("RmDup", o, tt, "slices")
        assert out.n_segments == 0 and host(op, out) == oracle.rmdup(d, True, json.dumps({"BySeq": True}))
        op.close()


@pytest.mark.parametrize("name,opts,ofn", [("SeqTransform", {"MinLen": 40}, oracle.seq), ("Grep", {"Pattern": ["AC"], "BySeq": True}, oracle.grep)])
def test_kept_records_that_are_not_four_plain_lines_take_the_one_block(name, opts, ofn):
    """the records `seq` / `grep` keep are slices of the shard only while they leave exactly as they stand: a '+' line that
    repeats the name is re-written (SeqParser.Read + Format, helper.go:252-311) -- one block then, whatever the switch says;
    bases wrapped over several lines likewise leave as four lines.  The reference's text either way"""
    import torch
    base = fastq(21, 400, dup=0.0)
    plus = base.replace(b"\n+\n", b"\n+read7/1 again\n", 3)
    recs = base.decode().split("\n")
    wrapped = []
    for i in range(0, len(recs) - 1, 4):   # bases and qualities over lines of 50
        h, s, p, q = recs[i:i + 4]
        q = q.replace("@", "A").replace("+", "B")   # (a wrapped quality line must not look like a header or a separator)
        wrapped += [h] + [s[j:j + 50] for j in range(0, len(s), 50)] + [p] + [q[j:j + 50] for j in range(0, len(q), 50)]
    wrapped = ("\n".join(wrapped) + "\n").encode()
    # (wrapped records are first made four-line text in a buffer of the context: what is kept may be slices of THAT)
    for label, data, one_block in (("plus", plus, True), ("wrapped", wrapped, False)):
        want = ofn(data, True, json.dumps(opts))
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        op, out = run(name, opts, t, "slices")
        assert host(op, out) == want and len(want) > 0, label
        assert out.n_segments == 0 or not one_block, label
        op.close()
    # ... and the same call on the plain records leaves slices
    t = torch.frombuffer(bytearray(base), dtype=torch.uint8).cuda()
    op, out = run(name, opts, t, "slices")
    assert out.n_segments > 0 and host(op, out) == ofn(base, True, json.dumps(opts))
    op.close()
