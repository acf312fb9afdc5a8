"""Parity of fq2fa, range / head and duplicate against the CPU oracle, through the C ABI."""
import json
import random

import pytest

import oracle
import seqgen
import bigseqkit_amd as bsk

pytestmark = pytest.mark.gpu


def dev(data):
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8) if len(data) else torch.empty(0, dtype=torch.uint8)
    return t.cuda()


class _Opts:
    def __init__(self, d):
        self.d = dict(d)
        self._v = self.d

    def to_json(self):
        return json.dumps(self.d)


def frame(data, fastq, nshards=1):
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    if nshards == 1:
        return bsk.SeqFrame(fmt, [dev(data)])
    fr = (bsk.ReadFASTQN if fastq else bsk.ReadFASTAN)(data, nshards)
    return bsk.SeqFrame(fr.format, [dev(s) for s in fr.shards])


def inputs():
    rng = random.Random(99)
    yield "fq", True, seqgen.random_fastq(rng, 500, 0, 200)
    yield "fq-nonl", True, seqgen.random_fastq(rng, 77, 1, 90, final_newline=False)
    yield "fa60", False, seqgen.random_fasta(rng, 200, 0, 700, width=60)
    yield "fa0-nonl", False, seqgen.random_fasta(rng, 150, 0, 300, width=0, final_newline=False)
    yield "fa7-blank", False, seqgen.random_fasta(rng, 90, 0, 100, width=7, trailing_blank=2)
    yield "tiny", False, b"".join(b">%d\n%s\n" % (i, b"ACGT"[i % 4:i % 4 + 1]) for i in range(20000))
    big = "".join(random.Random(5).choice("ACGT") for _ in range(300_000))
    yield "long", False, (">c1 x\n" + "".join(big[j:j + 60] + "\n" for j in range(0, len(big), 60)) + ">c2\nACGT\n").encode()


INPUTS = list(inputs())


@pytest.mark.parametrize("name,fastq,data", INPUTS, ids=[i[0] for i in INPUTS])
def test_fq2fa(name, fastq, data, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    for o in ({}, {"Config": {"LineWidth": 10}}, {"Config": {"IDNCBI": True}}):
        assert bsk.Fq2Fa(frame(data, fastq), _Opts(o)) == oracle.fq2fa(data, fastq, json.dumps(o)), o
    assert bsk.Fq2Fa(frame(data, fastq, 3), _Opts({})) == oracle.fq2fa(data, fastq, "{}", nparts=3)


@pytest.mark.parametrize("name,fastq,data", INPUTS, ids=[i[0] for i in INPUTS])
def test_duplicate(name, fastq, data, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    for times in (1, 2, 5, 0):
        o = {"Times": times}
        assert bsk.Duplicate(frame(data, fastq), _Opts(o)) == oracle.duplicate(data, fastq, json.dumps(o)), times
    assert bsk.Duplicate(frame(data, fastq, 4), _Opts({"Times": 3})) == oracle.duplicate(data, fastq, '{"Times": 3}', nparts=4)
    assert bsk.Duplicate(frame(data, fastq)) == oracle.duplicate(data, fastq)  # default: one copy = the input records


RANGES = ["1:1", "1:10", "5:5", "3:40", "2", "17:", "1:-1", "-1:-1", "-1:5", "-10:-1", "-10:-3", "-3", "100000:100001", "-7:1000000"]


@pytest.mark.parametrize("name,fastq,data", INPUTS, ids=[i[0] for i in INPUTS])
def test_range(name, fastq, data, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    for r in RANGES:
        o = {"Range": r}
        try:
            want = oracle.range_(data, fastq, json.dumps(o))
        except oracle.OracleError as e:
            with pytest.raises(bsk.BskError) as ei:
                bsk.Range(frame(data, fastq), _Opts(o))
            assert str(e) in str(ei.value), r
            continue
        assert bsk.Range(frame(data, fastq), _Opts(o)) == want, r
        assert bsk.Range(frame(data, fastq, 3), _Opts(o)) == want, r  # the index is global: shards do not matter


@pytest.mark.parametrize("name,fastq,data", INPUTS[:4], ids=[i[0] for i in INPUTS[:4]])
def test_head(name, fastq, data):
    for n in (1, 10, 33, 10 ** 9):
        o = {"N": n}
        assert bsk.Head(frame(data, fastq), _Opts(o)) == oracle.head(data, fastq, json.dumps(o)), n
    assert bsk.Head(frame(data, fastq, 2)) == oracle.head(data, fastq)


def test_records_option_errors_and_empty_input():
    for o, msg in (({"Range": ""}, "flag -r (--range) needed"), ({"Range": "0:4"}, "either start and end should not be 0"),
                   ({"Range": "4:0"}, "either start and end should not be 0"), ({"Range": "9:3"}, "start must be > than end"),
                   ({"Range": "a:3"}, 'strconv.ParseInt: parsing "a": invalid syntax')):
        with pytest.raises(bsk.BskError) as ei:
            bsk.Range(frame(b">a\nA\n", False), _Opts(o))
        assert msg in str(ei.value)
    with pytest.raises(bsk.BskError):
        bsk.Duplicate(frame(b">a\nA\n", False), _Opts({"Times": -1}))
    with pytest.raises(bsk.BskError) as ei:
        bsk.Fq2Fa(frame(b">a\nA\n", False), _Opts({"Config": {"SeqType": "bogus"}}))
    assert "invalid sequence type" in str(ei.value)
    for fn, o in ((bsk.Fq2Fa, {}), (bsk.Duplicate, {"Times": 2}), (bsk.Range, {"Range": "1:5"}), (bsk.Head, {})):
        assert fn(frame(b"", True), _Opts(o)) == b""


def test_range_backend_of_the_multi_gpu_path_with_virtual_ranks():
    """dist.HipRangeBackend is what every rank runs between the all_gather of the counts and the store: emulate three
    ranks on one GPU (the collective itself is covered by the world_size-2 gloo test)"""
    from bigseqkit_amd import dist as bdist
    rng = random.Random(2)
    data = seqgen.random_fastq(rng, 700, 0, 90)
    bounds = bdist.shard_bounds(data, 3, bsk.FORMAT_FASTQ)
    for op_name, o in (("Range", {"Range": "-250:-20"}), ("Head", {"N": 400}), ("Range", {"Range": "30:31"})):
        backs = [bdist.HipRangeBackend(op_name, json.dumps(o), 0) for _ in bounds]
        shards = [dev(data[lo:hi]) for lo, hi in bounds]
        counts = [b.count(s, bsk.FORMAT_FASTQ) for b, s in zip(backs, shards)]
        got = b"".join(b.run(sum(counts[:r]), sum(counts)) for r, b in enumerate(backs))
        want = oracle.head(data, True, json.dumps(o)) if op_name == "Head" else oracle.range_(data, True, json.dumps(o))
        assert got == want and len(want) > 0
        for b in backs:
            b.close()
