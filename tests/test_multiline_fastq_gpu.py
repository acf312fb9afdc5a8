"""Multi-line FASTQ (sequence and quality wrapped over several lines; SeqParser.Read accepts it,
bigseqkit-lib/helper.go:252-269): the HIP path rewrites such a shard as 4-line FASTQ on the device (ops_mlfq.hip) and
runs the operator on that; results are compared with the oracle reading the ORIGINAL text (PARITY.md SPLIT-FQ)."""
import json
import random

import pytest

import oracle
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib

pytestmark = pytest.mark.gpu


def dev(data):
    import torch
    return torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()


class _Opts:
    def __init__(self, d):
        self.d = dict(d)
        self._v = self.d

    def to_json(self):
        return json.dumps(self.d)


def wrapped_fastq(rng, nrec, width, final_newline=True, trailing_blank=0, qwidth=None):
    qwidth = qwidth or width
    out = []
    for i in range(nrec):
        L = rng.choice([0, 1, width - 1, width, width + 1, 2 * width, rng.randint(0, 6 * width)])
        seq = "".join(rng.choice("ACGTN") for _ in range(L))
        qual = "".join(chr(rng.randint(33, 126)) for _ in range(L))
        if L > width and rng.random() < 0.5:    # a quality CONTINUATION line that looks like a '+' line
            k = qwidth * rng.randint(1, (L - 1) // qwidth) if (L - 1) // qwidth >= 1 else 0
            if 0 < k < L:
                qual = qual[:k] + "+" + qual[k + 1:]
        # ('@' at the start of a continuation line ends the record under the grammar, PARITY.md SPLIT-FQ: see the
        #  malformed cases below)
        for k in range(qwidth, L, qwidth):
            if qual[k] == "@":
                qual = qual[:k] + "A" + qual[k + 1:]
        if L and rng.random() < 0.2:
            qual = rng.choice("@+") + qual[1:]
        name = "r%d" % i + (" d%d" % rng.randint(0, 99) if rng.random() < 0.5 else "")
        plus = "+" + (name if rng.random() < 0.2 else "")
        sl = [seq[j:j + width] for j in range(0, L, width)] or [""]
        ql = [qual[j:j + qwidth] for j in range(0, L, qwidth)] or [""]
        out.append("@%s\n%s\n%s\n%s\n" % (name, "\n".join(sl), plus, "\n".join(ql)))
    s = "".join(out)
    if not final_newline:
        s = s[:-1]
    return (s + "\n" * trailing_blank).encode()


def frame(data, on_device=True):
    return bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(data) if on_device else data])


OPS = [
    ("seq", {}),
    ("seq", {"Name": True}),
    ("seq", {"Name": True, "OnlyId": True}),
    ("seq", {"Reverse": True, "Complement": True}),
    ("seq", {"MinLen": 10}),
    ("grep", {"Pattern": ["ACG"], "BySeq": True}),
    ("grep", {"Pattern": ["ACGTACGTACGT"], "BySeq": True, "InvertMatch": True}),
    ("locate", {"Pattern": ["ACG"]}),
    ("subseq", {"Region": "2:9"}),
    ("rmdup", {"BySeq": True}),
    ("fq2fa", {}),
]


def run(cmd, fr, opts):
    o = _Opts(opts)
    return {"seq": bsk.Seq, "grep": bsk.Grep, "locate": bsk.Locate, "subseq": bsk.Subseq, "rmdup": bsk.RmDup,
            "fq2fa": bsk.Fq2Fa}[cmd](fr, o)


def want_of(cmd, data, opts):
    j = json.dumps(opts)
    if cmd == "rmdup":
        return oracle.rmdup(data, True, j)
    return getattr(oracle, cmd)(data, True, j)


@pytest.mark.parametrize("k", range(len(OPS)))
@pytest.mark.parametrize("seed,width", [(0, 7), (1, 60), (2, 1)])
def test_operators_on_multiline_fastq(seed, width, k, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "2048")
    rng = random.Random(3100 + 31 * seed + k)
    data = wrapped_fastq(rng, 400, width, final_newline=seed != 1, trailing_blank=2 if seed == 2 else 0)
    assert not oracle.is_strict_4line_fastq(data)
    cmd, opts = OPS[k]
    assert run(cmd, frame(data), opts) == want_of(cmd, data, opts)


@pytest.mark.parametrize("seed,width", [(0, 7), (1, 60), (2, 13)])
def test_stats_on_multiline_fastq(seed, width):
    rng = random.Random(3200 + seed)
    data = wrapped_fastq(rng, 700, width, final_newline=seed != 1, qwidth=width if seed != 2 else 5)
    for opts in ({"All": True}, {}):
        o = bsk.SeqKitStatsOptions()
        for kk, v in opts.items():
            getattr(o, kk)(v)
        for on_device in (True, False):
            got = bsk.StatsString("input0", "N/A", frame(data, on_device), o)
            assert got == oracle.stats_string(data, True, json.dumps(opts))


def test_stats_judges_every_shard_by_its_own_head():
    # a 4-line shard followed by a wrapped one (and the other way round): each is read by the reader its own head calls for
    rng = random.Random(3250)
    plain = "".join("@p%d\n%s\n+\n%s\n" % (i, "ACGTN" * (i % 9), "IIIII" * (i % 9)) for i in range(300)).encode()
    wrapped = wrapped_fastq(rng, 300, 9)
    for a, b in ((plain, wrapped), (wrapped, plain)):
        for opts in ({"All": True}, {}):
            o = bsk.SeqKitStatsOptions()
            for kk, v in opts.items():
                getattr(o, kk)(v)
            for on_device in (True, False):
                fr = bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(a), dev(b)] if on_device else [a, b])
                assert bsk.StatsString("input0", "N/A", fr, o) == oracle.stats_string(a + b, True, json.dumps(opts))


def test_multiline_head_example_from_the_parser():
    data = b"@a\nACGT\nAC\n+\nIIII\nII\n@b desc\nA\nC\nG\n+b\n@\n+\nI\n"
    assert run("seq", frame(data), {}) == oracle.seq(data, True, "{}") == b"@a\nACGTAC\n+\nIIIIII\n@b desc\nACG\n+\n@+I\n"
    assert run("seq", frame(data, on_device=False), {"Name": True}) == b"a\nb desc\n"


def test_malformed_multiline_fastq_is_an_error():
    for bad in (b"@a\nACGT\nAC\n+\nIIII\nIII\n",            # quality longer than the sequence
                b"@a\nACGT\nAC\n+\nIIII\n@b\nAC\n+\nII\n",    # quality shorter, then a header
                b"@a\nACGT\nAC\nIIII\nII\n",                  # no '+' line
                b"@a\nACGT\nAC\n+\nIIII\n@I\n"):               # a quality continuation line that begins with '@'
        with pytest.raises(oracle.OracleError):
            oracle.seq(bad, True, "{}")
        with pytest.raises(bsk.BskError) as e:
            run("seq", frame(bad), {})
        assert e.value.code in (_lib.BSK_ERR_FORMAT, _lib.BSK_ERR_UNSUPPORTED)


@pytest.mark.parametrize("seed,width", [(0, 7), (1, 60), (2, 1)])
def test_record_text_operators_on_multiline_fastq(seed, width, monkeypatch):
    # range / head / duplicate print the record TEXT, wrapped as it stands (bigseqkit-lib/range.go, duplicate.go:22-29): the
    # multi-line reader says where the records begin, the text leaves verbatim
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "2048")
    rng = random.Random(3300 + seed)
    data = wrapped_fastq(rng, 300, width, final_newline=seed != 1, trailing_blank=2 if seed == 2 else 0)
    assert not oracle.is_strict_4line_fastq(data)
    assert bsk.Head(frame(data), _Opts({"N": 7})) == oracle.head(data, True, '{"N": 7}')
    for r in ("1:12", "5:9", "-20:-1", "250:400"):
        assert bsk.Range(frame(data), _Opts({"Range": r})) == oracle.range_(data, True, json.dumps({"Range": r})), r
    for times in (1, 2, 3, 7):
        assert bsk.Duplicate(frame(data), _Opts({"Times": times})) == oracle.duplicate(data, True, json.dumps({"Times": times})), times


def test_records_wrapped_only_further_down(monkeypatch):
    # the head of the shard (what the reader is chosen from) is 4-line FASTQ, the records behind it are wrapped: the strict
    # reader gives up, the multi-line reader takes the shard
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(3400)
    plain = "".join("@p%d\n%s\n+\n%s\n" % (i, "ACGT" * 20, "IIII" * 20) for i in range(5000)).encode()   # > 256 KiB
    assert len(plain) > 300 * 1024
    data = plain + wrapped_fastq(rng, 200, 11)
    assert run("seq", frame(data), {}) == oracle.seq(data, True, "{}")
    assert run("grep", frame(data), {"Pattern": ["ACG"], "BySeq": True}) == oracle.grep(data, True, '{"Pattern": ["ACG"], "BySeq": true}')
    assert bsk.Head(frame(data), _Opts({"N": 5100})) == oracle.head(data, True, '{"N": 5100}')
    assert bsk.Range(frame(data), _Opts({"Range": "-50:-1"})) == oracle.range_(data, True, '{"Range": "-50:-1"}')
    # a shard that is FASTQ under neither reader keeps the strict reader's complaint
    bad = plain + b"@x\nACGT\n+\nIII\n"
    with pytest.raises(bsk.BskError) as e:
        run("seq", frame(bad), {})
    assert "unmatched length" in str(e.value)


def test_two_input_operators_on_multiline_fastq(monkeypatch):
    # pair / common / concat read several texts: every text is rewritten by itself
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "2048")
    rng = random.Random(3500)
    a = wrapped_fastq(rng, 120, 9)
    # a second file with (mostly) the same IDs on four lines each
    b4 = "".join("@r%d%s\n%s\n+\n%s\n" % (i, " mate" if i % 3 else "", "ACGTN"[i % 5] * (i % 40), "I" * (i % 40))
                 for i in range(0, 150, 1) if i % 7).encode()
    assert oracle.is_strict_4line_fastq(b4) and not oracle.is_strict_4line_fastq(a)
    fa, fb = frame(a), frame(b4)
    for x, y, xd, yd in ((fa, fb, a, b4), (fb, fa, b4, a), (fa, fa, a, a)):
        assert bsk.Common(x, y, _Opts({})) == oracle.common([xd, yd], True, "{}")
        assert bsk.Concat(x, y, _Opts({})) == oracle.concat(xd, yd, True, "{}")
        assert tuple(bsk.Pair(x, y, _Opts({"SaveUnpaired": True}))) == tuple(oracle.pair(xd, yd, True, '{"SaveUnpaired": true}'))


def test_file_to_file_pipeline_on_multiline_fastq(tmp_path):
    import ctypes as C
    rng = random.Random(3400)
    data = wrapped_fastq(rng, 2000, 60)
    want = oracle.seq(data, True, '{"Reverse": true}')
    out = tmp_path / "o.fq"
    st = C.c_void_p()
    _lib.check(_lib.lib.bsk_store_open(str(out).encode(), 1, C.byref(st)))
    with bsk.Operator("SeqTransform", '{"Reverse": true}', 0) as op:
        b = C.create_string_buffer(data, len(data))
        nb, nr = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib.bsk_run_to_store(op.ctx, b, len(data), bsk.FORMAT_FASTQ, 0, st, 0, C.byref(nb), C.byref(nr)), op.ctx)
    tot = C.c_uint64()
    _lib.check(_lib.lib.bsk_store_close(st, C.byref(tot)))
    assert out.read_bytes() == want


def test_file_to_file_duplicate_on_multiline_fastq(tmp_path):
    # the record-text operators through the chunked pipeline: the wrapped shard goes as one piece, the text verbatim
    import ctypes as C
    rng = random.Random(3600)
    data = wrapped_fastq(rng, 500, 13)
    want = oracle.duplicate(data, True, '{"Times": 2}')
    out = tmp_path / "d.fq"
    st = C.c_void_p()
    _lib.check(_lib.lib.bsk_store_open(str(out).encode(), 1, C.byref(st)))
    with bsk.Operator("Duplicate", '{"Times": 2}', 0) as op:
        b = C.create_string_buffer(data, len(data))
        nb, nr = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib.bsk_run_to_store(op.ctx, b, len(data), bsk.FORMAT_FASTQ, 0, st, 0, C.byref(nb), C.byref(nr)), op.ctx)
    tot = C.c_uint64()
    _lib.check(_lib.lib.bsk_store_close(st, C.byref(tot)))
    assert out.read_bytes() == want


@pytest.mark.parametrize("width", [60, 13])
def test_host_shard_in_small_staging_chunks(width, tmp_path, monkeypatch):
    """the staging pipelines (bsk_stats_run on a host buffer, bsk_run_to_store) cut a wrapped host shard on record starts of
    the wrapped grammar (round 4; before, a wrapped shard went as one piece)"""
    import ctypes as C
    monkeypatch.setenv("BSK_STAGE_BYTES", "20000")
    rng = random.Random(3700 + width)
    data = wrapped_fastq(rng, 3000, width)
    assert len(data) > 5 * 20000
    for opts in ({"All": True}, {}):
        o = bsk.SeqKitStatsOptions()
        for kk, v in opts.items():
            getattr(o, kk)(v)
        assert bsk.StatsString("input0", "N/A", frame(data, False), o) == oracle.stats_string(data, True, json.dumps(opts))
    want = oracle.seq(data, True, '{"Reverse": true}')
    out = tmp_path / "o.fq"
    st = C.c_void_p()
    _lib.check(_lib.lib.bsk_store_open(str(out).encode(), 1, C.byref(st)))
    with bsk.Operator("SeqTransform", '{"Reverse": true}', 0) as op:
        b = C.create_string_buffer(data, len(data))
        nb, nr = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib.bsk_run_to_store(op.ctx, b, len(data), bsk.FORMAT_FASTQ, 0, st, 0, C.byref(nb), C.byref(nr)), op.ctx)
    tot = C.c_uint64()
    _lib.check(_lib.lib.bsk_store_close(st, C.byref(tot)))
    assert out.read_bytes() == want


@pytest.mark.parametrize("world", [2, 3])
def test_rmdup_exchange_on_wrapped_shards(world, monkeypatch):
    from test_translate_rmdup_gpu import _virtual_ranks
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(3800 + world)
    seqs, recs = [], []
    for i in range(1500):
        s = seqs[rng.randrange(len(seqs))] if (i > 5 and rng.random() < 0.3) else "".join(rng.choice("ACGTacgt") for _ in range(rng.randint(1, 200)))
        seqs.append(s)
        q = [chr(rng.randint(35, 73)) for _ in s]
        for k in range(50, len(s), 50):
            if q[k] == "@":
                q[k] = "A"     # ('@' at the start of a continuation line begins a record under the grammar)
        q = "".join(q)
        recs.append("@r%d c\n%s\n+\n%s\n" % (i, "\n".join(s[j:j + 50] for j in range(0, len(s), 50)), "\n".join(q[j:j + 50] for j in range(0, len(q), 50))))
    data = "".join(recs).encode()
    for o in ({"BySeq": True}, {"BySeq": True, "IgnoreCase": True}, {}):
        assert _virtual_ranks(data, True, o, world) == oracle.rmdup(data, True, json.dumps(o))
