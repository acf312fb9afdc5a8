"""Parity of the `faidx` index rows against the CPU oracle, through the C ABI."""
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


def test_faidx_hand_cases():
    fa = b">a x\nACGTACGT\nACGTACGT\nACG\n>b\nGG\n>c\n>d q\nTTTT\nTT"
    assert bsk.Faidx(frame(fa, False)) == b"a\t19\t5\t8\t9\nb\t2\t30\t2\t3\nc\t0\t36\t0\t0\nd\t6\t41\t4\t5\n" == oracle.faidx(fa, False)
    assert bsk.Faidx(frame(fa, False), _Opts({"FullHead": True})).startswith(b"a x\t19\t5\t8\t9\n")
    fq = b"@r1 d\nACGT\n+\nIIII\n@r2\nGG\n+r2\n##\n"
    assert bsk.Faidx(frame(fq, True)) == b"r1\t4\t6\t4\t5\t13\nr2\t2\t22\t2\t3\t29\n" == oracle.faidx(fq, True)
    assert bsk.Faidx(frame(b"", False)) == b""
    for bad in (b">ok\nAC\n>a\nACGT\nACGTAC\nAC\n", b">a z\nACGTAC\nACGT\nAC\n", b">a\nAC\nACGT\n"):
        with pytest.raises(oracle.OracleError) as e1:
            oracle.faidx(bad, False)
        with pytest.raises(bsk.BskError) as e2:
            bsk.Faidx(frame(bad, False))
        assert str(e1.value) in str(e2.value) and "different line length in sequence: a." in str(e2.value)
    # one change of width, downwards, is what the reference accepts (also in the middle)
    ok = b">a\nACGTAC\nACGTAC\nACG\nACG\n"
    assert bsk.Faidx(frame(ok, False)) == oracle.faidx(ok, False) == b"a\t18\t3\t6\t7\n"


@pytest.mark.parametrize("width", [60, 0, 7, 16])
def test_faidx_fasta(width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(width)
    data = seqgen.random_fasta(rng, 400, 0, 900, width=width, final_newline=width != 7)
    for o in ({}, {"FullHead": True}, {"Config": {"IDNCBI": True}}):
        assert bsk.Faidx(frame(data, False), _Opts(o)) == oracle.faidx(data, False, json.dumps(o)), o
    # shards: the offsets are file offsets
    assert bsk.Faidx(frame(data, False, 3)) == oracle.faidx(data, False)


def test_faidx_fastq_and_long_records(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(3)
    fq = seqgen.random_fastq(rng, 500, 0, 200)
    assert bsk.Faidx(frame(fq, True)) == oracle.faidx(fq, True)
    assert bsk.Faidx(frame(fq, True, 4)) == oracle.faidx(fq, True)
    big = "".join(rng.choice("ACGT") for _ in range(400_000))
    fa = (">chr1 x\n" + "".join(big[j:j + 60] + "\n" for j in range(0, len(big), 60)) + ">chr2\nACGT\n").encode()
    assert bsk.Faidx(frame(fa, False)) == oracle.faidx(fa, False) == b"chr1\t400000\t8\t60\t61\nchr2\t4\t406681\t4\t5\n"


# ---------------------------------------------------------------- region queries (FaidxQuery)
QUERIES = [["chr1"], ["chr1:2-5", "chr2:-3", "chr2:1-2"], ["chr1:5-2"], ["chr1:-5--1"], ["chr1:3"], ["chr1:15-"], ["chr3"], ["chr9"],
           ["chr2:100-200"], ["Chr3:2-", "chr1:-4", "x:y:1-2"]]


@pytest.mark.parametrize("qi", range(len(QUERIES)))
def test_faidx_queries_hand_cases(qi, tmp_path):
    fa = b">chr1 x\nACGTACGTAC\nGGGGGTTTTT\n>chr2\nAAAACCCC\n>Chr3\nTTTT\n>x:y d\nACGTT\n"
    for extra in ({}, {"IgnoreCase": True}, {"Config": {"LineWidth": 3}}):
        o = dict({"Regions": QUERIES[qi]}, **extra)
        assert bsk.FaidxQuery(frame(fa, False), _Opts(o)) == oracle.faidx_query(fa, False, json.dumps(o)), o
    rf = tmp_path / "regions.txt"
    rf.write_text("\n".join(QUERIES[qi]) + "\n\n")
    o = {"RegionFile": str(rf), "Regions": ["chr2:2-3"]}
    assert bsk.FaidxQuery(frame(fa, False), _Opts(o)) == oracle.faidx_query(fa, False, json.dumps(o))


def test_faidx_queries_random_and_fastq(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(8)
    for fastq in (False, True):
        data = seqgen.random_fastq(rng, 300, 0, 150) if fastq else seqgen.random_fasta(rng, 300, 0, 400, width=60)
        ids = [f"r{k}" if fastq else f"s{k}" for k in range(300)]
        qs = []
        for _ in range(120):
            i = rng.choice(ids)
            kind = rng.randrange(6)
            b, e = rng.randint(-60, 120), rng.randint(-60, 120)
            if b == 0: b = 1
            if e == 0: e = -1
            qs.append([i, f"{i}:{b}-{e}", f"{i}:{abs(b)}", f"{i}:{b}-", f"{i}:-{abs(e)}", i.upper()][kind])
        for extra in ({}, {"IgnoreCase": True}):
            o = dict({"Regions": qs}, **extra)
            want = oracle.faidx_query(data, fastq, json.dumps(o))
            assert bsk.FaidxQuery(frame(data, fastq), _Opts(o)) == want and len(want) > 100
    # -r: the queries are regular expressions on the ID, the hits come back whole
    fa = seqgen.random_fasta(rng, 200, 0, 300, width=60)
    for qs in (["^s1\\d$"], ["7$", "^s2"], ["s(3|4)5", "zzz"], ["^S1"]):
        o = {"Regions": qs, "UseRegexp": True, "Config": {"LineWidth": 40}}
        want = oracle.faidx_query(fa, False, json.dumps(o))
        assert bsk.FaidxQuery(frame(fa, False), _Opts(o)) == want, qs
    assert len(oracle.faidx_query(fa, False, json.dumps({"Regions": ["7$", "^s2"], "UseRegexp": True}))) > 500
    with pytest.raises(bsk.BskError) as e:
        bsk.FaidxQuery(frame(b">a\nA\n", False), _Opts({"Regions": ["a("], "UseRegexp": True}))
    assert "invalid regular expression: a(" in str(e.value)
