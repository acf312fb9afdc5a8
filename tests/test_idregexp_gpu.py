"""Custom --id-regexp (FindSubmatch(head)[1], /root/reference/bigseqkit-lib/helper.go:362-368) on the HIP path: every
operator that uses record IDs, against the oracle (whose engine is std::regex -- the product runs its own Pike VM)."""
import json
import random

import pytest

import oracle
import bigseqkit_amd as bsk

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


def headers(rng, n):
    out = []
    for i in range(n):
        kind = i % 5
        if kind == 0: h = "gi|%d|ref|NC_%06d.%d| Pseudomonas sp" % (rng.randrange(10**6), rng.randrange(10**6), rng.randint(1, 3))
        elif kind == 1: h = "read_%d/1 sample=%s" % (i % 40, rng.choice("ABC"))
        elif kind == 2: h = "id=%s;len=%d" % (rng.choice(["aa", "bb9", "c_c"]), i % 30)
        elif kind == 3: h = "plain%d" % (i % 25)
        else: h = "x y z %d" % (i % 7)
        out.append(h)
    return out


REGEXPS = [r"^([^\s/]+)", r"gi\|(\d+)\|", r"id=(\w+)", r"^(\S+?)_", r"(\d+)$", r"^(?:read|plain)(_?\d+)", r"sample=([A-Z])|^(x)", r"\b(\w+)\b$", r"^(\w+)\b"]


@pytest.mark.parametrize("re_", REGEXPS)
def test_operators_that_use_ids(re_, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(len(re_))
    hs = headers(rng, 600)
    fq = "".join("@%s\n%s\n+\n%s\n" % (h, s, "I" * len(s)) for h, s in ((h, "".join(rng.choice("ACGT") for _ in range(rng.randint(5, 60)))) for h in hs)).encode()
    fa = "".join(">%s\n%s\n" % (h, "".join(rng.choice("ACGT") for _ in range(rng.randint(5, 90)))) for h in hs).encode()
    cfg = {"Config": {"IDRegexp": re_}}
    fr_q = lambda: bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(fq)])
    fr_a = lambda: bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(fa)])

    def chk(fn, ofn, data, fastq, opts, fr):
        o = dict(opts, **cfg)
        assert fn(fr(), _Opts(o)) == ofn(data, fastq, json.dumps(o)), (re_, opts)

    chk(bsk.Seq, oracle.seq, fq, True, {"Name": True, "OnlyId": True}, fr_q)
    chk(bsk.Seq, oracle.seq, fa, False, {"OnlyId": True}, fr_a)
    # grep by ID: a few IDs taken from the oracle's own parse of the headers
    ids = sorted(set(oracle.seq(fq, True, json.dumps(dict({"Name": True, "OnlyId": True}, **cfg))).decode().split("\n")))
    pats = [x for x in ids if x][:6] or ["none"]
    chk(bsk.Grep, oracle.grep, fq, True, {"Pattern": pats}, fr_q)
    chk(bsk.Grep, oracle.grep, fq, True, {"Pattern": pats[:2], "InvertMatch": True, "IgnoreCase": True}, fr_q)
    chk(bsk.RmDup, oracle.rmdup, fq, True, {}, fr_q)
    chk(bsk.Sort, oracle.sort, fq, True, {}, fr_q)
    chk(bsk.Rename, oracle.rename, fa, False, {}, fr_a)
    chk(bsk.Translate, oracle.translate, fa, False, {"Frame": ["1", "-1"], "AppendFrame": True}, fr_a)
    chk(bsk.Locate, oracle.locate, fa, False, {"Pattern": ["ACG"]}, fr_a)
    chk(bsk.Faidx, oracle.faidx, fa, False, {}, fr_a)


def test_rejections():
    for re_, msg in ((r"^\S+", "must contain"), (r"^(\S+))", "fail to compile regexp"), (r"^(\pL+)", "not supported by the HIP path")):
        with pytest.raises(bsk.BskError) as e:
            bsk.Operator("SeqTransform", json.dumps({"Config": {"IDRegexp": re_}}), 0)
        assert msg in str(e.value), (re_, str(e.value))
