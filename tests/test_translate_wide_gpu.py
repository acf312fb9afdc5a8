"""k_translate_wide (sixteen codon slots per lane, 16-byte stores) against the oracle: every record length around the
lane / step boundaries, source line widths 0 / 50 / 60 / 70, output line widths 0 / 16 / 17 / 60, all frames, --trim, -M,
--clean, and texts it must hand to k_translate_frames4 (N, IUPAC letters, narrow lines)."""
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


def fasta(recs, width):
    out = []
    for name, s in recs:
        if width:
            body = "".join(s[i:i + width] + "\n" for i in range(0, len(s), width))
        else:
            body = s + "\n"
        out.append(">%s\n%s" % (name, body))
    return "".join(out).encode()


def check(data, fastq, opts, monkeypatch):
    want = oracle.translate(data, fastq, json.dumps(opts))
    fr = bsk.SeqFrame(bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA, [dev(data)])
    got = bsk.Translate(fr, _Opts(opts))
    assert got == want, (opts, len(got), len(want))
    monkeypatch.setenv("BSK_TRANSLATE", "frames4")     # the same through the general kernel alone
    assert bsk.Translate(fr, _Opts(opts)) == want
    monkeypatch.delenv("BSK_TRANSLATE")


LENGTHS = list(range(0, 60)) + list(range(90, 105)) + list(range(140, 160)) + [700, 767, 768, 769, 770, 771, 815, 816, 817, 1535, 1536, 1537,
                                                                         3071, 3072, 3073, 3074, 3075, 3119, 3120, 3121, 4999, 5000, 5001, 5002, 6143, 6144, 6145, 9300]


@pytest.mark.parametrize("width", [0, 50, 60, 70])
@pytest.mark.parametrize("lw", [0, 16, 17, 60])
def test_all_lengths_all_frames(width, lw, monkeypatch):
    rng = random.Random(width * 100 + lw)
    recs = [("s%d len=%d" % (i, L), "".join(rng.choice("ACGT") for _ in range(L))) for i, L in enumerate(LENGTHS)]
    rng.shuffle(recs)
    data = fasta(recs, width)
    check(data, False, {"Frame": ["6"], "LineWidth": lw}, monkeypatch)


@pytest.mark.parametrize("opts", [{"Frame": ["1"]}, {"Frame": ["-2"]}, {"Frame": ["2", "-3"]}, {"Frame": ["6"], "Trim": True},
                                  {"Frame": ["6"], "InitCodonAsM": True}, {"Frame": ["6"], "Clean": True, "AppendFrame": True},
                                  {"Frame": ["6"], "TranslTable": 11, "Trim": True, "InitCodonAsM": True},
                                  {"Frame": ["3", "1", "-1"], "TranslTable": 2}])
def test_options(opts, monkeypatch):
    rng = random.Random(len(json.dumps(opts)))
    recs = []
    for i in range(120):
        L = rng.choice([0, 1, 2, 3, 5, 47, 48, 49, 50, 51, 95, 96, 97, 150, 767, 768, 769, 800, 3100, 5001])
        s = "".join(rng.choice("ACGTacgt" if i % 3 == 0 else "ACGT") for _ in range(L))
        if i % 4 == 0 and L >= 9:
            s = "ATG" + s[3:-3] + rng.choice(["TAA", "TAG", "TGA"])       # start and stop codons: -M and --trim matter
        if i % 5 == 0 and L >= 12:
            s = rng.choice(["TTATTA", "CTACTA", "TCATCA"]) + s[6:]         # stops of the reverse strand at its end
        recs.append(("r%d some desc" % i, s))
    check(fasta(recs, 60), False, opts, monkeypatch)
    check(fasta(recs, 0), False, opts, monkeypatch)


def test_fastq_reads(monkeypatch):
    rng = random.Random(3)
    recs = []
    for i in range(500):
        L = rng.choice([150, 150, 150, 151, 100, 36, 250, 49, 50])
        recs.append("@q%d\n%s\n+\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(L)), "I" * L))
    check("".join(recs).encode(), True, {"Frame": ["6"]}, monkeypatch)


def test_records_it_must_hand_over(monkeypatch):
    """an N, an IUPAC letter, RNA, a gap, narrow source lines, narrow output lines: k_translate_frames4 takes those records"""
    rng = random.Random(8)
    recs = []
    for i in range(200):
        L = rng.choice([30, 160, 800, 3100])
        s = [rng.choice("ACGT") for _ in range(L)]
        kind = i % 5
        if kind == 1: s[rng.randrange(L)] = "N"
        if kind == 2: s[rng.randrange(L)] = rng.choice("RYKMSWBDHV")
        if kind == 3: s = [c if c != "T" else "U" for c in s]
        if kind == 4: s[L - 1] = "n"                                       # in the very last window of the record
        recs.append(("m%d" % i, "".join(s)))
    for width in (60, 0):
        check(fasta(recs, width), False, {"Frame": ["6"], "AllowUnknownCodon": True}, monkeypatch)
    plain = [("p%d" % i, "".join(rng.choice("ACGT") for _ in range(rng.choice([10, 200, 1000])))) for i in range(50)]
    check(fasta(plain, 20), False, {"Frame": ["6"]}, monkeypatch)           # source lines narrower than a lane's window
    check(fasta(plain, 60), False, {"Frame": ["6"], "LineWidth": 7}, monkeypatch)   # output lines narrower than 16 residues
    check(fasta(plain, 60), False, {"Frame": ["6"], "LineWidth": 3}, monkeypatch)
