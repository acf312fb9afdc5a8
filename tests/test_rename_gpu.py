"""Parity of `rename` against the CPU oracle, through the C ABI."""
import json
import random

import pytest

import oracle
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


def make(rng, nrec, fastq, width=60, ids=40, final_newline=True):
    out = []
    for i in range(nrec):
        k = rng.randrange(ids)
        name = f"id{k}" if k % 7 else f"gi|{k}|x"
        r = rng.random()
        if r < 0.3:
            name += " " + "".join(rng.choice("abc _\t") for _ in range(rng.randint(0, 12)))
        elif r < 0.4:
            name += "\tdesc"
        elif r < 0.45:
            name += "  two spaces"
        L = rng.randint(0, 200)
        s = "".join(rng.choice("ACGTN") for _ in range(L))
        if fastq:
            q = "".join(chr(rng.randint(33, 73)) for _ in range(L))
            out.append(f"@{name}\n{s}\n+\n{q}\n")
        else:
            w = width if width > 0 else max(1, L)
            out.append(f">{name}\n" + "".join(s[j:j + w] + "\n" for j in range(0, L, w)))
    text = "".join(out)
    if not final_newline:
        text = text[:-1]
    return text.encode()


CASES = [(True, 60, 40), (False, 60, 40), (False, 0, 5), (False, 7, 300), (True, 60, 3), (False, 60, 1)]


@pytest.mark.parametrize("fastq,width,ids", CASES)
def test_rename_matches_oracle(fastq, width, ids, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(fastq * 1000 + width * 7 + ids)
    data = make(rng, 700, fastq, width, ids, final_newline=ids != 5)
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    for o in ({}, {"ByName": True}, {"Config": {"LineWidth": 11}}, {"Config": {"IDNCBI": True}}):
        want = oracle.rename(data, fastq, json.dumps(o))
        got = bsk.Rename(bsk.SeqFrame(fmt, [dev(data)]), _Opts(o))
        assert got == want, (o, got[:300], want[:300])


def test_rename_many_records():
    # 200 k records, IDs drawn from 70 k names: ordinals from the radix sort of (group, index) with groups beyond 16 bits
    rng = random.Random(606)
    out = []
    for i in range(200000):
        k = rng.randrange(70000)
        L = k % 23
        out.append(f"@n{k} d{i % 3}\n{'ACGTTGCAACGTTGCAACGTTGC'[:L]}\n+\n{'F' * L}\n")
    data = "".join(out).encode()
    want = oracle.rename(data, True, "{}")
    got = bsk.Rename(bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(data)]), _Opts({}))
    assert got == want
    assert b"_2 " in got and b"_5 " in got


def test_rename_hand_cases_and_ordinals_beyond_one_digit():
    fa = b">a x y\nACGT\n>b\nGG\n>a\tz\nTT\n>a\nC\n>b q\nA\n"
    got = bsk.Rename(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(fa)]), _Opts({}))
    assert got == b">a x y\nACGT\n>b\nGG\n>a_1 z\nTT\n>a_2 \nC\n>b_1 q\nA\n" == oracle.rename(fa, False)
    many = b"".join(b"@same read %d\nAC\n+\nII\n" % i for i in range(1234))
    got = bsk.Rename(bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(many)]), _Opts({}))
    assert got == oracle.rename(many, True)
    assert b"@same_9 read 9\n" in got and b"@same_10 read 10\n" in got and b"@same_1233 read 1233\n" in got
    assert bsk.Rename(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(b"")]), _Opts({})) == b""
    with pytest.raises(bsk.BskError):
        bsk.Rename(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(fa)]), _Opts({"Config": {"SeqType": "bogus"}}))


def test_rename_long_records(monkeypatch):
    monkeypatch.setenv("BSK_LONG_BYTES", "20000")
    rng = random.Random(4)
    big = "".join(rng.choice("ACGT") for _ in range(90_000))
    rec = ">chr1 assembly\n" + "".join(big[j:j + 60] + "\n" for j in range(0, len(big), 60))
    data = (rec + ">chr2\nACGT\n" + rec + rec).encode()
    assert bsk.Rename(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)]), _Opts({})) == oracle.rename(data, False)
