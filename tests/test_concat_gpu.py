"""Parity of `concat` (PARITY.md CONCAT) against the CPU oracle, through the C ABI."""
import json
import os
import random
import subprocess

import pytest

import oracle
import bigseqkit_amd as bsk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bigseqkit_amd", "bin", "bigseqkit")


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


def make(rng, n, fastq, width):
    out = []
    for _ in range(n):
        k = rng.randrange(n)
        name = f"id{k}" + rng.choice(["", " d", "\tother text"])
        L = rng.choice([0, 1, 59, 60, 61, rng.randint(0, 200)])
        s = "".join(rng.choice("ACGTN") for _ in range(L))
        if fastq:
            q = "".join(chr(rng.randint(33, 73)) for _ in range(L))
            out.append(f"@{name}\n{s}\n+\n{q}\n")
        else:
            w = width if width > 0 else max(1, L)
            out.append(f">{name}\n" + "".join(s[j:j + w] + "\n" for j in range(0, L, w)))
    return "".join(out).encode()


@pytest.mark.parametrize("fastq,width", [(True, 60), (False, 60), (False, 0), (False, 7)])
def test_concat_matches_oracle(fastq, width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(70 + width + fastq)
    a, b = make(rng, 400, fastq, width), make(rng, 400, fastq, width)
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    fa, fb = bsk.SeqFrame(fmt, [dev(a)]), bsk.SeqFrame(fmt, [dev(b)])
    for o in ({}, {"Full": True}, {"Full": True, "Config": {"LineWidth": 13}}, {"Config": {"LineWidth": 0}}):
        want = oracle.concat(a, b, fastq, json.dumps(o))
        got = bsk.Concat(fa, fb, _Opts(o))
        assert got == want, (o, got[:200], want[:200])
        assert len(want) > 100


def test_concat_many_records():
    # 200 k records with repeated IDs on both sides (every A of an ID with every B of it)
    rng = random.Random(505)
    n = 100000

    def side(tag):
        out = []
        for _ in range(n):
            k = rng.randrange(n // 2)
            L = (k * 7 + tag) % 40
            out.append(f"@id{k} {tag}\n{'ACGT' * 10}\n+\n{'I' * 40}\n".replace("ACGT" * 10, ("ACGT" * 10)[:L]).replace("I" * 40, "I" * L))
        return "".join(out).encode()

    a, b = side(1), side(2)
    fa, fb = bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(a)]), bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(b)])
    for o in ({}, {"Full": True}):
        want = oracle.concat(a, b, True, json.dumps(o))
        got = bsk.Concat(fa, fb, _Opts(o))
        assert got == want, o
        assert want.count(b"\n") > 4 * 100000


def test_concat_hand_cases_and_cli(tmp_path):
    a = b">x 1\nACGT\n>y\nGG\n>x 2\nTT\n"
    b = b">z q\nAA\n>x d\nCCC\n>x e\nG"           # no newline at the end
    fr = lambda x: bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(x)])
    assert bsk.Concat(fr(a), fr(b), _Opts({})) == b">x\nACGTCCC\n>x\nACGTG\n>x\nTTCCC\n>x\nTTG\n"
    assert bsk.Concat(fr(a), fr(b), _Opts({"Full": True, "Config": {"LineWidth": 3}})) == \
        b">x\nACG\nTCC\nC\n>x\nACG\nTG\n>y\nGG\n>x\nTTC\nCC\n>x\nTTG\n>z q\nAA\n"
    fqa, fqb = b"@r1 a\nAC\n+\nII\n", b"@r1 b\nGGG\n+\n###\n"
    assert bsk.Concat(bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(fqa)]), bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(fqb)]), _Opts({})) == \
        b"@r1\nACGGG\n+\nII###\n"
    assert bsk.Concat(fr(a), fr(b""), _Opts({})) == b"" and bsk.Concat(fr(a), fr(b""), _Opts({"Full": True})) == a
    pa, pb = tmp_path / "a.fa", tmp_path / "b.fa"
    pa.write_bytes(a)
    pb.write_bytes(b)
    r = subprocess.run([CLI, "concat", "-f", str(pa), str(pb), "-o", "-"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == oracle.concat(a, b + b"\n", False, '{"Full": true}')
