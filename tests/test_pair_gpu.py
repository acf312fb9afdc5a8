"""Parity of `pair` against the CPU oracle, through the C ABI."""
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


def reads(rng, ids, fastq, tag, width=60):
    out = []
    for k in ids:
        L = rng.randint(1, 120)
        s = "".join(rng.choice("ACGT") for _ in range(L))
        name = f"read{k}" + (f" {tag}:N:0" if rng.random() < 0.7 else "")
        if fastq:
            out.append(f"@{name}\n{s}\n+\n{'I' * L}\n")
        else:
            out.append(f">{name}\n" + "".join(s[j:j + width] + "\n" for j in range(0, L, width)))
    return "".join(out).encode()


def run_pair(a, b, fastq, o):
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    return bsk.Pair(bsk.SeqFrame(fmt, [dev(a)]), bsk.SeqFrame(fmt, [dev(b)]), _Opts(o))


@pytest.mark.parametrize("fastq", [True, False])
def test_pair_matches_oracle(fastq, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(12 + fastq)
    n = 800
    ids1 = [k for k in range(n) if rng.random() < 0.9]
    ids2 = [k for k in range(n) if rng.random() < 0.9]
    ids1 += [rng.randrange(n) for _ in range(40)]   # repeated IDs: the k-th goes with the k-th
    ids2 += [rng.randrange(n) for _ in range(40)]
    rng.shuffle(ids2)                               # the second file need not be in the order of the first
    a, b = reads(rng, ids1, fastq, 1), reads(rng, ids2, fastq, 2)
    if not fastq:
        a = a[:-1]                                  # no newline at the end of file 1
    want = oracle.pair(a, b, fastq, '{"SaveUnpaired": true}')
    got = run_pair(a, b, fastq, {"SaveUnpaired": True})
    assert got == want
    assert got[0].count(b"\n") > 100 and len(got[2]) and len(got[3])
    got2 = run_pair(a, b, fastq, {})
    assert got2[:2] == want[:2] and got2[2:] == (b"", b"")


def many_reads(rng, ids, tag):
    seqs = ["".join(rng.choice("ACGT") for _ in range(L)) for L in (1, 7, 16, 33, 50)]
    out = []
    for k in ids:
        s = seqs[k % 5]
        out.append(f"@r{k} {tag}\n{s}\n+\n{'F' * len(s)}\n")
    return "".join(out).encode()


def test_pair_many_records_shuffled_mates():
    # 250 k records: key table of 2^19 slots, group values beyond 16 bits in the radix sort of (group, index)
    rng = random.Random(404)
    n = 120000
    ids1 = [k for k in range(n) if rng.random() < 0.97] + [rng.randrange(n) for _ in range(6000)]
    ids2 = [k for k in range(n) if rng.random() < 0.97] + [rng.randrange(n) for _ in range(6000)]
    rng.shuffle(ids1)
    rng.shuffle(ids2)
    a, b = many_reads(rng, ids1, 1), many_reads(rng, ids2, 2)
    want = oracle.pair(a, b, True, '{"SaveUnpaired": true}')
    got = run_pair(a, b, True, {"SaveUnpaired": True})
    assert got == want
    assert got[0].count(b"\n") > 4 * 100000 and len(got[2]) and len(got[3])


def test_pair_hand_cases(tmp_path):
    a = b"@r1 1\nAC\n+\nII\n@r2 1\nGG\n+\nII\n@r1 1b\nTT\n+\nII\n@r5\nA\n+\nI\n"
    b = b"@r2 2\nCC\n+\nII\n@r9\nT\n+\nI\n@r1 2\nGT\n+\nII\n"
    p1, p2, u1, u2 = run_pair(a, b, True, {"SaveUnpaired": True})
    assert p1 == b"@r1 1\nAC\n+\nII\n@r2 1\nGG\n+\nII\n" and p2 == b"@r1 2\nGT\n+\nII\n@r2 2\nCC\n+\nII\n"
    assert u1 == b"@r1 1b\nTT\n+\nII\n@r5\nA\n+\nI\n" and u2 == b"@r9\nT\n+\nI\n"
    assert (p1, p2, u1, u2) == oracle.pair(a, b, True)
    assert run_pair(a, b"", True, {"SaveUnpaired": True}) == (b"", b"", a, b"")
    assert run_pair(b"", b"", True, {}) == (b"", b"", b"", b"")
    fa, fb = tmp_path / "a.fq", tmp_path / "b.fq"
    fa.write_bytes(a)
    fb.write_bytes(b)
    out = tmp_path / "out"
    r = subprocess.run([CLI, "pair", "-u", "-O", str(out), str(fa), str(fb)], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert [(out / n).read_bytes() for n in ("paired.1", "paired.2", "unpaired.1", "unpaired.2")] == [p1, p2, u1, u2]
    r = subprocess.run([CLI, "pair", str(fa), str(fb)], capture_output=True, timeout=300)
    assert r.returncode != 0 and b"out-dir required" in r.stderr
