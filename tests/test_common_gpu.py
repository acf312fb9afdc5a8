"""Parity of `common` (PARITY.md COMMON) against the CPU oracle, through the C ABI."""
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


def make(rng, n, fastq, pool, width=60):
    out = []
    for _ in range(n):
        k = rng.randrange(300)
        name = rng.choice(["id", "ID"]) + str(k) + rng.choice(["", " d", " other text"])
        s = rng.choice(pool)
        if rng.random() < 0.3:
            s = s.lower()
        if fastq:
            out.append(f"@{name}\n{s}\n+\n{'I' * len(s)}\n")
        else:
            out.append(f">{name}\n" + "".join(s[j:j + width] + "\n" for j in range(0, len(s), width)))
    return "".join(out).encode()


@pytest.mark.parametrize("fastq", [True, False])
@pytest.mark.parametrize("nfiles", [2, 3])
def test_common_matches_oracle(fastq, nfiles, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(40 + nfiles + fastq)
    pool = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 150))) for _ in range(200)]
    files = [make(rng, 500, fastq, pool) for _ in range(nfiles)]
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    frames = [bsk.SeqFrame(fmt, [dev(f)]) for f in files]
    for o in ({}, {"ByName": True}, {"BySeq": True}, {"IgnoreCase": True}, {"BySeq": True, "IgnoreCase": True},
              {"ByName": True, "IgnoreCase": True, "Config": {"LineWidth": 20}}):
        want = oracle.common(files, fastq, json.dumps(o))
        got = bsk.Common(frames[0], frames[1], _Opts(o), *frames[2:])
        assert got == want, (o, len(got), len(want))
        assert want.count(b"\n") > 4, o


def test_common_hand_cases_and_cli(tmp_path):
    a = b">x 1\nACGT\n>y\nGG\n>x 2\nTT\n>z\nAC\n"
    b = b">z q\nAA\n>x\nC"                       # no newline at the end
    c = b">X\nA\n>z\nT\n"
    fr = [bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(x)]) for x in (a, b, c)]
    assert bsk.Common(fr[0], fr[1], _Opts({})) == b">x 1\nACGT\n>z\nAC\n"
    assert bsk.Common(fr[0], fr[1], _Opts({}), fr[2]) == b">z\nAC\n"
    assert bsk.Common(fr[0], fr[1], _Opts({"IgnoreCase": True}), fr[2]) == b">x 1\nACGT\n>z\nAC\n"
    assert bsk.Common(fr[0], fr[1], _Opts({"BySeq": True})) == b"" == bsk.Common(fr[0], fr[1], _Opts({"BySeq": True, "OnlyPositiveStrand": True}))
    for o, msg in (({"BySeq": True, "ByName": True}, "only one/none of the flags"), ({"OnlyPositiveStrand": True}, "flag -s (--by-seq) needed"),
                   ):
        with pytest.raises(bsk.BskError) as e:
            bsk.Common(fr[0], fr[1], _Opts(o))
        assert msg in str(e.value)
    paths = []
    for k, x in enumerate((a, b, c)):
        p = tmp_path / f"f{k}.fa"
        p.write_bytes(x)
        paths.append(str(p))
    r = subprocess.run([CLI, "common", "-i", *paths, "-o", "-"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == b">x 1\nACGT\n>z\nAC\n" == oracle.common([a, b + b"\n", c], False, '{"IgnoreCase": true}')
