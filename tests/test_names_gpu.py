"""`seq -n` / `seq -n -i` on FASTQ written by the streaming pass (stream_names.hip) against the oracle
(SeqTransform.Call with only Name set, bigseqkit-lib/seq.go:143-175) and against the record-table path."""
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
        self.d = d

    def to_json(self):
        return json.dumps(self.d)


def fastq_with_headers(rng, nrec, final_newline=True):
    """headers of every length class the copy loop distinguishes (0 .. 70 bytes), blanks and tabs in odd places, NCBI ids"""
    out = []
    for i in range(nrec):
        k = rng.random()
        if k < 0.05:
            name = ""
        elif k < 0.15:
            name = "gi|%d|ref|NM_%06d.%d| Homo sapiens %s" % (rng.randint(1, 10 ** 6), i, rng.randint(1, 9), "x" * rng.randint(0, 30))
        elif k < 0.25:
            name = " lead%d tail" % i
        elif k < 0.35:
            name = "t%d\tafter tab and blank" % i
        else:
            n = rng.randint(1, 70)
            body = "".join(rng.choice("abcXYZ019_:/.#") for _ in range(n))
            if rng.random() < 0.5 and n > 4:
                j = rng.randrange(1, n - 1)
                body = body[:j] + rng.choice(" \t") + body[j + 1:]
            name = body
        L = rng.randint(0, 90)
        seq = "".join(rng.choice("ACGTN") for _ in range(L))
        qual = "".join(chr(rng.randint(33, 126)) for _ in range(L))
        if L and rng.random() < 0.3:
            qual = rng.choice("@+") + qual[1:]
        out.append("@%s\n%s\n+\n%s\n" % (name, seq, qual))
    s = "".join(out)
    if not final_newline:
        s = s[:-1]
    return s.encode()


def run(data, opts):
    return bsk.Seq(bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(data)]), _Opts(opts))


OPTS = [
    {"Name": True},
    {"Name": True, "OnlyId": True},
    {"Name": True, "OnlyId": True, "Config": {"IDNCBI": True}},
    {"Name": True, "Config": {"IDRegexp": "^(\\w+)"}},                    # custom expression, whole names: still the pass
    {"Name": True, "OnlyId": True, "Config": {"IDRegexp": "^(\\w+)"}},    # custom IDs: record-table path
]


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("k", range(len(OPTS)))
def test_names_pass_equals_oracle_and_table_path(seed, k, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "1024")
    rng = random.Random(7000 + 10 * seed + k)
    data = fastq_with_headers(rng, 1500, final_newline=seed % 2 == 0)
    want = oracle.seq(data, True, json.dumps(OPTS[k]))
    got = run(data, OPTS[k])
    assert got == want
    monkeypatch.setenv("BSK_NAMES", "off")
    assert run(data, OPTS[k]) == want


def test_names_slice_overflow_takes_the_table_path(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "1024")
    rng = random.Random(7100)
    data = fastq_with_headers(rng, 3000)
    want = oracle.seq(data, True, '{"Name": true}')
    monkeypatch.setenv("BSK_NAMES_SCALE", "0.02")   # slices far too small: ERR_CAPACITY -> fallback, same text
    assert run(data, {"Name": True}) == want


def test_names_sample_underestimates_later_headers(monkeypatch):
    # the head of the shard (what the slice size is estimated from) has short names, the rest long ones
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    recs = ["@s%d\nACGT\n+\nIIII\n" % i for i in range(30000)]
    recs += ["@%s_%d some description\nACGT\n+\nIIII\n" % ("L" * 60, i) for i in range(30000)]
    data = "".join(recs).encode()
    want = oracle.seq(data, True, '{"Name": true}')
    assert run(data, {"Name": True}) == want


def test_names_pass_reports_format_errors():
    bad = b"@a\nACGT\n+\nIII\n@b\nAC\n+\nII\n"          # len(seq) != len(qual)
    with pytest.raises(Exception):
        run(bad, {"Name": True})
    bad2 = b"@a\nACGT\n-\nIIII\n"                         # third line does not start with '+'
    with pytest.raises(Exception):
        run(bad2, {"Name": True})


def test_names_record_count_and_c2_layout():
    import ctypes as C
    import torch
    from bigseqkit_amd import _lib
    rb, nrec = 317, 200000
    t = torch.empty(rb * nrec, dtype=torch.uint8, device="cuda")
    assert _lib.lib.bsk_synth_device(0, 42, 0, 0, C.c_void_p(t.data_ptr()), rb * nrec, 0, None) == 0
    got = bsk.Seq(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), bsk.SeqKitSeqOptions().Name(True))
    head = bytes(t[:rb * 20000].cpu().numpy().tobytes())
    want = oracle.seq(head, True, '{"Name": true}')
    assert got[:len(want)] == want
    assert got.count(b"\n") == nrec
