"""`subseq -r a:b` on FASTQ written by the streaming pass (stream_subseq.hip) against the oracle
(SubseqTransform.Call in region mode, bigseqkit-lib/subseq.go:167-191, 314-317) and against the record-table path."""
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


def fastq(rng, nrec, lmin=0, lmax=90, final_newline=True, plus_names=False):
    """reads of every length class the piece copy distinguishes (0 .. lmax), names of 0 .. 70 bytes, '@' / '+' as the
    first quality"""
    out = []
    for i in range(nrec):
        n = rng.randint(0, 70) if rng.random() < 0.9 else 0
        name = "".join(rng.choice("abcXYZ019_:/.# \t") for _ in range(n))
        L = rng.randint(lmin, lmax)
        seq = "".join(rng.choice("ACGTN") for _ in range(L))
        qual = "".join(chr(rng.randint(33, 126)) for _ in range(L))
        if L and rng.random() < 0.3:
            qual = rng.choice("@+") + qual[1:]
        plus = "+" + (name if plus_names and rng.random() < 0.5 else "")
        out.append("@%s\n%s\n%s\n%s\n" % (name, seq, plus, qual))
    s = "".join(out)
    if not final_newline:
        s = s[:-1]
    return s.encode()


def run(data, region):
    return bsk.Subseq(bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(data)]), _Opts({"Region": region}))


REGIONS = ["1:50", "1:12", "5:-5", "-30:-1", "17:17", "1:-1", "40:20", "-200:-100", "3:1000", "-7:-3", "16:31", "2:16"]


@pytest.mark.parametrize("seed", range(3))
@pytest.mark.parametrize("region", REGIONS)
def test_subseq_pass_equals_oracle_and_table_path(seed, region, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "1024")
    rng = random.Random(9100 + 31 * seed + REGIONS.index(region))
    data = fastq(rng, 1500, final_newline=seed % 2 == 0, plus_names=seed == 1)
    want = oracle.subseq(data, True, json.dumps({"Region": region}))
    got = run(data, region)
    assert got == want
    monkeypatch.setenv("BSK_SUBSEQ", "table")
    assert run(data, region) == want


@pytest.mark.parametrize("region", ["1:50", "-60:-1", "100:260"])
def test_subseq_pass_lines_longer_than_the_carry(region, monkeypatch):
    # lines of 300 .. 1500 bytes: pieces that begin more than 512 bytes before the tile of their newline come from global
    # memory, pieces cut by the end of the shard byte by byte
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(9200 + len(region))
    data = fastq(rng, 400, lmin=300, lmax=1500, final_newline=False)
    want = oracle.subseq(data, True, json.dumps({"Region": region}))
    assert run(data, region) == want


def test_subseq_slice_overflow_takes_the_table_path(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "1024")
    rng = random.Random(9300)
    data = fastq(rng, 3000)
    want = oracle.subseq(data, True, '{"Region": "1:50"}')
    monkeypatch.setenv("BSK_SUBSEQ_SCALE", "0.02")   # slices far too small: ERR_CAPACITY -> fallback, same text
    assert run(data, "1:50") == want


def test_subseq_sample_underestimates_later_records(monkeypatch):
    # the head of the shard (what the slice size is estimated from) has reads of 4 bases, the rest 80: the slices overflow
    # and the record-table path answers
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    recs = ["@s%d\nACGT\n+\nIIII\n" % i for i in range(30000)]
    recs += ["@r%d\n%s\n+\n%s\n" % (i, "ACGTACGTAC" * 8, "IIIIIHHHHH" * 8) for i in range(30000)]
    data = "".join(recs).encode()
    want = oracle.subseq(data, True, '{"Region": "1:60"}')
    assert run(data, "1:60") == want


def test_subseq_pass_reports_format_errors():
    bad = b"@a\nACGT\n+\nIII\n@b\nAC\n+\nII\n"          # len(seq) != len(qual)
    with pytest.raises(Exception):
        run(bad, "1:2")
    bad2 = b"@a\nACGT\n-\nIIII\n"                         # third line does not start with '+'
    with pytest.raises(Exception):
        run(bad2, "1:2")


def test_subseq_long_reads_stay_with_the_table_path():
    rng = random.Random(9400)
    data = fastq(rng, 40, lmin=3000, lmax=9000)
    want = oracle.subseq(data, True, '{"Region": "-4000:-1"}')
    assert run(data, "-4000:-1") == want


def test_subseq_c2_layout_record_count_and_prefix():
    import ctypes as C
    import torch
    from bigseqkit_amd import _lib
    rb, nrec = 317, 200000
    t = torch.empty(rb * nrec, dtype=torch.uint8, device="cuda")
    assert _lib.lib.bsk_synth_device(0, 42, 0, 0, C.c_void_p(t.data_ptr()), rb * nrec, 0, None) == 0
    got = bsk.Subseq(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), _Opts({"Region": "1:50"}))
    head = bytes(t[:rb * 20000].cpu().numpy().tobytes())
    want = oracle.subseq(head, True, '{"Region": "1:50"}')
    assert got[:len(want)] == want
    assert len(want) % 20000 == 0 and len(got) == (len(want) // 20000) * nrec   # (every record of the layout has one size)
    assert got.count(b"\n") == 4 * nrec
