"""`stats` on FASTA, round 5: the streaming pass that publishes only the newlines the sink acts on (csrc/stream_fasta2_dev.hpp:
the one in front of a '>' and the one behind a header line; the other newlines are counted) against the pass that
publishes every newline (`stats_fasta=events`) and against the oracle (SeqParser.Read + Stats.Call,
/root/reference/bigseqkit-lib/helper.go:271-283, stats.go:88) -- on the layouts that stress it: pieces queued across tiles
and ranges, lines of a few bytes (every newline is published: the event ring fills inside a round), empty records,
'>' inside header and at line ends, tabs and carriage returns (control bytes that are no newlines), no final newline,
ranges of one tile, lines longer than a range."""
import json
import random
import zlib

import pytest

import oracle
import seqgen
import bigseqkit_amd as bsk

pytestmark = pytest.mark.gpu


class _Opts:
    def __init__(self, d):
        self.d = d

    def to_json(self):
        return json.dumps(self.d)


def gpu_map(data, opts):
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    m, op = bsk.stats_map(bsk.SeqFrame(bsk.FORMAT_FASTA, [t]), _Opts(opts))
    op.close()
    return m


# the default row, and `-a`: the gap letters counted over ALL bytes with the header lines taken off -- with the default
# letters (a blank: in every header of the generators below), with bases, '>' and a tab as gap letters (letters that fill
# sequences, begin headers, sit below 0x20), and with a letter above 126 (no "all sixteen above" shortcut)
OPTS = [{}, {"All": True}, {"All": True, "GapLetters": "ACg \t>"}, {"All": True, "GapLetters": "-\x7f"}]


def both(data, monkeypatch, opts_list=OPTS):
    for opts in opts_list:
        want = oracle.stats_map(data, False, json.dumps(opts))
        assert gpu_map(data, opts) == want, opts
        monkeypatch.setenv("BSK_STATS_FASTA", "events")
        assert gpu_map(data, opts) == want, opts
        monkeypatch.delenv("BSK_STATS_FASTA")


def fasta(rng, nrec, lens, width, head=lambda i, rng: b"r%d some text" % i, final_newline=True):
    out = []
    for i in range(nrec):
        L = lens(rng)
        s = bytes(rng.choice(b"ACGTN") for _ in range(L))
        body = b"".join(s[j:j + width] + b"\n" for j in range(0, L, width)) if width else (s + b"\n" if L else b"")
        out.append(b">" + head(i, rng) + b"\n" + body)
    data = b"".join(out)
    return data if final_newline else data[:-1]


HAND = [
    b">a\nACGT\n",
    b">a\nACGT",
    b">a\n>b\n>c\nA\n",                      # empty records: a header line that is closed by the next header
    b">a\n",
    b">a",
    b">a>b desc>\nAC\nGT\n>c\tx\ty\nA\r\nC\r\n",   # '>' inside a header, tabs, carriage returns (data, as the reference reads them)
    b">x\n" + b"A" * 15 + b"\n>y\n" + b"C" * 16 + b"\n>z\n" + b"G" * 17 + b"\n",  # newlines around a 16-byte piece end
    b">x\n" + (b"ACGTACGTAC" * 6 + b"\n") * 200,
]


@pytest.mark.parametrize("k", range(len(HAND)))
def test_hand_cases(k, monkeypatch):
    both(HAND[k], monkeypatch)


@pytest.mark.parametrize("min_range", [None, "4096", "65536"])
@pytest.mark.parametrize("shape", ["wrapped60", "wrapped70", "oneline", "tiny_lines", "mixed", "empty_records", "long_headers"])
def test_layouts(shape, min_range, monkeypatch):
    if min_range:
        monkeypatch.setenv("BSK_MIN_RANGE_BYTES", min_range)
    rng = random.Random(zlib.crc32(repr((shape, min_range)).encode()) & 0xFFFF)  # (str hashes differ from process to process)
    if shape == "wrapped60":
        data = fasta(rng, 700, lambda r: r.randint(0, 3000), 60)
    elif shape == "wrapped70":
        data = fasta(rng, 300, lambda r: r.choice((1000, 5001, 70, 69, 71, 140)), 70, final_newline=False)
    elif shape == "oneline":
        data = fasta(rng, 500, lambda r: r.randint(1, 2500), 0)
    elif shape == "tiny_lines":
        data = fasta(rng, 4000, lambda r: r.randint(0, 9), 2, head=lambda i, r: b"%d" % (i % 10))
    elif shape == "mixed":
        data = b"".join(fasta(rng, 40, lambda r: r.randint(0, 4000), w, head=lambda i, r: b"m%d\tx>y" % i) for w in (1, 3, 16, 15, 17, 60, 200, 0))
    elif shape == "empty_records":
        data = fasta(rng, 3000, lambda r: 0 if r.random() < 0.7 else r.randint(1, 100), 60)
    else:
        data = fasta(rng, 300, lambda r: r.randint(50, 1500), 60, head=lambda i, r: b"h%d " % i + bytes(r.choice(b"abc >\t|=") for _ in range(r.randint(0, 700))))
    both(data, monkeypatch)


def test_lines_longer_than_a_range(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(9)
    parts = []
    for i, L in enumerate((30000, 5, 70000, 4096, 4095, 12289, 100000)):
        parts.append(b">chr%d\n" % i + bytes(rng.choice(b"ACGT") for _ in range(L)) + b"\n")
    both(b"".join(parts), monkeypatch)
    both(b"".join(parts)[:-1], monkeypatch)
    # ... and a HEADER line longer than a range (its middle is counted by no one: the chunks' own ranges count it, the stitch
    # kernel adds those counts for sequence lines only, and the header-end event leaves the same bytes out)
    long_head = b">h " + bytes(rng.choice(b"ab -.c") for _ in range(50000)) + b"\nAC-GT\n>next one\nA.C\n"
    both(b"".join(parts[:2]) + long_head + parts[3], monkeypatch)


def test_synthetic_layouts_of_the_bench(monkeypatch):
    """C1 (1 kb records, 60 columns) and C4's input (5 kb CDS records): 40 MB each, exact against the oracle"""
    import ctypes as C
    import torch
    from bigseqkit_amd._lib import lib
    for kind in (1, 2):
        rb = lib.bsk_synth_record_bytes(kind)
        n = rb * (40_000_000 // rb) - 7
        t = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert lib.bsk_synth_device(kind, 42, 0, 0, C.c_void_p(t.data_ptr()), n, 0, None) == 0
        torch.cuda.synchronize()
        both(bytes(t.cpu().numpy().tobytes()), monkeypatch, OPTS[:2])
