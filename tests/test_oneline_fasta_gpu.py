"""Lines longer than a range's nominal chunk -- chromosomes on ONE line, a header of 100 kB: k_prep lets every boundary
search only its own chunk, boundaries without a line start take the next anchor (empty ranges), and the streaming kernels
do not read the newline-free middle of such a line (stream_core_dev.hpp skip_from).  Round 2 gave the whole line to one
wave (390 ms per 250 MB in every command).  Everything must stay byte for byte the oracle's: stats (default row: skip;
-a: every byte), the record table behind seq / subseq / grep / locate / faidx / rmdup / translate
(/root/reference/bigseqkit-lib/helper.go:219-250: a record's sequence is the concatenation of its lines, however long)."""
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


def genome(rng, sizes, wrap=0, long_header=0):
    out = []
    for k, n in enumerate(sizes):
        s = "".join(rng.choice("ACGT") for _ in range(n))
        if n > 50000:  # (random.choice per base is slow: build long ones from a block)
            block = "".join(rng.choice("ACGTN") for _ in range(4099))
            s = (block * (n // 4099 + 1))[:n]
        head = ">chr%d" % k + (" " + "d" * long_header if long_header and k == 1 else "")
        body = s + "\n" if not wrap else "".join(s[j:j + wrap] + "\n" for j in range(0, len(s), wrap))
        out.append(head + "\n" + body)
    return "".join(out).encode()


CASES = [([300_000, 17, 1_200_000, 0, 5, 700_001], 0, 0),
         ([2_000_000], 0, 0),
         ([100, 90_000, 100, 250_000], 0, 100_000),          # a header line longer than any chunk
         ([500_000, 60, 500_000], 60, 0)]                    # wrapped: nothing to skip, same answers


@pytest.mark.parametrize("min_range", ["4096", "65536"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_stats_of_lines_longer_than_a_range(case, min_range, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", min_range)
    sizes, wrap, lh = CASES[case]
    data = genome(random.Random(case), sizes, wrap, lh)
    if case == 0:
        data = data[:-1]                                   # no newline at the end of the file
    for o in ({"Tabular": True}, {"Tabular": True, "All": True}, {"Tabular": True, "All": True, "GapLetters": "N"}):
        want = oracle.stats_string(data, False, json.dumps(o), name="input0")
        got = bsk.StatsString("input0", "N/A", bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)]), _Opts(o))
        assert got == want, o


@pytest.mark.parametrize("case", range(len(CASES)))
def test_record_table_operators_on_one_line_chromosomes(case, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    monkeypatch.setenv("BSK_LONG_BYTES", "100000")
    sizes, wrap, lh = CASES[case]
    data = genome(random.Random(10 + case), sizes, wrap, lh)
    fr = lambda: bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)])
    j = json.dumps
    assert bsk.Seq(fr(), _Opts({"Name": True})) == oracle.seq(data, False, j({"Name": True}))
    assert bsk.Seq(fr(), _Opts({"Reverse": True, "Complement": True, "Config": {"LineWidth": 70}})) == \
        oracle.seq(data, False, j({"Reverse": True, "Complement": True, "Config": {"LineWidth": 70}}))
    assert bsk.Subseq(fr(), _Opts({"Region": "5:-7"})) == oracle.subseq(data, False, j({"Region": "5:-7"}))
    assert bsk.Faidx(fr(), _Opts({})) == oracle.faidx(data, False, "{}")
    first = b"".join(data.split(b">")[1].split(b"\n")[1:])          # the first record's sequence, line breaks removed
    pat = first[1000:1014].decode() if len(first) > 2000 else "ACGTAC"
    assert bsk.Grep(fr(), bsk.SeqKitGrepOptions().BySeq(True).Pattern([pat])) == oracle.grep(data, False, j({"BySeq": True, "Pattern": [pat]}))
    assert bsk.Locate(fr(), _Opts({"Pattern": [pat], "OnlyPositiveStrand": True})) == \
        oracle.locate(data, False, j({"Pattern": [pat], "OnlyPositiveStrand": True}))
    assert bsk.RmDup(fr(), _Opts({"BySeq": True})) == oracle.rmdup(data, False, j({"BySeq": True}))


def test_range_boundaries_inside_a_header_without_newline_for_several_ranges(monkeypatch):
    """a header that holds no newline for more than one range"""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(5)
    recs = [">a\nACGT\n", ">b " + "x>y " * 10000 + "\n" + "ACGTN" * 9000 + "\n", ">c\nAC\nGT\n", ">d " + "z" * 30000 + "\n", ">e\nA\n"]
    data = "".join(recs).encode()
    for o in ({"Tabular": True}, {"Tabular": True, "All": True}):
        assert bsk.StatsString("input0", "N/A", bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)]), _Opts(o)) == \
            oracle.stats_string(data, False, json.dumps(o), name="input0")
    assert bsk.Seq(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)]), _Opts({"Name": True, "OnlyId": True})) == \
        oracle.seq(data, False, json.dumps({"Name": True, "OnlyId": True}))
    assert bsk.Faidx(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)]), _Opts({})) == oracle.faidx(data, False, "{}")
