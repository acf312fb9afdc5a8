"""Parity of the record table and of `seq` (SeqTransform) against the CPU oracle, through the C ABI."""
import json
import random

import pytest

import oracle
import seqgen
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib

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


def check_index(data, fastq):
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    got = bsk.build_index(bsk.SeqFrame(fmt, [dev(data)]))
    spans = oracle.record_spans(data, fastq)
    assert len(got) == len(spans)
    for (st, hl, sl, ax), (s0, ln) in zip(got, spans):
        assert st == s0
        el = data[s0:s0 + ln]
        lines = el.split(b"\n")
        assert hl == len(lines[0])
        if fastq:
            assert sl == len(lines[1]) and ax == len(lines[2])
        else:
            assert sl == sum(len(x) for x in lines[1:])
            body = s0 + len(lines[0]) + 1
            end = min(len(data), s0 + ln + 1)  # the element has lost its final newline
            region = data[body:end]
            if region.endswith(b"\n"):          # blank lines at EOF are not part of the shard
                region = region.rstrip(b"\n") + b"\n"
            assert ax == len(region), (s0, ln, ax, len(region))


@pytest.mark.parametrize("seed", range(4))
def test_record_table_fastq(seed, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", str([256, 1000, 4096, 65536][seed % 4]))
    rng = random.Random(seed)
    check_index(seqgen.random_fastq(rng, 2000, 0, [40, 300, 2500][seed % 3], final_newline=seed % 2 == 0,
                                    trailing_blank=seed % 3), True)


@pytest.mark.parametrize("seed", range(4))
def test_record_table_fasta(seed, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", str([256, 1000, 4096, 65536][seed % 4]))
    rng = random.Random(50 + seed)
    check_index(seqgen.random_fasta(rng, 1200, 0, [100, 1500, 9000][seed % 3], width=[60, 70, 0, 13][seed % 4],
                                    final_newline=seed % 2 == 0, trailing_blank=seed % 3, gt_in_header=True), False)


def test_record_table_fasta_regression(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "2048")
    rng = random.Random(2001)
    data = seqgen.random_fasta(rng, 300, 0, 1200, width=60, final_newline=False, alphabet="ACGTNacgtu",
                               gt_in_header=True)
    check_index(data, False)


def check_seq(data, fastq, opts, on_device=True):
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    want = oracle.seq(data, fastq, json.dumps(opts))
    got = bsk.Seq(bsk.SeqFrame(fmt, [dev(data) if on_device else data]), _Opts(opts))
    assert got == want, (opts, got[:300], want[:300])


FQ_OPTS = [
    {},
    {"Name": True},
    {"Name": True, "OnlyId": True},
    {"Seq": True},
    {"Qual": True},
    {"Name": True, "Seq": True},
    {"Reverse": True},
    {"Reverse": True, "Complement": True},
    {"Seq": True, "Reverse": True, "Complement": True, "LowerCase": True},
    {"Qual": True, "Reverse": True},
    {"Dna2rna": True, "UpperCase": True},
    {"RemoveGaps": True},
    {"RemoveGaps": True, "Reverse": True, "GapLetters": "-.N"},
    {"MinLen": 50},
    {"MaxLen": 80, "MinLen": 10},
    {"MinQual": 30.5},
    {"MaxQual": 33, "MinQual": 20},
    {"OnlyId": True, "Config": {"LineWidth": 7}},
]


@pytest.mark.parametrize("i", range(len(FQ_OPTS)))
def test_seq_fastq_options(i, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "2048")
    rng = random.Random(1000 + i)
    data = seqgen.random_fastq(rng, 700, 0, 160, final_newline=i % 2 == 0, alphabet="ACGTNacgtRY")
    check_seq(data, True, FQ_OPTS[i])


FA_OPTS = [
    {},
    {"Config": {"LineWidth": 0}},
    {"Config": {"LineWidth": 25}},
    {"Name": True},
    {"Name": True, "OnlyId": True},
    {"Seq": True},
    {"Seq": True, "Reverse": True, "Complement": True},
    {"Reverse": True, "Config": {"LineWidth": 80}},
    {"RemoveGaps": True, "UpperCase": True},
    {"Rna2dna": True},
    {"MinLen": 100, "MaxLen": 900},
    {"MinQual": 10},
]


@pytest.mark.parametrize("width", [60, 0])
@pytest.mark.parametrize("i", range(len(FA_OPTS)))
def test_seq_fasta_options(i, width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "2048")
    rng = random.Random(2000 + i)
    data = seqgen.random_fasta(rng, 300, 0, 1200, width=width, final_newline=i % 2 == 0, alphabet="ACGTNacgtu",
                               gt_in_header=True)
    check_seq(data, False, FA_OPTS[i])


def test_seq_id_rules_and_ncbi():
    fq = (b"@id1 desc here\nAC\n+\nII\n@id2\ttab desc\nAC\n+\nII\n@ lead\nAC\n+\nII\n@nospace\nA\n+\nI\n"
          b"@gi|110645304|ref|NC_002516.2| Pseudomonas\nACGT\n+\nIIII\n")
    check_seq(fq, True, {"Name": True, "OnlyId": True})
    check_seq(fq, True, {"Name": True, "OnlyId": True, "Config": {"IDNCBI": True}})
    check_seq(fq, True, {"OnlyId": True})


def test_seq_errors():
    fa = b">a\nACGT\n"
    with pytest.raises(bsk.BskError, match="FASTA format has no quality"):
        bsk.Seq(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(fa)]), _Opts({"Qual": True}))
    with pytest.raises(oracle.OracleError, match="FASTA format has no quality"):
        oracle.seq(fa, False, '{"Qual": true}')
    # validation switched on by -t dna: 'X' is not a DNA letter
    bad = b">a\nACGTXX\n"
    with pytest.raises(bsk.BskError, match="invalid"):
        bsk.Seq(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(bad)]), _Opts({"Config": {"SeqType": "dna"}}))
    with pytest.raises(oracle.OracleError, match="invalid"):
        oracle.seq(bad, False, '{"Config": {"SeqType": "dna"}}')
    for opts, msg in [({"LowerCase": True, "UpperCase": True}, "could not give both flags"),
                      ({"MinLen": 10, "MaxLen": 5}, "should be >="),
                      ({"MinQual": 30, "MaxQual": 20}, "should be <=")]:
        with pytest.raises(bsk.BskError, match=msg):
            bsk.Operator("SeqTransform", json.dumps(opts), -1)
        with pytest.raises(oracle.OracleError, match=msg):
            oracle.seq(fa, False, json.dumps(opts))


def test_seq_host_shard_and_empty():
    rng = random.Random(77)
    data = seqgen.random_fastq(rng, 300, 0, 100)
    check_seq(data, True, {"Name": True}, on_device=False)
    check_seq(b"", True, {})
    check_seq(b"", False, {"Name": True})


def test_seq_n_on_synthetic_c2_layout():
    """BASELINE C2 `seq -n`: every record contributes its 12-byte name; checked at a size the oracle
    can still do (prefix) and by construction at 2 GB."""
    import ctypes as C
    import torch
    rb, nrec = 317, 6_000_000
    t = torch.empty(rb * nrec, dtype=torch.uint8, device="cuda")
    assert _lib.lib.bsk_synth_device(0, 42, 0, 0, C.c_void_p(t.data_ptr()), rb * nrec, 0, None) == 0
    got = bsk.Seq(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), bsk.SeqKitSeqOptions().Name(True))
    assert len(got) == 12 * nrec
    assert got[:24] == b"S0000000000\nS0000000001\n"
    assert got[-12:] == b"S%010d\n" % (nrec - 1)
    head = bytes(t[:rb * 20000].cpu().numpy().tobytes())
    assert got[:12 * 20000] == oracle.seq(head, True, '{"Name": true}')


def test_fasta_layout_from_index_pass_equals_separate_classification(monkeypatch):
    """RecordTable::text_w is produced by the index pass (line lengths seen by the event sink); BSK_TEXT=classify
    keeps the older separate pass over the line ends.  Same operator output either way, on every kind of wrapping."""
    import json
    import random
    import oracle
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(123)
    recs = []
    for k in range(400):
        L = rng.randint(0, 900)
        s = "".join(rng.choice("ACGT") for _ in range(L))
        kind = k % 6
        if kind == 0:
            lines = [s[j:j + 60] for j in range(0, L, 60)]
        elif kind == 1:
            lines = [s] if L else []
        elif kind == 2:
            lines = [s[j:j + 17] for j in range(0, L, 17)]
        elif kind == 3:
            lines = [s[j:j + 11] for j in range(0, L, 11)]       # narrower than 16: linearised
        elif kind == 4:                                            # one line too long / too short somewhere
            lines = [s[j:j + 60] for j in range(0, L, 60)]
            if len(lines) > 2:
                i = rng.randrange(len(lines) - 1)
                lines[i], lines[-1] = lines[i][:-3], lines[-1] + lines[i][-3:]
        else:                                                      # last line longer than the others
            lines = [s[j:j + 40] for j in range(0, max(0, L - 70), 40)] + ([s[max(0, L - 70) // 40 * 40:]] if L else [])
        recs.append(f">s{k} d\n" + "".join(l + "\n" for l in lines))
    data = "".join(recs).encode()
    opts = {"Reverse": True, "Complement": True, "Config": {"LineWidth": 50}}
    want = oracle.seq(data, False, json.dumps(opts))
    t = dev(data)
    for mode in ("", "classify"):
        monkeypatch.setenv("BSK_TEXT", mode)
        assert bsk.Seq(bsk.SeqFrame(bsk.FORMAT_FASTA, [t]), _Opts(opts)) == want, mode


def test_records_much_longer_than_a_range_through_every_operator(monkeypatch):
    """FASTA ranges begin on line starts: a record spans many ranges and k_index_stitch completes its table entry
    (bases, region, line layout).  Every operator must see exactly the oracle's records."""
    import json
    import random
    import oracle
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    monkeypatch.setenv("BSK_LONG_BYTES", "30000")   # records above 30 kB of output take the block-per-chunk emit
    rng = random.Random(321)
    recs = []
    specs = [(150_000, 60), (10, 60), (70_000, -1), (0, 60), (33_333, 11), (90_001, 80), (4096, 4096), (50_000, 16),
             (25_000, -2), (61, 60)]
    for k, (L, w) in enumerate(specs):
        s = "".join(rng.choice("ACGT") for _ in range(L))
        if w == -1:      # irregular wrapping
            lines, j = [], 0
            while j < L:
                ww = rng.randint(1, 200)
                lines.append(s[j:j + ww])
                j += ww
        elif w == -2:    # regular except one short line in the middle
            lines = [s[j:j + 60] for j in range(0, L, 60)]
            lines[len(lines) // 2] = lines[len(lines) // 2][:-7]
        else:
            lines = [s[j:j + w] for j in range(0, L, w)]
        recs.append(f">chr{k} test record\n" + "".join(l + "\n" for l in lines))
    data = "".join(recs).encode()
    t = dev(data)
    fr = lambda: bsk.SeqFrame(bsk.FORMAT_FASTA, [t])

    class _Opts:  # the option object the operators mutate (Grep forces Count off)
        def __init__(self, d):
            self.d = dict(d)
            self._v = self.d

        def to_json(self):
            return json.dumps(self.d)

    for mode in ("", "classify"):
        monkeypatch.setenv("BSK_TEXT", mode)
        o = {"Reverse": True, "Complement": True, "Config": {"LineWidth": 50}}
        assert bsk.Seq(fr(), _Opts(o)) == oracle.seq(data, False, json.dumps(o)), mode
    monkeypatch.setenv("BSK_TEXT", "")
    o = {}
    assert bsk.Seq(fr(), _Opts(o)) == oracle.seq(data, False, json.dumps(o))
    o = {"Region": "1000:-1000"}
    assert bsk.Subseq(fr(), _Opts(o)) == oracle.subseq(data, False, json.dumps(o))
    o = {"Pattern": ["ACGTACGTAC", "GGGGGGGGGGGG"], "BySeq": True}
    assert bsk.Grep(fr(), _Opts(o)) == oracle.grep(data, False, json.dumps(o))
    o = {"Pattern": ["ACGTACGT"]}
    assert bsk.Locate(fr(), _Opts(o)) == oracle.locate(data, False, json.dumps(o))
    o = {"Frame": ["6"], "AllowUnknownCodon": True}
    assert bsk.Translate(fr(), _Opts(o)) == oracle.translate(data, False, json.dumps(o))
    o = {"BySeq": True}
    assert bsk.RmDup(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data + data)]), _Opts(o)) == oracle.rmdup(data + data, False, json.dumps(o))


# ---------------------------------------------------------------- FASTA re-wrapped / mapped 16 output bytes per lane (k_seq_emit)
REWRAP_XFORMS = [{}, {"Reverse": True}, {"Complement": True}, {"Reverse": True, "Complement": True}, {"UpperCase": True, "Dna2rna": True},
                 {"Seq": True}, {"Seq": True, "Reverse": True, "Complement": True}]


@pytest.mark.parametrize("src_w", [60, 16, 17, 31, 70, 0, 9])
def test_seq_fasta_rewrap_matrix(src_w, monkeypatch):
    """source width x output width x transform: the output-driven path (source and output lines of >= 16 bases), the
    source-driven one (narrow lines) and the byte path (records under 16 bytes of text) against the oracle; lengths around
    the multiples of both widths, records at the very end of the shard (no 32-byte load past it)"""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "2048")
    rng = random.Random(4300 + src_w)
    recs = []
    lens = list(range(0, 40)) + [59, 60, 61, 69, 70, 71, 119, 120, 121, 139, 140, 141, 419, 420, 421] + [rng.randint(1, 1500) for _ in range(120)]
    rng.shuffle(lens)
    for k, L in enumerate(lens):
        s = "".join(rng.choice("ACGTNacgtu") for _ in range(L))
        body = s + "\n" if src_w == 0 else "".join(s[j:j + src_w] + "\n" for j in range(0, L, src_w))
        if L == 0:
            body = "\n" if k % 2 else ""
        recs.append(">r%d some words\n%s" % (k, body))
    data = "".join(recs).encode()
    for out_w in (0, 16, 17, 60, 61, 70, 100, 7):
        for x in REWRAP_XFORMS:
            o = dict(x, Config={"LineWidth": out_w})
            check_seq(data, False, o)
    for region in ("3:-3", "17:200", "-100:-1", "61:61"):
        for out_w in (0, 60, 33):
            o = {"Region": region, "Config": {"LineWidth": out_w}}
            assert bsk.Subseq(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)]), _Opts(o)) == oracle.subseq(data, False, json.dumps(o))


def test_seq_fasta_rewrap_long_records(monkeypatch):
    """records above BSK_LONG_BYTES are written by whole blocks, 64 KiB of output each: the 16-byte steps at the slice borders"""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    monkeypatch.setenv("BSK_LONG_BYTES", "20000")
    rng = random.Random(4400)
    recs = []
    for k, (L, w) in enumerate([(200_001, 60), (131_072, 64), (65_535, 0), (70_000, 17), (19_000, 60), (66_000, 100)]):
        s = "".join(rng.choice("ACGTacgtn") for _ in range(L))
        body = s + "\n" if w == 0 else "".join(s[j:j + w] + "\n" for j in range(0, L, w))
        recs.append(">long%d\n%s" % (k, body))
    data = "".join(recs).encode()
    for out_w in (0, 60, 70, 64, 16):
        for x in ({}, {"Reverse": True, "Complement": True}, {"LowerCase": True}):
            check_seq(data, False, dict(x, Config={"LineWidth": out_w}))
    o = {"Region": "1001:-1001", "Config": {"LineWidth": 80}}
    assert bsk.Subseq(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)]), _Opts(o)) == oracle.subseq(data, False, json.dumps(o))


def test_long_read_fastq_with_small_ranges(monkeypatch):
    """reads of 3-40 kb on one line, ranges of a few KiB: every range boundary falls inside a line that is longer than the
    range, the anchor search (k_prep, one wave per boundary) walks several such lines; all commands against the oracle"""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(9100)
    recs = []
    for i in range(120):
        L = rng.choice([0, 1, 50, 3000, 8191, 8192, 8193, 20000, 40000])
        s = "".join(rng.choice("ACGT") for _ in range(L))
        q = "".join(chr(rng.randint(33, 100)) for _ in range(L))
        if L and i % 5 == 0:
            q = "@" + q[1:]          # a quality line that looks like a header
        if L > 1 and i % 7 == 0:
            q = q[0] + "+" + q[2:]
        recs.append("@long%d ch=%d\n%s\n+\n%s\n" % (i, i, s, q))
    for final_newline in (True, False):
        data = "".join(recs).encode()
        if not final_newline:
            data = data[:-1]
        t = dev(data)
        fr = lambda: bsk.SeqFrame(bsk.FORMAT_FASTQ, [t])
        so = bsk.SeqKitStatsOptions()
        so.All(True)
        so.Tabular(True)
        assert bsk.StatsString("input0", "N/A", fr(), so) == oracle.stats_string(data, True, '{"All": true, "Tabular": true}')
        check_seq(data, True, {})
        check_seq(data, True, {"Name": True})
        check_seq(data, True, {"Reverse": True, "Complement": True, "MinLen": 100})
        o = {"Pattern": ["ACGTTGCA"], "BySeq": True}
        g = bsk.SeqKitGrepOptions()
        g.Pattern(["ACGTTGCA"])
        g.BySeq(True)
        assert bsk.Grep(fr(), g) == oracle.grep(data, True, json.dumps(dict(o, Count=False)))
        o = {"Region": "11:-11"}
        assert bsk.Subseq(fr(), _Opts(o)) == oracle.subseq(data, True, json.dumps(o))
        assert bsk.RmDup(fr(), _Opts({"BySeq": True})) == oracle.rmdup(data, True, '{"BySeq": true}')
