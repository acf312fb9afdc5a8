"""Parity of `grep` (exact patterns) and `subseq -r` against the CPU oracle, through the C ABI."""
import ctypes as C
import json
import random
import zlib

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
        self.d = dict(d)
        self._v = self.d

    def to_json(self):
        return json.dumps(self.d)


def frame(data, fastq, on_device=True):
    return bsk.SeqFrame(bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA, [dev(data) if on_device else data])


def check_grep(data, fastq, opts):
    want = oracle.grep(data, fastq, json.dumps(dict(opts, Count=False)))
    got = bsk.Grep(frame(data, fastq), _Opts(opts))
    assert got == want, (opts, len(got), len(want))
    wantc = int(oracle.grep(data, fastq, json.dumps(dict(opts, Count=True))))
    assert bsk.GrepCount(frame(data, fastq), _Opts(opts)) == wantc
    return wantc


MOTIF = "ACGTTGCAAGCT"


def planted_fastq(rng, n, L=150):
    out = []
    for i in range(n):
        s = [rng.choice("ACGT") for _ in range(L)]
        if i % 7 == 0:
            p = rng.randrange(L - 12 + 1)
            s[p:p + 12] = MOTIF
        elif i % 7 == 3:
            p = rng.randrange(L - 12 + 1)
            s[p:p + 12] = "AGCTTGCAACGT"  # reverse complement
        elif i % 7 == 5:
            s[-6:] = MOTIF[:6]
            s[:6] = MOTIF[6:]            # only a circular target contains the motif
        s = "".join(s)
        if i % 11 == 0:
            s = s.lower()
        out.append(f"@r{i} d{i % 5}\n{s}\n+\n{'I' * L}\n")
    return "".join(out).encode()


GREP_SEQ_OPTS = [
    {"BySeq": True, "Pattern": [MOTIF]},
    {"BySeq": True, "Pattern": [MOTIF], "OnlyPositiveStrand": True},
    {"BySeq": True, "Pattern": [MOTIF], "IgnoreCase": True},
    {"BySeq": True, "Pattern": [MOTIF], "InvertMatch": True},
    {"BySeq": True, "Pattern": [MOTIF], "Circular": True},
    {"BySeq": True, "Pattern": [MOTIF.lower(), "GGGGGGGG", "TTTTTTTTT"], "IgnoreCase": True},
    {"Pattern": [MOTIF], "Region": "1:40"},
    {"Pattern": [MOTIF], "Region": "-40:-1"},
    {"Pattern": [MOTIF], "Region": "20:-20", "OnlyPositiveStrand": True},
    {"BySeq": True, "Pattern": ["ACG"]},
    {"BySeq": True, "Pattern": ["A" * 200]},
    {"BySeq": True, "Pattern": ["ACGTN", "RYRY"], "Config": {"SeqType": "dna"}},
]


@pytest.mark.parametrize("i", range(len(GREP_SEQ_OPTS)))
def test_grep_by_seq_fastq(i, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(300 + i)
    data = planted_fastq(rng, 900)
    n = check_grep(data, True, GREP_SEQ_OPTS[i])
    if i == 0:
        assert n >= 900 // 7 * 2 * 9 // 11 - 5   # every 11th read is lower case


@pytest.mark.parametrize("width", [60, 0, 13])
@pytest.mark.parametrize("i", [0, 2, 3, 4, 6, 7])
def test_grep_by_seq_fasta(i, width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(400 + i)
    recs = []
    for k in range(250):
        L = rng.randint(0, 700)
        s = [rng.choice("ACGTacgt") for _ in range(L)]
        if L >= 12 and k % 3 == 0:
            p = rng.randrange(L - 11)
            s[p:p + 12] = MOTIF if k % 2 else "AGCTTGCAACGT"
        s = "".join(s)
        w = width if width else max(1, L)
        recs.append(f">s{k} x\n" + "".join(s[j:j + w] + "\n" for j in range(0, L, w)))
    data = "".join(recs).encode()
    check_grep(data, False, GREP_SEQ_OPTS[i])


def test_grep_irregularly_wrapped_fasta():
    rng = random.Random(5)
    recs = []
    for k in range(120):
        L = rng.randint(20, 400)
        s = [rng.choice("ACGT") for _ in range(L)]
        if k % 2 == 0:
            p = rng.randrange(L - 11)
            s[p:p + 12] = MOTIF
        s = "".join(s)
        lines, j = [], 0
        while j < L:
            w = rng.randint(1, 50)
            lines.append(s[j:j + w])
            j += w
        recs.append(f">s{k}\n" + "\n".join(lines) + "\n")
    data = "".join(recs).encode()
    for o in ({"BySeq": True, "Pattern": [MOTIF]}, {"BySeq": True, "Pattern": [MOTIF], "Circular": True},
              {"Pattern": [MOTIF], "Region": "5:-5"}):
        check_grep(data, False, o)


GREP_NAME_OPTS = [
    {"Pattern": ["r3", "r10", "r899", "nope"]},
    {"Pattern": ["r3", "R10"], "IgnoreCase": True},
    {"Pattern": ["r3 d3", "r10 d0", "r4"], "ByName": True},
    {"Pattern": ["r3"], "InvertMatch": True},
]


@pytest.mark.parametrize("i", range(len(GREP_NAME_OPTS)))
def test_grep_by_id_and_name(i):
    rng = random.Random(500 + i)
    data = planted_fastq(rng, 900, L=30)
    n = check_grep(data, True, GREP_NAME_OPTS[i])
    assert n > 0


def test_grep_option_errors():
    for opts, msg in [({}, "one of flags -p (--pattern) and -f (--pattern-file) needed"),
                      ({"Pattern": ["ACGT"], "Region": "0:5"}, "both start and end should not be 0"),
                      ({"Pattern": ["ACGT"], "Region": "-5:5"}, "when start < 0, end should not > 0"),
                      ({"Pattern": ["ACGT"], "Region": "abc"}, "invalid region: abc"),
                      ({"Pattern": ["AC!T"], "BySeq": True}, "illegal DNA/RNA/Protein sequence: AC!T")]:
        with pytest.raises(bsk.BskError) as e:
            bsk.Operator("Grep", json.dumps(opts), -1)
        assert msg in str(e.value)
        with pytest.raises(oracle.OracleError) as oe:
            oracle.grep(b"@a\nA\n+\nI\n", True, json.dumps(opts))
        assert msg in str(oe.value)


def test_grep_c3_synthetic_motif_count():
    """BASELINE C3 planting rule: motif on '+' when i%100==0, reverse complement when i%100==50."""
    import torch
    rb, nrec = 317, 3_000_000
    t = torch.empty(rb * nrec, dtype=torch.uint8, device="cuda")
    assert _lib.lib.bsk_synth_device(0, 42, _lib.SYNTH_FLAG_MOTIF, 0, C.c_void_p(t.data_ptr()), rb * nrec, 0, None) == 0
    o = {"BySeq": True, "Pattern": [MOTIF]}
    n = bsk.GrepCount(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), _Opts(o))
    assert nrec // 50 <= n <= nrec // 50 + 400          # 2 % planted + ~1.7e-5 * 2 background
    npos = bsk.GrepCount(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), _Opts(dict(o, OnlyPositiveStrand=True)))
    assert nrec // 100 <= npos <= nrec // 100 + 200
    head = bytes(t[:rb * 40000].cpu().numpy().tobytes())
    assert bsk.Grep(frame(head, True), _Opts(o)) == oracle.grep(head, True, json.dumps(o))


SUBSEQ_REGIONS = ["1:1", "2:4", "-4:-2", "-4:-1", "-1:-1", "2:-2", "1:-1", "1:12", "-12:-1", "50:60", "100:90", "-3:-9"]


def test_subseq_region_table_kat():
    # bigseqkit-cli/helper.go:348-361
    fa = b">s\nACGTNacgtn\n"
    want = {"1:1": "A", "2:4": "CGT", "-4:-2": "cgt", "-4:-1": "cgtn", "-1:-1": "n", "2:-2": "CGTNacgt",
            "1:-1": "ACGTNacgtn", "1:12": "ACGTNacgtn", "-12:-1": "ACGTNacgtn"}
    for r, w in want.items():
        got = bsk.Subseq(frame(fa, False), _Opts({"Region": r}))
        assert got == b">s\n" + w.encode() + b"\n"
        assert got == oracle.subseq(fa, False, json.dumps({"Region": r}))


@pytest.mark.parametrize("fastq", [True, False])
@pytest.mark.parametrize("r", SUBSEQ_REGIONS)
def test_subseq_region_random(r, fastq, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(zlib.crc32(repr(r).encode()) & 0xFFFF)  # (str hashes differ from process to process)
    data = seqgen.random_fastq(rng, 400, 0, 120) if fastq else seqgen.random_fasta(rng, 200, 0, 300, width=60)
    for lw in ([60] if fastq else [60, 0, 17]):
        o = {"Region": r, "Config": {"LineWidth": lw}}
        assert bsk.Subseq(frame(data, fastq), _Opts(o)) == oracle.subseq(data, fastq, json.dumps(o))


def test_subseq_option_errors():
    for opts, msg in [({}, "one of the options needed: -r/--region, --bed, --gtf"),
                      ({"Region": "1:5", "UpStream": 3}, "when flag -r (--region) given"),
                      ({"OnlyFlank": True, "Region": "1:2"}, "when flag -f (--only-flank) given")]:
        with pytest.raises(bsk.BskError) as e:
            bsk.Operator("SubseqTransform", json.dumps(opts), -1)
        assert msg in str(e.value)
        with pytest.raises(oracle.OracleError) as oe:
            oracle.subseq(b">a\nA\n", False, json.dumps(opts))
        assert msg in str(oe.value)


# ---------------------------------------------------------------- -d, -m, -f (class patterns, pattern files)
GREP_GEN_OPTS = [
    {"Pattern": ["ACGTTGCAAGCT"], "MaxMismatch": 1},
    {"Pattern": ["ACGTTGCAAGCT"], "MaxMismatch": 3, "OnlyPositiveStrand": True},
    {"Pattern": ["acgttgcaagct"], "MaxMismatch": 2, "IgnoreCase": True},
    {"Pattern": ["ACGTTGCAAGCT", "GGGGGGGGGG"], "MaxMismatch": 2, "InvertMatch": True},
    {"Pattern": ["ACGTTGCAAGCT"], "MaxMismatch": 1, "Circular": True},
    {"Pattern": ["ACGTTGCAAGCT"], "MaxMismatch": 2, "Region": "1:60"},
    {"Pattern": ["ACGTTGCAAGCT"], "MaxMismatch": 2, "Region": "-60:-1"},
    {"Pattern": ["ACGNTGCRAGYT"], "Degenerate": True},
    {"Pattern": ["acgnnnnaagct"], "Degenerate": True, "IgnoreCase": True},
    {"Pattern": ["acgnnnnaagct"], "Degenerate": True},                       # lower-case classes only
    {"Pattern": ["WSWSWSWSWS", "ACGNTGCRAGYT"], "Degenerate": True, "Circular": True},
    {"Pattern": ["ACGNTGCRAGYT"], "Degenerate": True, "Region": "10:-10", "OnlyPositiveStrand": True},
    {"Pattern": ["BBBBBBHHHHDDDDVVVV"], "Degenerate": True},
]


@pytest.mark.parametrize("i", range(len(GREP_GEN_OPTS)))
def test_grep_mismatch_and_degenerate_fastq(i, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(1300 + i)
    n = check_grep(planted_fastq(rng, 600), True, GREP_GEN_OPTS[i])
    assert n > 0


@pytest.mark.parametrize("width", [60, 0, 13, -1])
@pytest.mark.parametrize("i", [0, 2, 4, 6, 7, 8, 10])
def test_grep_mismatch_and_degenerate_fasta(i, width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(1400 + i)
    recs = []
    for k in range(200):
        L = rng.randint(0, 500)
        s = [rng.choice("ACGTacgtN") for _ in range(L)]
        if L >= 12 and k % 3 == 0:
            p = rng.randrange(L - 11)
            s[p:p + 12] = MOTIF if k % 2 else "AGCTTGCAACGT"
            if k % 4 == 0:
                s[p + 5] = "T" if s[p + 5] != "T" else "A"   # one mismatch
        s = "".join(s)
        if width < 0:  # irregular wrapping
            lines, j = [], 0
            while j < L:
                w = rng.randint(1, 40)
                lines.append(s[j:j + w])
                j += w
            recs.append(f">s{k} x\n" + "".join(l + "\n" for l in lines))
        else:
            w = width if width else max(1, L)
            recs.append(f">s{k} x\n" + "".join(s[j:j + w] + "\n" for j in range(0, L, w)))
    check_grep("".join(recs).encode(), False, GREP_GEN_OPTS[i])


def test_grep_protein_degenerate_with_seqtype_protein():
    prot = b">p1\nMKVLAAGIVDMEE\n>p2\nMKVLAAGIVNMQE\n>p3\nMKVLAAGIVAMAE\n"
    o = {"Pattern": ["VBMZE"], "Degenerate": True, "Config": {"SeqType": "protein"}}
    assert check_grep(prot, False, o) == 2
    assert check_grep(prot, False, {"Pattern": ["GIVXM"], "Degenerate": True, "Config": {"SeqType": "protein"}}) == 3


def test_grep_pattern_file_and_large_id_set(tmp_path):
    rng = random.Random(77)
    data = planted_fastq(rng, 3000, L=20)
    ids = [f"r{i}" for i in rng.sample(range(6000), 2500)] + ["", "r5", "r5"]
    f = tmp_path / "ids.txt"
    f.write_text("\r\n".join(ids) + "\n")
    n = check_grep(data, True, {"PatternFile": str(f)})
    assert 800 < n < 2000
    check_grep(data, True, {"PatternFile": str(f), "InvertMatch": True})
    names = tmp_path / "names.txt"
    names.write_text("".join(f"R{i} D{i % 5}\n" for i in range(0, 3000, 3)))
    assert check_grep(data, True, {"PatternFile": str(names), "ByName": True, "IgnoreCase": True}) == 1000
    seqs = tmp_path / "seqs.txt"
    seqs.write_text(MOTIF + "\nGGGGGGGGGGGG\n")
    check_grep(data, True, {"PatternFile": str(seqs), "BySeq": True})
    check_grep(planted_fastq(rng, 500), True, {"PatternFile": str(seqs), "MaxMismatch": 1})
    # many patterns given with -p take the same set path
    many = [f"r{i}" for i in range(0, 3000, 7)]
    assert check_grep(data, True, {"Pattern": many}) == len(many)


def test_grep_general_option_errors():
    for opts, msg in [({"Pattern": ["ACGT"], "MaxMismatch": 5}, "mismatch should be <= length of sequence: ACGT"),
                      ({"Pattern": ["ACGT"], "MaxMismatch": 1, "Degenerate": True}, "not allowed when giving flag -m"),
                      ({"PatternFile": "/nonexistent/ids.txt"}, "no such file or directory")]:
        with pytest.raises(bsk.BskError) as e:
            bsk.Operator("Grep", json.dumps(opts), -1)
        assert msg in str(e.value)
        with pytest.raises(oracle.OracleError) as oe:
            oracle.grep(b"@a\nA\n+\nI\n", True, json.dumps(opts))
        assert msg in str(oe.value)


# ---------------------------------------------------------------- subseq --bed / --gtf
def check_subseq(data, fastq, opts):
    want = oracle.subseq(data, fastq, json.dumps(opts))
    got = bsk.Subseq(frame(data, fastq), _Opts(opts))
    assert got == want, (opts, got[:300], want[:300])
    return got


def feature_files(tmp_path, rng, nrec, prefix):
    bed, gtf = [], []
    bed.append("track name=test\n")
    bed.append("#comment\n")
    for k in rng.sample(range(nrec + 20), nrec // 2):
        for _ in range(rng.randint(1, 2)):   # a second feature of the same name is never used
            st = rng.randint(0, 300)
            en = st + rng.randint(1, 200)
            strand = rng.choice("+-.")
            name = f"{prefix}{k}" if rng.random() < 0.7 else f"{prefix.upper()}{k}"
            cols = [name, str(st), str(en)]
            if rng.random() < 0.8:
                cols += [f"gene{k}", "0", strand]
            elif rng.random() < 0.5:
                cols += [f"only name {k}"]
            bed.append("\t".join(cols) + "\n")
            typ = rng.choice(["gene", "CDS", "exon"])
            gtf.append(f'{name}\tsrc\t{typ}\t{st + 1}\t{en}\t.\t{strand}\t.\tgene_id "G{k}"; transcript_id "T{k}.1";\n')
    b, g = tmp_path / "f.bed", tmp_path / "f.gtf"
    b.write_text("".join(bed))
    g.write_text("#!gtf\n" + "".join(gtf))
    return str(b), str(g)


FEATURE_FLAGS = [{}, {"UpStream": 5}, {"DownStream": 7}, {"UpStream": 3, "DownStream": 4},
                 {"UpStream": 6, "OnlyFlank": True}, {"DownStream": 6, "OnlyFlank": True}]


@pytest.mark.parametrize("width", [60, 0, 17, -1])
@pytest.mark.parametrize("fi", range(len(FEATURE_FLAGS)))
def test_subseq_bed_and_gtf_fasta(tmp_path, fi, width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(2100 + fi)
    recs = []
    for k in range(150):
        L = rng.randint(0, 600)
        s = "".join(rng.choice("ACGTacgtNRY") for _ in range(L))
        if width < 0:
            lines, j = [], 0
            while j < L:
                w = rng.randint(1, 40)
                lines.append(s[j:j + w])
                j += w
            recs.append(f">s{k} desc\n" + "".join(l + "\n" for l in lines))
        else:
            w = width if width else max(1, L)
            recs.append(f">s{k} desc\n" + "".join(s[j:j + w] + "\n" for j in range(0, L, w)))
    data = "".join(recs).encode()
    bed, gtf = feature_files(tmp_path, rng, 150, "s")
    flags = FEATURE_FLAGS[fi]
    got = check_subseq(data, False, dict(flags, Bed=bed))
    assert got.count(b">") > 30
    check_subseq(data, False, dict(flags, Gtf=gtf, GtfTag="gene_id"))
    check_subseq(data, False, dict(flags, Gtf=gtf, GtfTag="transcript_id", Feature=["cds", "Exon"]))
    check_subseq(data, False, dict(flags, Bed=bed, Chr=["s3", "s10", "S11", "s12"]))
    check_subseq(data, False, dict(flags, Gtf=gtf, Config={"LineWidth": 0}))


@pytest.mark.parametrize("fi", range(len(FEATURE_FLAGS)))
def test_subseq_bed_and_gtf_fastq(tmp_path, fi):
    rng = random.Random(2200 + fi)
    data = seqgen.random_fastq(rng, 300, 0, 400, alphabet="ACGTN")
    bed, gtf = feature_files(tmp_path, rng, 300, "r")
    got = check_subseq(data, True, dict(FEATURE_FLAGS[fi], Bed=bed))
    assert got.count(b"\n+\n") > 50
    check_subseq(data, True, dict(FEATURE_FLAGS[fi], Gtf=gtf, GtfTag="gene_id"))


def test_subseq_feature_edge_cases(tmp_path):
    fa = b">chr1 d\nAAAATTTTGGAAAACCCC\n>Chr2\nACGTTGCAAGCT\n>chr3\nACGT\n>chr4\n\n"
    bed = tmp_path / "a.bed"
    bed.write_text("browser position\nchr1\t0\t8\tfirst\t0\t+\nchr1\t3\t8\tnever\t0\t-\nCHR2\t100\t105\tbeyond\nchr3\t1\t3\nchr4\t0\t5\tx\t0\t-\n"
                   "chr9\t1\t3\nshort\t1\n")
    got = check_subseq(fa, False, {"Bed": str(bed)})
    assert got.startswith(b">chr1_1-8:+ first\nAAAATTTT\n") and b"never" not in got
    assert b">Chr2_101-105:. beyond\n\n" in got                                  # feature past the end: empty sequence
    check_subseq(fa, False, {"Bed": str(bed), "UpStream": 4, "OnlyFlank": True})   # flank of a feature at position 1 is empty
    check_subseq(fa, False, {"Bed": str(bed), "DownStream": 50})
    empty = tmp_path / "none.bed"
    empty.write_text("#nothing\n")
    assert check_subseq(fa, False, {"Bed": str(empty)}) == b""
    for text, msg in [("chr1\tx\t5\n", "chr1: bad start: x"), ("chr1\t5\t5\n", "chr1: start (5) must be <= end (5)"),
                      ("chr1\t1\t5\tn\t0\t*\n", "bad strand: *")]:
        bad = tmp_path / "bad.bed"
        bad.write_text(text)
        with pytest.raises(bsk.BskError) as e:
            bsk.Operator("SubseqTransform", json.dumps({"Bed": str(bad)}), -1)
        assert msg in str(e.value)
        with pytest.raises(oracle.OracleError) as oe:
            oracle.subseq(fa, False, json.dumps({"Bed": str(bad)}))
        assert msg in str(oe.value)
    with pytest.raises(bsk.BskError) as e:
        bsk.Operator("SubseqTransform", json.dumps({"Bed": str(bed), "Feature": ["gene"]}), -1)
    assert "when given flag -b (--bed), flag -f (--feature) is not allowed" in str(e.value)


# ---------------------------------------------------------------- grep -r (regular expressions on the device)
GREP_RE_OPTS = [
    {"Pattern": ["^r1[0-9]$"], "UseRegexp": True},
    {"Pattern": ["^r\\d+ d[03]$"], "UseRegexp": True, "ByName": True},
    {"Pattern": ["R1", "^r2\\d\\d$"], "UseRegexp": True, "IgnoreCase": True},
    {"Pattern": ["7$"], "UseRegexp": True, "InvertMatch": True},
    {"Pattern": ["ACGTT[GA]CAAGCT"], "UseRegexp": True, "BySeq": True},
    {"Pattern": ["ACGTTGCA{2}GCT"], "UseRegexp": True, "BySeq": True, "OnlyPositiveStrand": True},
    {"Pattern": ["^A.*T$"], "UseRegexp": True, "BySeq": True},
    {"Pattern": ["acgttg(ca|gg)agct"], "UseRegexp": True, "BySeq": True, "IgnoreCase": True},
    {"Pattern": ["GCAAGCT.*ACGTTG|TTTTTTTT"], "UseRegexp": True, "BySeq": True, "Circular": True},
    {"Pattern": ["^ACG"], "UseRegexp": True, "Region": "5:60"},
    {"Pattern": ["(AC){3,}G+T"], "UseRegexp": True, "Region": "-50:-1", "InvertMatch": True},
    {"Pattern": ["A[TU]G(?:.{3})+?[TU](?:AG|AA|GA)"], "UseRegexp": True, "BySeq": True},
]


# expressions the bit-parallel automaton does not take go to the thread-list matcher (k_grep_vm, round 4): word boundaries
GREP_VM_OPTS = [
    {"Pattern": [r"\br1\d\b"], "UseRegexp": True},
    {"Pattern": [r"\bd[03]\b"], "UseRegexp": True, "ByName": True},
    {"Pattern": [r"R1\B", r"^r2\d\d$"], "UseRegexp": True, "IgnoreCase": True},
    {"Pattern": [r"\b7$"], "UseRegexp": True, "ByName": True, "InvertMatch": True},
    {"Pattern": [r"\bACG", r"GCT\b"], "UseRegexp": True, "BySeq": True},
    {"Pattern": [r"acgttg(ca|gg)agct\B"], "UseRegexp": True, "BySeq": True, "IgnoreCase": True, "OnlyPositiveStrand": True},
    {"Pattern": [r"GCAAGCT.*\BACGTTG|TTTTTTTT\b"], "UseRegexp": True, "BySeq": True, "Circular": True},
    {"Pattern": [r"^ACG\B"], "UseRegexp": True, "Region": "5:60"},
    {"Pattern": [r"\bAC", r"\br1", r"T\b"], "UseRegexp": True, "BySeq": True, "DeleteMatched": True},
]


@pytest.mark.parametrize("i", range(len(GREP_VM_OPTS)))
def test_grep_regexp_word_boundaries(i, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(3350 + i)
    n = check_grep(planted_fastq(rng, 700), True, GREP_VM_OPTS[i])
    assert 0 <= n <= 700
    recs = []
    for k in range(150):
        L = rng.choice((0, 1, 7, 60, 61, 200))
        recs.append((b"r%d d%d" % (k, k % 7), bytes(rng.choice(b"ACGTacgtN") for _ in range(L))))
    for width in (60, 0):
        fa = b"".join(b">" + h + b"\n" + (b"\n".join(s[j:j + width] for j in range(0, len(s), width)) if width else s) + b"\n" for h, s in recs)
        check_grep(fa, False, GREP_VM_OPTS[i])


@pytest.mark.parametrize("i", range(len(GREP_RE_OPTS)))
def test_grep_regexp_fastq(i, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(3300 + i)
    n = check_grep(planted_fastq(rng, 700), True, GREP_RE_OPTS[i])
    assert 0 < n < 700


@pytest.mark.parametrize("width", [60, 0, -1])
@pytest.mark.parametrize("i", [4, 6, 7, 8, 9])
def test_grep_regexp_fasta(i, width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(3400 + i)
    recs = []
    for k in range(200):
        L = rng.randint(0, 400)
        s = [rng.choice("ACGTacgt") for _ in range(L)]
        if L >= 12 and k % 3 == 0:
            p = rng.randrange(L - 11)
            s[p:p + 12] = MOTIF if k % 2 else "AGCTTGCAACGT"
        s = "".join(s)
        if width < 0:
            lines, j = [], 0
            while j < L:
                w = rng.randint(1, 40)
                lines.append(s[j:j + w])
                j += w
            recs.append(f">s{k} x\n" + "".join(l + "\n" for l in lines))
        else:
            w = width if width else max(1, L)
            recs.append(f">s{k} x\n" + "".join(s[j:j + w] + "\n" for j in range(0, L, w)))
    check_grep("".join(recs).encode(), False, GREP_RE_OPTS[i])


def test_grep_regexp_errors_and_pattern_file(tmp_path):
    for opts, msg in [({"Pattern": ["a(b"], "UseRegexp": True}, "missing closing )"),
                      ({"Pattern": ["ACGT"], "UseRegexp": True, "Degenerate": True}, "could not give both flags -d")]:
        with pytest.raises(bsk.BskError) as e:
            bsk.Operator("Grep", json.dumps(opts), -1)
        assert msg in str(e.value)
    f = tmp_path / "re.txt"
    f.write_text("^r1\\d$\n^r2\\d\\d$\n")
    rng = random.Random(8)
    assert check_grep(planted_fastq(rng, 400, L=20), True, {"PatternFile": str(f), "UseRegexp": True}) == 110


# ---------------------------------------------------------------- --delete-matched (PARITY.md DEL)
def test_grep_delete_matched_keeps_the_first_record_of_every_pattern(tmp_path, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    fa = b">a 1\nACGT\n>b\nGG\n>a 2\nTT\n>c\nACGA\n>b x\nAA\n>A 3\nCC\n"
    def both(data, fastq, o):
        check_grep(data, fastq, o)   # bytes and count against the oracle
        return bsk.Grep(frame(data, fastq), _Opts(o))
    assert both(fa, False, {"Pattern": ["a", "b"], "DeleteMatched": True}) == b">a 1\nACGT\n>b\nGG\n"
    assert both(fa, False, {"Pattern": ["a"], "DeleteMatched": True, "IgnoreCase": True}) == b">a 1\nACGT\n"
    assert both(fa, False, {"Pattern": ["a 2", "A 3", "zz"], "ByName": True, "DeleteMatched": True}) == b">a 2\nTT\n>A 3\nCC\n"
    assert both(fa, False, {"Pattern": ["ACG"], "BySeq": True, "DeleteMatched": True}) == b">a 1\nACGT\n"
    assert both(fa, False, {"Pattern": ["^[ab]$"], "UseRegexp": True, "DeleteMatched": True}) == b">a 1\nACGT\n"
    assert both(fa, False, {"Pattern": ["ACN"], "Degenerate": True, "DeleteMatched": True}) == b">a 1\nACGT\n"
    # with -v it changes nothing (grep.go:463)
    assert both(fa, False, {"Pattern": ["a"], "DeleteMatched": True, "InvertMatch": True}) == b">b\nGG\n>c\nACGA\n>b x\nAA\n>A 3\nCC\n"
    # a large ID list against reads with repeated IDs
    rng = random.Random(6)
    ids = [f"r{rng.randrange(300)}" for _ in range(2000)]
    fq = "".join(f"@{i} n{k}\nACGT\n+\nIIII\n" for k, i in enumerate(ids)).encode()
    pf = tmp_path / "ids.txt"
    pf.write_text("".join(f"r{k}\n" for k in range(0, 300, 2)))
    got = both(fq, True, {"PatternFile": str(pf), "DeleteMatched": True})
    assert got.count(b"@") == len({i for i in ids if int(i[1:]) % 2 == 0})
    assert bsk.GrepCount(frame(fq, True), _Opts({"PatternFile": str(pf), "DeleteMatched": True})) == got.count(b"@")
    # several sequence patterns: records in file order, a record is a hit when a REMAINING pattern matches; that pattern (the
    # first one given, '+' strand before '-') is dropped (grep.go:463-511, PARITY.md DEL)
    assert both(fa, False, {"Pattern": ["AC", "GG"], "BySeq": True, "DeleteMatched": True}) == b">a 1\nACGT\n>b\nGG\n"
    assert both(fa, False, {"Pattern": ["AC", "CG"], "BySeq": True, "DeleteMatched": True}) == b">a 1\nACGT\n>c\nACGA\n"
    assert both(fa, False, {"Pattern": ["GG", "CC"], "BySeq": True, "DeleteMatched": True}) == b">b\nGG\n>A 3\nCC\n"
    assert both(fa, False, {"Pattern": ["GG", "CC"], "BySeq": True, "DeleteMatched": True, "OnlyPositiveStrand": True}) == b">b\nGG\n>A 3\nCC\n"
    assert both(fa, False, {"Pattern": ["^A", "G$", "T"], "UseRegexp": True, "BySeq": True, "DeleteMatched": True}) == \
        b">a 1\nACGT\n>b\nGG\n>a 2\nTT\n"
    assert both(fa, False, {"Pattern": ["ACN", "GGN"], "Degenerate": True, "DeleteMatched": True}) == b">a 1\nACGT\n"
    # with -m the reference never drops a pattern (grepBySeqMismatches, grep.go:255-365): a plain grep -m
    assert both(fa, False, {"Pattern": ["ACGT"], "MaxMismatch": 1, "DeleteMatched": True}) == b">a 1\nACGT\n>c\nACGA\n"
    with pytest.raises(bsk.BskError) as e:
        bsk.Grep(frame(fa, False), _Opts({"Pattern": ["A" * k for k in range(1, 258)], "BySeq": True, "DeleteMatched": True}))
    assert "more than 255" in str(e.value)


@pytest.mark.parametrize("npat", [16, 17, 31, 40, 120])
def test_grep_delete_matched_more_than_fifteen_patterns(npat, monkeypatch):
    """round 2 refused --delete-matched with more than 15 sequence / regexp patterns (one 32-bit hit word per record); the
    hit bits now sit in arrays of 15 patterns each and the greedy walk of grep.go:463-511 goes over all of them"""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(npat)
    recs = []
    for i in range(500):
        L = rng.randint(20, 70)
        recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(L)), "I" * L))
    data = "".join(recs).encode()
    pats = []
    while len(pats) < npat:
        p = "".join(rng.choice("ACGT") for _ in range(rng.randint(3, 6)))
        if p not in pats:
            pats.append(p)
    check_grep(data, True, {"Pattern": pats, "BySeq": True, "DeleteMatched": True})
    check_grep(data, True, {"Pattern": pats, "BySeq": True, "DeleteMatched": True, "OnlyPositiveStrand": True})
    check_grep(data, True, {"Pattern": [p[:2] + "[AG]" + p[2:] for p in pats[:20]], "UseRegexp": True, "BySeq": True, "DeleteMatched": True})


def test_grep_delete_matched_many_patterns_random(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(77)
    for trial in range(12):
        recs = []
        for i in range(300):
            L = rng.randint(20, 60)
            recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(L)), "I" * L))
        data = "".join(recs).encode()
        pats = ["".join(rng.choice("ACGT") for _ in range(rng.randint(3, 5))) for _ in range(rng.randint(2, 8))]
        o = {"Pattern": pats, "BySeq": True, "DeleteMatched": True}
        if trial % 3 == 1: o["OnlyPositiveStrand"] = True
        if trial % 3 == 2: o["IgnoreCase"] = True
        check_grep(data, True, o)
        check_grep(data, True, {"Pattern": [p[:2] + "[AC]" + p[2:] for p in pats[:4]], "UseRegexp": True, "BySeq": True, "DeleteMatched": True})


@pytest.mark.parametrize("width", [60, 17, 0])
def test_wrapped_fasta_searches_on_linear_copies_equal_the_text_views(width, monkeypatch):
    """grep -s / locate / rmdup -s on wrapped FASTA read linear copies of the records (ops_text.hip k_text_flatten);
    BSK_TEXT=view keeps the in-place views of round 1: same bytes, and both equal the oracle."""
    import seqgen
    rng = random.Random(8800 + width)
    data = seqgen.random_fasta(rng, 400, 0, 900, width=width)
    data += data[:len(data) // 3]   # duplicates for rmdup
    g = {"Pattern": ["ACGT", "TTGCA"], "BySeq": True}
    l = {"Pattern": ["ACGT"]}
    r = {"BySeq": True}
    want = (oracle.grep(data, False, json.dumps(g)), oracle.locate(data, False, json.dumps(l)), oracle.rmdup(data, False, json.dumps(r)))
    # class patterns (Shift-And on views reads byte by byte through the line map) and the position-reporting matcher
    gm = {"Pattern": ["ACGTTGCA"], "BySeq": True, "MaxMismatch": 2}
    gd = {"Pattern": ["ACNNTGCR"], "BySeq": True, "Degenerate": True, "Region": "5:-5"}
    lm = {"Pattern": ["ACGTTG"], "MaxMismatch": 1}
    lr = {"Pattern": ["AC+G[AT]"], "UseRegexp": True}
    want2 = (oracle.grep(data, False, json.dumps(gm)), oracle.grep(data, False, json.dumps(gd)),
             oracle.locate(data, False, json.dumps(lm)), oracle.locate(data, False, json.dumps(lr)))
    for mode in (None, "view"):
        if mode:
            monkeypatch.setenv("BSK_TEXT", mode)
        got = (bsk.Grep(frame(data, False), _Opts(g)), bsk.Locate(frame(data, False), _Opts(l)), bsk.RmDup(frame(data, False), _Opts(r)))
        assert got == want
        got2 = (bsk.Grep(frame(data, False), _Opts(gm)), bsk.Grep(frame(data, False), _Opts(gd)),
                bsk.Locate(frame(data, False), _Opts(lm)), bsk.Locate(frame(data, False), _Opts(lr)))
        assert got2 == want2


# ---------------------------------------------------------------- -d / -m: Shift-And (k_grep_shiftand) against the position-wise kernel
@pytest.mark.parametrize("i", range(len(GREP_GEN_OPTS)))
def test_grep_class_patterns_both_engines(i, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(1500 + i)
    data = planted_fastq(rng, 500)
    n1 = check_grep(data, True, GREP_GEN_OPTS[i])
    monkeypatch.setenv("BSK_GREP_SHIFTAND", "off")
    assert check_grep(data, True, GREP_GEN_OPTS[i]) == n1


def test_grep_class_patterns_engine_limits(monkeypatch):
    """64 positions is the last pattern length with one state word; 9 (strand, pattern) tables and 4 mismatches leave the fast kernel"""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(1600)
    recs, pats = [], []
    base = "".join(rng.choice("ACGT") for _ in range(70))
    for k in range(300):
        L = rng.randint(0, 200)
        s = [rng.choice("ACGT") for _ in range(L)]
        if k % 3 == 0 and L >= 70:
            p = rng.randrange(L - 69)
            s[p:p + 70] = base
            if k % 2:
                s[p + 10] = "A" if s[p + 10] != "A" else "C"
                s[p + 69] = "A" if s[p + 69] != "A" else "C"
        recs.append("@r%d\n%s\n+\n%s\n" % (k, "".join(s), "I" * L))
    data = "".join(recs).encode()
    for m in (63, 64, 65, 70):
        for mm in (0, 1, 2, 3, 4):
            o = {"Pattern": [base[:m]], "MaxMismatch": mm} if mm else {"Pattern": [base[:m - 1] + "N"], "Degenerate": True}
            assert check_grep(data, True, o) > 0
    five = [base[j:j + 12] for j in range(0, 50, 10)]
    assert check_grep(data, True, {"Pattern": five, "MaxMismatch": 1}) > 0              # 10 tables
    assert check_grep(data, True, {"Pattern": five[:4], "MaxMismatch": 1}) > 0          # 8 tables
    assert check_grep(data, True, {"Pattern": five, "MaxMismatch": 1, "OnlyPositiveStrand": True}) > 0
    # windows shorter than the pattern, empty sequences, a region of one base
    assert check_grep(data, True, {"Pattern": [base[:20]], "MaxMismatch": 2, "Region": "1:19"}) == 0
    check_grep(data, True, {"Pattern": [base[:20]], "MaxMismatch": 2, "Region": "-20:-1"})
    check_grep(b"@e\n\n+\n\n@f\nA\n+\nI\n", True, {"Pattern": ["A"], "MaxMismatch": 1})
