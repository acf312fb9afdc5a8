"""Parity of `locate` (exact patterns) against the CPU oracle, through the C ABI."""
import json
import random

import pytest

import oracle
import seqgen
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


def check(data, fastq, opts):
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    want = oracle.locate(data, fastq, json.dumps(opts))
    got = bsk.Locate(bsk.SeqFrame(fmt, [dev(data)]), _Opts(opts))
    assert got == want, (opts, got[:400], want[:400])
    return got


def test_locate_hand_cases_overlap_and_strands():
    fa = b">s1 d\nAAAATTTTGGAAAA\n>s2\nACGTTGCAAGCT\n"
    got = check(fa, False, {"Pattern": ["AA"]})
    assert got.startswith(b"seqID\tpatternName\tpattern\tstrand\tstart\tend\tmatched\ns1\tAA\tAA\t+\t1\t2\tAA\n")
    assert b"s1\tAA\tAA\t-\t7\t8\tAA\n" in got
    check(fa, False, {"Pattern": ["AA"], "NonGreedy": True})
    check(fa, False, {"Pattern": ["AA"], "NonGreedy": True, "OnlyPositiveStrand": True, "HideMatched": True})
    check(fa, False, {"Pattern": ["AAAA", "TGCA"], "Bed": True})
    check(fa, False, {"Pattern": ["AAAA", "TGCA", "AAAA"], "Gtf": True})
    check(fa, False, {"Pattern": ["aaaa"], "IgnoreCase": True})
    check(fa, False, {"Pattern": ["AAAAAAAA"], "Circular": True})
    check(fa, False, {"Pattern": ["GCTAC"], "Circular": True})   # wraps around the end of s2
    check(b"", False, {"Pattern": ["AA"]})


LOC_OPTS = [
    {"Pattern": ["ACG"]},
    {"Pattern": ["ACG", "TTT", "GGCC"], "HideMatched": True},
    {"Pattern": ["AAA"], "NonGreedy": True},
    {"Pattern": ["ACGT"], "Circular": True},
    {"Pattern": ["ACGT"], "Circular": True, "NonGreedy": True, "Bed": True},
    {"Pattern": ["acg", "TTt"], "IgnoreCase": True, "Gtf": True},
    {"Pattern": ["TGCATG"], "OnlyPositiveStrand": True},
    {"Pattern": ["A" * 40]},
    {"Pattern": ["NNN", "RY"], "Config": {"SeqType": "dna"}},
]


@pytest.mark.parametrize("i", range(len(LOC_OPTS)))
def test_locate_fastq(i, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(900 + i)
    data = seqgen.random_fastq(rng, 400, 0, 120, alphabet="ACGTNacgt")
    check(data, True, LOC_OPTS[i])


@pytest.mark.parametrize("width", [60, 0, 7])
@pytest.mark.parametrize("i", range(len(LOC_OPTS)))
def test_locate_fasta(i, width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(950 + i)
    data = seqgen.random_fasta(rng, 120, 0, 500, width=width, alphabet="ACGTacgtN", final_newline=i % 2 == 0)
    check(data, False, LOC_OPTS[i])


def test_locate_protein_minus_strand_is_searched_as_written():
    # locate.go:669 consults the option, not the guessed alphabet: the reverse of a protein is searched too
    prot = b">p\nMKVLAAGIVKM\n"
    got = check(prot, False, {"Pattern": ["MK"]})
    assert b"\t-\t" in got


def test_locate_option_errors():
    for opts, msg in [({}, "one of flags -p (--pattern) and -f (--pattern-file) needed"),
                      ({"Pattern": ["AC.T"]}, "illegal DNA/RNA/Protein sequence: AC.T, you may switch on"),
                      ({"Pattern": ["AC!"]}, "illegal DNA/RNA/Protein sequence: AC!")]:
        with pytest.raises(bsk.BskError) as e:
            bsk.Operator("Locate", json.dumps(opts), -1)
        assert msg in str(e.value)
        with pytest.raises(oracle.OracleError) as oe:
            oracle.locate(b">a\nA\n", False, json.dumps(opts))
        assert msg in str(oe.value)


def test_locate_header_only_on_partition_zero():
    rng = random.Random(3)
    data = seqgen.random_fastq(rng, 300, 10, 80, alphabet="ACGT")
    fr = bsk.ReadFASTQN(data, 3)
    assert len(fr.shards) == 3
    fr = bsk.SeqFrame(fr.format, [dev(s) for s in fr.shards])
    got = bsk.Locate(fr, _Opts({"Pattern": ["ACG"]}))
    assert got == oracle.locate(data, True, '{"Pattern": ["ACG"]}', nparts=1)
    assert got.count(b"seqID\t") == 1


# ---------------------------------------------------------------- -d, -m, -F, -f
LOC_GEN_OPTS = [
    {"Pattern": ["ACGT"], "MaxMismatch": 1},
    {"Pattern": ["ACGTAC", "TTTTT"], "MaxMismatch": 2, "HideMatched": True},
    {"Pattern": ["acgtac"], "MaxMismatch": 1, "IgnoreCase": True},
    {"Pattern": ["ACGTAC"], "MaxMismatch": 1, "Circular": True},
    {"Pattern": ["ACGTAC"], "MaxMismatch": 1, "Circular": True, "Bed": True},
    {"Pattern": ["ACGTAC"], "MaxMismatch": 2, "Gtf": True, "OnlyPositiveStrand": True},
    {"Pattern": ["ACGTAC"], "MaxMismatch": 1, "NonGreedy": True},
    {"Pattern": ["ACG", "GGCC"], "UseFmi": True},
    {"Pattern": ["acg"], "UseFmi": True, "IgnoreCase": True, "Circular": True},
    {"Pattern": ["ANNT"], "Degenerate": True},
    {"Pattern": ["RYRY", "ACGN"], "Degenerate": True, "HideMatched": True},
    {"Pattern": ["acgn"], "Degenerate": True, "IgnoreCase": True},
    {"Pattern": ["ACNNGT"], "Degenerate": True, "NonGreedy": True},
    {"Pattern": ["ACNNGT"], "Degenerate": True, "Circular": True, "Bed": True},
    {"Pattern": ["WSWS"], "Degenerate": True, "Circular": True, "NonGreedy": True, "Gtf": True},
    {"Pattern": ["NNNNNNNN"], "Degenerate": True, "OnlyPositiveStrand": True},
]


@pytest.mark.parametrize("i", range(len(LOC_GEN_OPTS)))
def test_locate_general_fastq(i, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(1900 + i)
    data = seqgen.random_fastq(rng, 300, 0, 100, alphabet="ACGT" * 5 + "acgtN")
    got = check(data, True, LOC_GEN_OPTS[i])
    assert got.count(b"\n") > 10


@pytest.mark.parametrize("width", [60, 0, 7])
@pytest.mark.parametrize("i", range(len(LOC_GEN_OPTS)))
def test_locate_general_fasta(i, width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(1950 + i)
    data = seqgen.random_fasta(rng, 80, 0, 300, width=width, alphabet="ACGT" * 5 + "acgtN", final_newline=i % 2 == 0)
    check(data, False, LOC_GEN_OPTS[i])


def test_locate_fm_index_branch_consults_the_alphabet_and_keeps_raw_minus_coordinates():
    prot = b">p\nMKVLAAGIVKM\n"
    assert b"\t-\t" not in check(prot, False, {"Pattern": ["MK"], "UseFmi": True})
    # circular '-' hit that wraps: begin = l - i - len + 1 without the +l shift of the exact branch (locate.go:330)
    got = check(b">c\nGTTTTGAC\n", False, {"Pattern": ["ACGTCA"], "MaxMismatch": 1, "Circular": True})
    assert b"c\tACGTCA\tACGTCA\t-\t-3\t2\tACGTCA\n" in got
    check(b">c\nAAAAAAAC\n", False, {"Pattern": ["TTTT"], "UseFmi": True, "Circular": True, "Bed": True})


def test_locate_pattern_file(tmp_path):
    rng = random.Random(8)
    data = seqgen.random_fasta(rng, 100, 0, 400, alphabet="ACGT")
    f = tmp_path / "motifs.fa"
    f.write_text(">m1 first motif\nACGT\n>m2\nTTG\nCA\n>m1 first motif\nGGCC\n>deg\nANNT\n")
    got = check(data, False, {"PatternFile": str(f)})
    assert b"\tm1 first motif\tGGCC\t" in got and b"\tm2\tTTGCA\t" in got
    check(data, False, {"PatternFile": str(f), "Degenerate": True, "HideMatched": True})
    check(data, False, {"PatternFile": str(f), "MaxMismatch": 1})
    empty = tmp_path / "empty.fa"
    empty.write_text("\n")
    with pytest.raises(bsk.BskError) as e:
        bsk.Operator("Locate", json.dumps({"PatternFile": str(empty)}), -1)
    assert "no FASTA sequences found in pattern file" in str(e.value)


# ---------------------------------------------------------------- chromosome-sized records (cells of 64 Ki start positions)
def _long_fasta(seed):
    rng = random.Random(seed)
    recs = []
    for k, (L, w) in enumerate([(200_000, 60), (50, 60), (140_000, 0), (70_001, 11), (0, 60), (65_536, 80), (131_073, 60)]):
        s = "".join(rng.choice("ACGT") for _ in range(L))
        if k == 0:   # matches that straddle the chunk boundary at 65536 and the end of the record
            s = s[:65530] + "ACGTACGTACGT" + s[65542:-6] + "ACGTAC"
        if w:
            body = "".join(s[j:j + w] + "\n" for j in range(0, L, w))
        else:
            body = s + "\n"
        recs.append(f">chr{k} long\n{body}")
    return "".join(recs).encode()


@pytest.mark.parametrize("i", range(len(LOC_OPTS) + len(LOC_GEN_OPTS)))
def test_locate_long_records(i, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    monkeypatch.setenv("BSK_LONG_BYTES", "60000")
    opts = (LOC_OPTS + LOC_GEN_OPTS)[i]
    if len(opts["Pattern"][0]) < 4 and not opts.get("NonGreedy"):
        opts = dict(opts, Pattern=[p * 2 for p in opts["Pattern"]])  # keep the row count of the oracle run reasonable
    check(_long_fasta(11 + i), False, opts)


def test_locate_every_record_is_long(monkeypatch):
    # threshold below every sequence: all rows come from cell launches, the per-record kernel has nothing to do
    monkeypatch.setenv("BSK_LONG_BYTES", "1")
    fa = seqgen.random_fasta(random.Random(5), 40, 0, 400, width=60, alphabet="ACGT")
    for opts in ({"Pattern": ["ACG", "TT"]}, {"Pattern": ["ACGT"], "Circular": True, "Bed": True},
                 {"Pattern": ["ACNT"], "Degenerate": True}, {"Pattern": ["ACG", "GGCC"], "UseFmi": True}):
        check(fa, False, opts)


# ---------------------------------------------------------------- -r for fixed-length expressions (PARITY.md LOCRE)
LOC_RE_OPTS = [
    {"Pattern": ["A[AT]T"], "UseRegexp": True},
    {"Pattern": ["g.a", "AC[^A]T"], "UseRegexp": True, "IgnoreCase": True},
    {"Pattern": ["AC\\wT", "[ACG][ACG][ACG]TT"], "UseRegexp": True, "HideMatched": True},
    {"Pattern": ["A.G.T"], "UseRegexp": True, "Circular": True, "Bed": True},
    {"Pattern": ["[AG]CG[CT]"], "UseRegexp": True, "NonGreedy": True, "Gtf": True},
    {"Pattern": ["TT[ACGT]AA"], "UseRegexp": True, "OnlyPositiveStrand": True},
]


@pytest.mark.parametrize("width", [60, 0, 7])
@pytest.mark.parametrize("i", range(len(LOC_RE_OPTS)))
def test_locate_regexp_of_fixed_length(i, width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(900 + i)
    data = seqgen.random_fasta(rng, 80, 0, 300, width=width, alphabet="ACGT" * 5 + "acgtN", final_newline=i % 2 == 0)
    check(data, False, LOC_RE_OPTS[i])
    fq = seqgen.random_fastq(rng, 200, 0, 100, alphabet="ACGT" * 5 + "acgtN")
    check(fq, True, LOC_RE_OPTS[i])


def test_locate_regexp_hand_case_pattern_file_and_what_is_refused(tmp_path):
    fa = b">s1 d\nAAAATTTTGGAAAA\n>s2\nACGTTGCAAGCT\n"
    got = check(fa, False, {"Pattern": ["A[AT]T"], "UseRegexp": True})
    assert got == (b"seqID\tpatternName\tpattern\tstrand\tstart\tend\tmatched\ns1\tA[AT]T\tA[AT]T\t+\t3\t5\tAAT\n"
                   b"s1\tA[AT]T\tA[AT]T\t+\t4\t6\tATT\ns1\tA[AT]T\tA[AT]T\t-\t4\t6\tAAT\ns1\tA[AT]T\tA[AT]T\t-\t3\t5\tATT\n")
    pf = tmp_path / "pats.fa"
    pf.write_text(">motif one\nG[GC]A\n>second\nTT.T\n")
    check(fa, False, {"PatternFile": str(pf), "UseRegexp": True})
    check(fa, False, {"Pattern": ["A{4}", "(TT)G"], "UseRegexp": True})   # a fixed count and a plain group are still a chain
    # expressions that need match priorities run on the position-reporting matcher (round 1 refused them)
    for expr in ("AC+G", "A|C", "(AC)?G", "^ACG", "AC{1,2}"):
        check(fa, False, {"Pattern": [expr], "UseRegexp": True})
    # hand case (locate.go:583-667): A+ on AAAATTTTGGAAAA, greedy stepping: 1-4; 2-4, 3-4, 4-4 lie inside it and are dropped
    # (:604-614); then 11-14 (12-14 .. dropped).  '-' strand: RevCom = TTTTCCAAAATTTT, A+ at 7-10 -> begin 14-10+1 = 5, end 8
    got = check(fa[:21], False, {"Pattern": ["A+"], "UseRegexp": True})
    assert got == (b"seqID\tpatternName\tpattern\tstrand\tstart\tend\tmatched\n"
                   b"s1\tA+\tA+\t+\t1\t4\tAAAA\ns1\tA+\tA+\t+\t11\t14\tAAAA\ns1\tA+\tA+\t-\t5\t8\tAAAA\n")


# ---------------------------------------------------------------- locate -r, matches of variable length (PARITY.md LOCRE)
VM_EXPRS = ["AC+G", "A[CG]*T", "(AC|GT)+", "^A.*T$", "T{2,4}", "G.*?C", "(?:CG|C)(A|AT)T?", "A+", "^[ACGT]{3}", "N+|ACGT", "A(C|G){0,2}T$", "T*",
            r"\bAC+G", r"T+\b", r"A\B[CG]+"]  # ASCII word boundaries (round 4): inside a sequence only the ends of the searched text are boundaries


@pytest.mark.parametrize("expr", VM_EXPRS)
def test_locate_regexp_variable_length(expr, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(len(expr) * 7 + ord(expr[0]))
    fa = seqgen.random_fasta(rng, 150, 0, 300, alphabet="ACGTN")
    fq = seqgen.random_fastq(rng, 200, 0, 80, alphabet="ACGT")
    for o in ({}, {"OnlyPositiveStrand": True}, {"NonGreedy": True}, {"IgnoreCase": True, "HideMatched": True}, {"Bed": True}):
        check(fa, False, dict({"Pattern": [expr], "UseRegexp": True}, **o))
    check(fq, True, {"Pattern": [expr, "ACG", "C[AT]"], "UseRegexp": True})
    # --circular with the position-reporting matcher (round 2 refused it): the text is the sequence twice, a match begins in
    # the first copy, '-' matches across the origin are reported l further on (locate.go:231-234, 595-597, 694-703)
    small = seqgen.random_fasta(rng, 120, 1, 40, alphabet="ACGT")
    for o in ({}, {"OnlyPositiveStrand": True}, {"NonGreedy": True}, {"Bed": True}):
        check(small, False, dict({"Pattern": [expr], "UseRegexp": True, "Circular": True}, **o))
    check(fq, True, {"Pattern": [expr, "C[AT]+G"], "UseRegexp": True, "Circular": True})


def test_locate_regexp_variable_length_circular_hand_case():
    # GTAAAAAC, C+G+T?: '+' strand: the doubled text GTAAAAACGTAAAAAC holds "CGT" at 8..10 (begins in the first copy, ends
    # beyond l = 8); '-' strand: RevCom doubled GTTTTTACGTTTTTAC holds "CGT" at 0-based 7: begin = 8-0-10+1 = -1, end = 1,
    # crossing -> 7, 9 (locate.go:697-702)
    got = check(b">c\nGTAAAAAC\n", False, {"Pattern": ["C+G+T?"], "UseRegexp": True, "Circular": True})
    assert got == (b"seqID\tpatternName\tpattern\tstrand\tstart\tend\tmatched\n"
                   b"c\tC+G+T?\tC+G+T?\t+\t8\t10\tCGT\nc\tC+G+T?\tC+G+T?\t-\t7\t9\tCGT\n")


@pytest.mark.parametrize("i", range(len(LOC_GEN_OPTS)))
def test_locate_general_with_and_without_the_prefilter(i, monkeypatch):
    """-d / -m: grep's Shift-And marks the records that hold an occurrence before the position-wise search runs (ops_host_search.cpp)"""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(2100 + i)
    data = seqgen.random_fastq(rng, 400, 0, 100, alphabet="ACGT" * 5 + "acgtN")
    got = check(data, True, LOC_GEN_OPTS[i])
    monkeypatch.setenv("BSK_LOCATE_NOPRE", "1")
    assert check(data, True, LOC_GEN_OPTS[i]) == got
    monkeypatch.delenv("BSK_LOCATE_NOPRE")
    # a rare pattern: most records are skipped by the prefilter
    o = dict(LOC_GEN_OPTS[i], Pattern=["ACGTTGCAAGCTAA"[:14]])
    if "MaxMismatch" in o or "Degenerate" in o:
        check(data, True, o)
