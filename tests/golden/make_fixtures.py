#!/usr/bin/env python3
"""Generates tests/golden/fixtures.json: small input / expected-output pairs for every operator, produced by the CPU
oracle (oracle/, the restatement of the reference algorithm) in the build container.  The reference ships no tests
and cannot be built here, so these are REGRESSION pins of the restatement -- the cases follow
SURVEY.md 8(c)'s list (region table, '@'-leading quality lines, wrapped FASTA at 60 / 70 / 0, translate of a 9-mer in
all frames and the ambiguous-codon examples, overlapping motifs for locate) plus one case per operator of 8(f).
Several expected values are also derived by hand in tests/test_oracle_kat.py and tests/test_records_cpu.py.
Data, not code: the file travels to the GPU box, where tests/test_golden_gpu.py compares the HIP path with it."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402

FQ10 = (b"@r1 first\nACGTACGTAC\n+\nIIIIIIIIII\n@r2\nGGGGCCCC\n+r2\n@@@@++++\n@r3 q starts with at\nTTTT\n+\n@III\n"
        b"@r4\nACGTNNNN\n+\n+III####\n@r5\nA\n+\n#\n@r6 empty\n\n+\n\n@r7\nacgtACGT\n+\n!!!!IIII\n@r8\nCCCCCCCCCCGGGGGGGGGG\n+\n"
        b"IIIIIIIIII##########\n@r9 x y z\nACGU\n+\nIIII\n@r10\nTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT\n+\n5555555555?????????????????????\n")
FQ10 = FQ10.replace(b"5555555555?????????????????????", b"5555555555????????????????????")
SEQ = "ACGTTGCAAGCTGATCGATCGTAGCTAGCTAGGATCCGATATCGCGCGATATAAGCTTGGCCAATTGGCCATGAAATAG" * 3


def fasta(width):
    body = "".join(SEQ[j:j + width] + "\n" for j in range(0, len(SEQ), width)) if width else SEQ + "\n"
    return (">chr1 first record\n" + body + ">chr2\nACGTNRYKM\n>chr3 empty\n>chr1 again\nAAAATTTTGGAAAA\n").encode()


CASES = []


def add(name, op, fn, data, fastq, opts, **kw):
    try:
        exp = fn(data, fastq, json.dumps(opts), **kw)
        CASES.append({"name": name, "op": op, "fastq": fastq, "opts": opts, "input": data.decode("latin1"),
                      "expected": exp.decode("latin1")})
    except oracle.OracleError as e:
        CASES.append({"name": name, "op": op, "fastq": fastq, "opts": opts, "input": data.decode("latin1"), "error": str(e)})


for w in (60, 70, 0):
    add(f"seq fasta w{w}", "seq", oracle.seq, fasta(w), False, {})
    add(f"seq -r -p fasta w{w}", "seq", oracle.seq, fasta(w), False, {"Reverse": True, "Complement": True, "Config": {"LineWidth": 50}})
    add(f"stats fasta w{w}", "stats", lambda d, f, o: oracle.stats_string(d, f, o, name="input0").encode(), fasta(w), False, {"All": True, "Tabular": True})
add("seq fastq", "seq", oracle.seq, FQ10, True, {})
add("seq -n -i fastq", "seq", oracle.seq, FQ10, True, {"Name": True, "OnlyId": True})
add("seq -m 5 -M 10 -u", "seq", oracle.seq, FQ10, True, {"MinLen": 5, "MaxLen": 10, "UpperCase": True})
add("stats fastq -a", "stats", lambda d, f, o: oracle.stats_string(d, f, o, name="input0").encode(), FQ10, True, {"All": True, "Tabular": True})
for region in ("1:12", "13:24", "-12:-1", "-24:-13", "13:1000", "-5:-1", "1:-1", "2:-2"):   # bigseqkit-cli/helper.go:348-361
    add(f"subseq -r {region} fasta", "subseq", oracle.subseq, fasta(60), False, {"Region": region})
    add(f"subseq -r {region} fastq", "subseq", oracle.subseq, FQ10, True, {"Region": region})
for frame in (["1"], ["2"], ["3"], ["-1"], ["-2"], ["-3"], ["6"]):
    add(f"translate 9-mer frame {','.join(frame)}", "translate", oracle.translate, b">s\nATGGCCTAA\n", False, {"Frame": frame})
for codon in ("TAN", "NNN", "ACN", "AAR", "AAY", "MGR", "YTA", "CTN", "RAY", "SAR", "ATH"):   # cli/translate.go:42-52
    add(f"translate ambiguous {codon}", "translate", oracle.translate, f">s\nATG{codon}TAA\n".encode(), False, {"AllowUnknownCodon": True})
add("translate -f 6 -F --trim table 11", "translate", oracle.translate, fasta(60), False,
    {"Frame": ["6"], "AppendFrame": True, "Trim": True, "TranslTable": 11, "AllowUnknownCodon": True})
LOC = b">s1 d\nAAAATTTTGGAAAA\n>s2\nACGTTGCAAGCT\n"
for o in ({"Pattern": ["AA"]}, {"Pattern": ["AA"], "NonGreedy": True}, {"Pattern": ["AAAA", "TGCA"], "Bed": True},
          {"Pattern": ["AAAA", "TGCA"], "Gtf": True}, {"Pattern": ["aaaa"], "IgnoreCase": True, "OnlyPositiveStrand": True},
          {"Pattern": ["GCTAC"], "Circular": True}, {"Pattern": ["ANNT"], "Degenerate": True}, {"Pattern": ["ACGTAG"], "MaxMismatch": 1},
          {"Pattern": ["A[AT]T"], "UseRegexp": True}):
    add("locate " + json.dumps(o), "locate", oracle.locate, LOC, False, o)
for o in ({"Pattern": ["r3", "r9"]}, {"Pattern": ["CCCC"], "BySeq": True}, {"Pattern": ["cccc"], "BySeq": True, "IgnoreCase": True, "InvertMatch": True},
          {"Pattern": ["^r1"], "UseRegexp": True}, {"Pattern": ["ACGN"], "Degenerate": True}, {"Pattern": ["r1", "r2"], "DeleteMatched": True}):
    add("grep " + json.dumps(o), "grep", oracle.grep, FQ10, True, o)
DUP = FQ10 + b"@r2 dup\nGGGGCCCC\n+\nIIIIIIII\n@R7\nACGTACGT\n+\nIIIIIIII\n"
for o in ({}, {"BySeq": True}, {"BySeq": True, "IgnoreCase": True}, {"ByName": True}):
    add("rmdup " + json.dumps(o), "rmdup", oracle.rmdup, DUP, True, o)
add("fq2fa", "fq2fa", oracle.fq2fa, FQ10, True, {})
for r in ("1:3", "-3:-1", "2", "-1:4"):
    add(f"range {r}", "range", oracle.range_, FQ10, True, {"Range": r})
add("range 9:3 (error)", "range", oracle.range_, FQ10, True, {"Range": "9:3"})
add("head -n 4", "head", oracle.head, fasta(60), False, {"N": 4})
add("duplicate -n 2", "duplicate", oracle.duplicate, fasta(70), False, {"Times": 2})
add("rename", "rename", oracle.rename, fasta(60), False, {})
add("sort -l -r", "sort", oracle.sort, FQ10, True, {"ByLength": True, "Reverse": True})
add("sort -s -i", "sort", oracle.sort, FQ10, True, {"BySeq": True, "IgnoreCase": True})
add("sort by id", "sort", oracle.sort, fasta(60), False, {})
NAT = "".join(f">{i} d\nACGT\n" for i in ["chr10", "chr2", "chr1", "chrX", "chr1_random", "chr01", "Chr3", "chr2a", "chr", "10", "9"]).encode()
add("sort -N", "sort", oracle.sort, NAT, False, {"InNaturalOrder": True})
add("sort -N -i -r", "sort", oracle.sort, NAT, False, {"InNaturalOrder": True, "IgnoreCase": True, "Reverse": True})
add("faidx rows fasta", "faidx", oracle.faidx, fasta(60), False, {})
add("faidx rows fastq", "faidx", oracle.faidx, FQ10, True, {"FullHead": True})
add("faidx queries", "faidx_query", oracle.faidx_query, fasta(60), False, {"Regions": ["chr1:5-20", "chr2:-3", "chr1:30-21", "chr9"]})

json.dump({"generator": "tests/golden/make_fixtures.py (oracle/ restatement; regression pins, see the docstring)", "cases": CASES},
          open(os.path.join(HERE, "fixtures.json"), "w"), indent=0)
print(len(CASES), "cases,", sum("error" in c for c in CASES), "of them errors")
