"""The fused pattern filter (stream_filter.hip): `grep -s -p` / `locate -p` with exact patterns of 11..64 bytes on FASTQ
select their records inside the streaming pass.  Parity against the oracle for every pattern alignment, both strands,
-i, -v, -P, several patterns, periodic and palindromic patterns (table entries shared by several alignments), patterns
that also occur in headers and quality lines, lines that span many tiles, range boundaries (tiny ranges), the overflow
fallback -- and a check that the filter kernel is the one that ran."""
import ctypes as C
import json
import random

import pytest

import oracle
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

pytestmark = pytest.mark.gpu

COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def rc(s):
    return s.translate(COMP)[::-1]


def dev(data):
    import torch
    return torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda() if len(data) else torch.empty(0, dtype=torch.uint8).cuda()


class _Opts:
    def __init__(self, d):
        self.d = dict(d)
        self._v = self.d

    def to_json(self):
        return json.dumps(self.d)


def fq(records):
    return "".join("@%s\n%s\n+\n%s\n" % (n, s, q) for n, s, q in records).encode()


def run_grep_profiled(data, opts):
    """bsk_grep_run with the launch profile on: (output bytes, k_filter launches, k_index launches)"""
    t = dev(data)
    out = _lib.Out()
    with bsk.Operator("Grep", json.dumps(opts), 0) as op:
        lib.bsk_profile_enable(op.ctx, 1)
        check(lib.bsk_grep_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out)), op.ctx)
        buf = C.create_string_buffer(max(1, out.len))
        check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
        ms, nf, ni = C.c_double(), C.c_uint64(), C.c_uint64()
        lib.bsk_profile_read(op.ctx, b"k_filter", C.byref(ms), C.byref(nf))
        lib.bsk_profile_read(op.ctx, b"k_index", C.byref(ms), C.byref(ni))
        return buf.raw[:out.len], nf.value, ni.value


def both(data, opts, expect_filter=True):
    want = oracle.grep(data, True, json.dumps(dict(opts, Count=False)))
    got, nf, ni = run_grep_profiled(data, dict(opts, Count=False))
    assert got == want, (opts, len(got), len(want))
    if expect_filter is not None:
        assert (nf > 0) == expect_filter, (nf, ni)
    lwant = oracle.locate(data, True, json.dumps({k: v for k, v in opts.items() if k in ("Pattern", "IgnoreCase", "OnlyPositiveStrand")}))
    lopts = {k: v for k, v in opts.items() if k in ("Pattern", "IgnoreCase", "OnlyPositiveStrand")}
    lgot = bsk.Locate(bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(data)]), _Opts(lopts))
    assert lgot == lwant, lopts
    return want


def random_reads(rng, n, L=150, alphabet="ACGT"):
    return [("r%d d%d" % (i, rng.randint(0, 9)), "".join(rng.choice(alphabet) for _ in range(L)),
             "".join(chr(rng.randint(35, 73)) for _ in range(L))) for i in range(n)]


@pytest.mark.parametrize("m", [11, 12, 13, 15, 16, 17, 18, 31, 32, 33, 64])
def test_every_alignment_and_both_strands(m, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(100 + m)
    pat = "".join(rng.choice("ACGT") for _ in range(m))
    recs = random_reads(rng, 400, 150 if m <= 33 else 200)
    planted = 0
    for i, (n, s, q) in enumerate(recs):
        if i % 3 == 0:
            pos = (i // 3) % (len(s) - m + 1)          # walks through every offset, hence every alignment mod 16
            use = pat if (i // 3) % 2 == 0 else rc(pat)
            recs[i] = (n, s[:pos] + use + s[pos + m:], q)
            planted += 1
    data = fq(recs)
    want = both(data, {"BySeq": True, "Pattern": [pat]})
    assert want.count(b"\n") // 4 >= planted
    both(data, {"BySeq": True, "Pattern": [pat], "OnlyPositiveStrand": True})
    both(data, {"BySeq": True, "Pattern": [pat], "InvertMatch": True})


def test_ignore_case_and_mixed_case_text(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(7)
    pat = "ACGTTGCAAGCTA"
    recs = random_reads(rng, 300, 120, "ACGTacgt")
    for i in range(0, 300, 4):
        pos = i % 100
        use = [pat, pat.lower(), rc(pat), "AcGtTgCaAgCtA"][(i // 4) % 4]
        n, s, q = recs[i]
        recs[i] = (n, s[:pos] + use + s[pos + len(use):], q)
    data = fq(recs)
    both(data, {"BySeq": True, "Pattern": [pat], "IgnoreCase": True})
    both(data, {"BySeq": True, "Pattern": [pat.lower()], "IgnoreCase": True})
    both(data, {"BySeq": True, "Pattern": [pat]})                      # case-sensitive: only the exact spellings


def test_several_patterns_share_one_pass(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "8192")
    rng = random.Random(11)
    pats = ["".join(rng.choice("ACGT") for _ in range(rng.randint(11, 40))) for _ in range(8)]
    recs = random_reads(rng, 600, 180)
    for i in range(0, 600, 5):
        p = pats[(i // 5) % 8]
        use = p if i % 2 == 0 else rc(p)
        pos = rng.randrange(180 - len(p) + 1)
        n, s, q = recs[i]
        recs[i] = (n, s[:pos] + use + s[pos + len(p):], q)
    data = fq(recs)
    both(data, {"BySeq": True, "Pattern": pats[:4]})                                # 4 patterns x 2 strands x 4 alignments = 32 entries
    both(data, {"BySeq": True, "Pattern": pats, "OnlyPositiveStrand": True})       # 8 x 1 x 4
    both(data, {"BySeq": True, "Pattern": pats[:3], "InvertMatch": True})
    # more (pattern, strand) pairs than the table has entry bits: the record-table path answers
    both(data, {"BySeq": True, "Pattern": pats[:5]}, expect_filter=False)
    both(data, {"BySeq": True, "Pattern": pats}, expect_filter=False)


@pytest.mark.parametrize("pat", ["AAAAAAAAAAAA", "ACGTACGTACGT", "ACACACACACACAC", "GAATTCGAATTC", "ACGTTGCAACGT"])
def test_periodic_and_palindromic_patterns(pat, monkeypatch):
    """the same 8 bytes at several (pattern, strand, alignment)s: one table entry, every alignment verified"""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(len(pat) + ord(pat[1]))
    recs = random_reads(rng, 240, 100)
    for i in range(0, 240, 3):
        pos = (i // 3) % (100 - len(pat) + 1)
        n, s, q = recs[i]
        recs[i] = (n, s[:pos] + pat + s[pos + len(pat):], q)
    data = fq(recs)
    both(data, {"BySeq": True, "Pattern": [pat]}, expect_filter=None)


def test_patterns_in_headers_and_quality_lines_do_not_count(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    pat = "ACGTTGCAAGCT"
    rng = random.Random(5)
    recs = []
    for i in range(300):
        s = "".join(rng.choice("ACG") for _ in range(90))                 # no T: the pattern cannot occur by chance
        q = "".join(chr(rng.randint(35, 73)) for _ in range(90))
        name = "r%d" % i
        kind = i % 6
        if kind == 0: name += " " + pat                                   # in the header
        if kind == 1: q = q[:20] + pat + q[32:]                           # in the quality line (all letters are legal there)
        if kind == 2: s = s[:30] + pat + s[42:]                           # the real thing
        if kind == 3: s = s[:84] + pat[:6]; q = pat[6:] + q[6:]           # across the "+" line: not an occurrence
        if kind == 4: name += " " + pat[:6]; s = pat[6:] + s[6:]          # header end + sequence start: not an occurrence
        recs.append((name, s, q))
    data = fq(recs)
    want = both(data, {"BySeq": True, "Pattern": [pat]})
    assert want.count(b"@r") == 50
    both(data, {"BySeq": True, "Pattern": [pat], "InvertMatch": True})


def test_long_reads_lines_span_many_tiles(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "65536")
    rng = random.Random(9)
    pat = "GATTACAGATTACAGATTACA"
    recs = []
    for i in range(60):
        L = rng.choice([5000, 12000, 300, 20000])
        s = "".join(rng.choice("ACGT") for _ in range(L))
        if i % 2 == 0:
            pos = rng.randrange(L - len(pat))
            s = s[:pos] + (pat if i % 4 == 0 else rc(pat)) + s[pos + len(pat):]
        recs.append(("long%d" % i, s, "I" * L))
    both(fq(recs), {"BySeq": True, "Pattern": [pat]})


def test_overflow_of_the_pending_hit_list_falls_back_to_the_table_path(monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    recs = [("a%d" % i, "A" * 150 if i % 2 else "ACGT" * 37 + "AC", "I" * 150) for i in range(200)]
    data = fq(recs)
    opts = {"BySeq": True, "Pattern": ["AAAAAAAAAAAA"]}
    want = oracle.grep(data, True, json.dumps(opts))
    got, nf, ni = run_grep_profiled(data, opts)
    assert got == want and want.count(b"\n") // 4 == 100
    assert nf > 0 and ni > 0        # the filter ran, gave up, and the index pass took over


def test_c3_motif_workload_counts(monkeypatch):
    """BASELINE C3 at 1/100 scale: the synthetic motif file, grep -s -p ACGTTGCAAGCT, count and records == oracle"""
    import torch
    nrec = 400_000
    t = torch.empty(317 * nrec, dtype=torch.uint8, device="cuda")
    check(lib.bsk_synth_device(_lib.SYNTH_FASTQ150, 42, _lib.SYNTH_FLAG_MOTIF, 0, C.c_void_p(t.data_ptr()), t.numel(), 0, None))
    torch.cuda.synchronize()
    host = bytes(t.cpu().numpy().tobytes())
    opts = {"BySeq": True, "Pattern": ["ACGTTGCAAGCT"]}
    want = oracle.grep(host, True, json.dumps(opts))
    got = bsk.Grep(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), _Opts(opts))
    assert got == want
    assert want.count(b"\n") // 4 >= nrec // 50        # every 100th record on each strand, plus chance hits
    assert bsk.GrepCount(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), _Opts(opts)) == want.count(b"\n") // 4
    lgot = bsk.Locate(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), _Opts({"Pattern": ["ACGTTGCAAGCT"]}))
    assert lgot == oracle.locate(host, True, json.dumps({"Pattern": ["ACGTTGCAAGCT"]}))
