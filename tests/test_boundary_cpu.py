"""C-ABI boundary without a GPU: the library loads, exports every symbol bsk.h declares,
decodes the reference's option JSON, reproduces its Before() error texts, and its
host-side halves (record-boundary repair, StatsReduce, Stats(), StatsString()) agree
with the oracle.  No compute entry point is called."""
import ctypes as C
import json
import os
import random
import re

import pytest

import oracle
import seqgen
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "bsk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(bsk_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/bsk.h but not exported by libbsk.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"


def test_no_device_is_a_loud_error():
    if lib.bsk_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(bsk.BskError) as e:
        bsk.Operator("Stats", "{}", 0)
    assert e.value.code == _lib.BSK_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_options_json_defaults_match_reference_schema():
    # SURVEY.md section 11 example for `stats -a`
    want = ('{"Config":{"SeqType":"auto","ChunkSize":null,"BufferSize":null,"LineWidth":60,'
            '"IDRegexp":"^(\\\\S+)\\\\s?","IDNCBI":false,"Quiet":false,"AlphabetGuessSeqLength":10000,'
            '"ValidateSeqLength":10000},"Tabular":false,"GapLetters":"- .","All":true,"SkipErr":false,'
            '"FqEncoding":"sanger","Basename":false}\n')
    op = bsk.Operator("Stats", bsk.SeqKitStatsOptions().All(True).to_json(), -1)
    assert op.opts_json() == want
    json.loads(op.opts_json())


@pytest.mark.parametrize("name,cls,probe", [
    ("SeqTransform", bsk.SeqKitSeqOptions, {"GapLetters": "- \t.", "MinLen": -1, "MaxQual": -1, "QualAsciiBase": 33}),
    ("Grep", lambda: bsk.SeqKitGrepOptions().Pattern(["id1"]), {"Pattern": ["id1"], "MaxMismatch": 0, "Region": ""}),
    ("Locate", lambda: bsk.SeqKitLocateOptions().Pattern(["ACGT"]),
     {"Pattern": ["ACGT"], "ValidateSeqLength": 10000, "NonGreedy": False}),
    ("SubseqTransform", lambda: bsk.SeqKitSubseqOptions().Region("1:2"),
     {"Chr": [], "Feature": [], "UpStream": 0, "GtfTag": "", "Region": "1:2"}),
    ("Translate", bsk.SeqKitTranslateOptions, {"TranslTable": 1, "Frame": ["1"], "ListTranslTable": -1}),
    ("RmDup", bsk.SeqKitRmDupOptions, {"BySeq": False, "DupNumFile": ""}),
])
def test_defaults_of_every_hot_path_command(name, cls, probe):
    o = cls()
    op = bsk.Operator(name, o.to_json(), -1)
    d = json.loads(op.opts_json())
    for k, v in probe.items():
        assert d[k] == v
    assert list(d)[0] == "Config" and d["Config"]["LineWidth"] == 60
    # field order == Go declaration order
    assert list(d)[1:] == list(o._fields)


def test_idncbi_overrides_idregexp():
    op = bsk.Operator("Stats", bsk.SeqKitStatsOptions().Config(bsk.SeqKitConfig().IDNCBI(True)).to_json(), -1)
    assert json.loads(op.opts_json())["Config"]["IDRegexp"] == r"\|([^\|]+)\| "


@pytest.mark.parametrize("opts,msg", [
    ({"GapLetters": ""}, "value of flag -G (--gap-letters) should not be empty"),
    ({"GapLetters": "é"}, "value of -G (--gap-letters) contains non-ASCII characters"),
    ({"Config": {"SeqType": "dnaa"}}, "invalid sequence type: dnaa, available value: dna|rna|protein|unlimit|auto"),
    ({"FqEncoding": "phred"}, "unsupported quality encoding: phred"),
])
def test_before_error_texts(opts, msg):
    with pytest.raises(bsk.BskError) as e:
        bsk.Operator("Stats", json.dumps(opts), -1)
    assert e.value.code == _lib.BSK_ERR_OPTS
    assert msg in str(e.value)
    with pytest.raises(oracle.OracleError) as oe:
        oracle.stats_map(b"@a\nA\n+\nI\n", True, json.dumps(opts))
    assert msg in str(oe.value)


def test_unknown_operator_and_bad_json():
    with pytest.raises(bsk.BskError):
        bsk.Operator("Nope", "{}", -1)
    with pytest.raises(bsk.BskError, match="invalid options JSON"):
        bsk.Operator("Stats", '{"All": tru}', -1)
    with pytest.raises(bsk.BskError, match="must be a bool"):
        bsk.Operator("Stats", '{"All": 1}', -1)


def _first_start(data, fmt, frm):
    out = C.c_size_t()
    buf = (C.c_char * max(1, len(data))).from_buffer_copy(data.ljust(1, b"\0"))
    assert lib.bsk_find_record_start(C.cast(buf, C.c_void_p), len(data), frm, fmt, C.byref(out)) == 0
    return out.value


@pytest.mark.parametrize("fastq", [True, False])
def test_record_boundary_repair_matches_oracle_spans(fastq):
    rng = random.Random(11)
    for trial in range(6):
        if fastq:
            data = seqgen.random_fastq(rng, 40, 0, 60, final_newline=trial % 2 == 0)
        else:
            data = seqgen.random_fasta(rng, 40, 0, 150, width=[60, 7, 0][trial % 3], final_newline=trial % 2 == 0,
                                       gt_in_header=True)
        starts = [s for s, _ in oracle.record_spans(data, fastq)]
        for frm in range(0, len(data) + 1, 3):
            want = next((s for s in starts if s >= frm), len(data))
            assert _first_start(data, int(fastq), frm) == want, (trial, frm)


def test_synth_host_is_wellformed_and_seeded():
    n = 1000
    for kind, fastq, seqlen in [(0, True, 150), (1, False, 1000), (2, False, 5001)]:
        rb = lib.bsk_synth_record_bytes(kind)
        nb = rb * (n if kind == 0 else 50)
        buf = C.create_string_buffer(nb)
        assert lib.bsk_synth_host(kind, 42, 0, 0, buf, nb) == 0
        data = buf.raw
        nrec = nb // rb
        assert oracle.count_records(data, fastq) == nrec
        assert oracle.stats_map(data, fastq) == {seqlen: nrec, -4: ord("D")}
        # a shard generated from record 17 equals the matching slice of the whole file
        part = C.create_string_buffer(3 * rb + 5)
        assert lib.bsk_synth_host(kind, 42, 0, 17, part, len(part)) == 0
        assert part.raw == data[17 * rb:20 * rb + 5]
        other = C.create_string_buffer(rb)
        lib.bsk_synth_host(kind, 43, 0, 0, other, rb)
        assert other.raw != data[:rb]
    assert data.startswith(b">cds00000000 len=5001\nATG")
    rec = data[:5107].split(b"\n", 1)[1].replace(b"\n", b"")
    assert rec.endswith(b"TAA") and len(rec) == 5001
    codons = [rec[i:i + 3] for i in range(3, 4998, 3)]
    assert not ({b"TAA", b"TAG", b"TGA"} & set(codons))


def test_synth_flags_motif_and_dups():
    rb, n = 317, 2000
    buf = C.create_string_buffer(rb * n)
    lib.bsk_synth_host(0, 42, _lib.SYNTH_FLAG_MOTIF | _lib.SYNTH_FLAG_DUPS, 0, buf, rb * n)
    recs = [buf.raw[i * rb:(i + 1) * rb] for i in range(n)]
    seqs = [r.split(b"\n")[1] for r in recs]
    assert all(b"ACGTTGCAAGCT" in seqs[i] for i in range(0, n, 100))
    assert all(b"AGCTTGCAACGT" in seqs[i] for i in range(50, n, 100))
    first = {}
    dups = 0
    for i, s in enumerate(seqs):
        if s in first:
            dups += 1
        else:
            first[s] = i
    assert abs(dups - n // 5) <= 2 + n // 100  # 20 % duplicates (motif records may differ)


def _product_string(m, opts):
    op = bsk.Operator("Stats", json.dumps(opts), -1)
    ks = sorted(m)
    keys, vals = (C.c_int64 * len(ks))(*ks), (C.c_int64 * len(ks))(*[m[k] for k in ks])
    info = _lib.StatInfo()
    assert lib.bsk_stats_finalize(op.ctx, keys, vals, len(ks), C.byref(info)) == 0
    out = C.create_string_buffer(1 << 16)
    assert lib.bsk_stats_string(op.ctx, b"input0", b"N/A", C.byref(info), out, len(out)) == 0
    return out.value.decode()


@pytest.mark.parametrize("tabular", [True, False])
def test_driver_side_finalise_and_format_match_oracle(tabular):
    rng = random.Random(5)
    first = b"@a\nACGT\n+\nIIII"
    for trial in range(40):
        k = rng.randint(1, 12)
        m = {rng.randint(0, 5000) if trial % 3 else rng.randint(0, 9): rng.randint(1, 10 ** rng.randint(0, 7))
             for _ in range(k)}
        total = sum(a * b for a, b in m.items())
        m[-1] = rng.randint(0, max(total, 1))
        m[-2] = rng.randint(0, m[-1])
        m[-3] = rng.randint(0, 1000)
        m[-4] = ord("DR"[trial % 2])
        opts = {"All": True, "Tabular": tabular}
        assert _product_string(m, opts) == oracle.stats_string_from_map(m, first, json.dumps(opts)), m
        opts = {"All": False, "Tabular": tabular}
        assert _product_string(m, opts) == oracle.stats_string_from_map(m, first, json.dumps(opts)), m
    # no records at all
    assert _product_string({-4: ord("U")}, {"Tabular": True, "All": True}) == \
        oracle.stats_string_from_map({-4: ord("U")}, b"", '{"Tabular": true, "All": true}')


def test_stats_merge_sums_like_oracle_reduce():
    a = {150: 3, 10: 1, -1: 5, -3: 0, -4: ord("D")}
    b = {150: 4, 7: 2, -1: 1, -2: 9, -3: 2, -4: ord("R")}
    ka, va = (C.c_int64 * len(a))(*a), (C.c_int64 * len(a))(*a.values())
    kb, vb = (C.c_int64 * len(b))(*b), (C.c_int64 * len(b))(*b.values())
    ko, vo, n = (C.c_int64 * 16)(), (C.c_int64 * 16)(), C.c_size_t()
    assert lib.bsk_stats_merge(ka, va, len(a), kb, vb, len(b), ko, vo, 16, C.byref(n)) == 0
    got = dict(zip(ko[:n.value], vo[:n.value]))
    assert got == {150: 7, 10: 1, 7: 2, -1: 6, -2: 9, -3: 2, -4: ord("D")}


def test_ctx_set_knows_its_switches_and_refuses_others():
    """bsk_ctx_set: run-time switches per context (INTEGRATION.md "Switches"); the environment only supplies defaults at
    bsk_create.  Options-only context: no device needed."""
    import ctypes as C
    from bigseqkit_amd._lib import lib
    ctx = C.c_void_p()
    assert lib.bsk_create(b"RmDup", b'{"BySeq": true}', -1, C.byref(ctx)) == 0
    try:
        for key in (b"segcopy", b"BSK_SEGCOPY", b"rmdup_keys", b"min_range_bytes", b"long_bytes", b"stage_bytes", b"translate"):
            assert lib.bsk_ctx_set(ctx, key, b"1") == 0, key
            assert lib.bsk_ctx_set(ctx, key, None) == 0, key
        assert lib.bsk_ctx_set(ctx, b"no_such_switch", b"1") != 0
        assert b"unknown switch" in lib.bsk_last_error(ctx)
    finally:
        lib.bsk_destroy(ctx)
