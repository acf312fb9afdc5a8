"""Parity of the HIP `stats` path against the CPU oracle -- through the C ABI, on a GPU.
Integer work: every map entry must be bit-exact; the driver-side text must be identical."""
import ctypes as C
import json
import os
import random

import pytest

import oracle
import seqgen
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib

pytestmark = pytest.mark.gpu


def dev(data):
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8) if len(data) else torch.empty(0, dtype=torch.uint8)
    return t.cuda()


def gpu_map(data, fastq, opts, on_device=True):
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    frame = bsk.SeqFrame(fmt, [dev(data) if on_device else data])
    m, op = bsk.stats_map(frame, _Opts(opts))
    op.close()
    return m


class _Opts:
    def __init__(self, d):
        self.d = d

    def to_json(self):
        return json.dumps(self.d)


def check_parity(data, fastq, opts=None, **kw):
    opts = opts or {}
    want = oracle.stats_map(data, fastq, json.dumps(opts))
    got = gpu_map(data, fastq, opts, **kw)
    assert got == want


@pytest.mark.parametrize("use_dpp", [1, 0])
def test_wave_scan_primitive(use_dpp):
    rng = random.Random(3)
    for _ in range(4):
        vals = [rng.randint(0, 1 << 20) for _ in range(64)]
        i, o = (C.c_uint32 * 64)(*vals), (C.c_uint32 * 64)()
        assert lib.bsk_selftest_scan(use_dpp, i, o) == 0
        acc, want = 0, []
        for v in vals:
            acc += v
            want.append(acc & 0xFFFFFFFF)
        assert list(o) == want


def test_synth_device_equals_host():
    import torch
    for kind in (0, 1, 2):
        rb = lib.bsk_synth_record_bytes(kind)
        n = rb * 1000 + 123
        t = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert lib.bsk_synth_device(kind, 42, 3, 5, C.c_void_p(t.data_ptr()), n, 0, None) == 0
        torch.cuda.synchronize()
        h = C.create_string_buffer(n)
        lib.bsk_synth_host(kind, 42, 3, 5, h, n)
        assert bytes(t.cpu().numpy().tobytes()) == h.raw


HAND = [
    b"@r1 d\nACGT\n+\nIIII\n@r2\nAC-N\n+\n5#5I\n",
    b"@a\nAC\n+\n@+\n@b\nGT\n+a\n+@",                 # '@'/'+' leading quality, no final newline
    b"@a\n\n+\n\n@b\nA\n+\nI\n\n\n",                  # empty sequence, trailing blank lines
    b"@a\n\n+\n",                                     # ends inside an empty quality line
    b"@only\nACGTACGT\n+\nIIIIIIII",
    b"",
]


@pytest.mark.parametrize("all_", [False, True])
@pytest.mark.parametrize("i", range(len(HAND)))
def test_fastq_hand_cases(i, all_):
    check_parity(HAND[i], True, {"All": all_})


HAND_FA = [
    b">s1 a>b\nAC\nGT\n>s2\n\n>s3",
    b">x\nACGT\n",
    b">x\nAC-GT\nA. T\n>y\n>z\nNNNN",
    b">p\nMKVLAAGIVGLLLAQ\n",
    b">r\nACGUACGU\n",
    b">lonely",
    b">a\n\n\n\n",
]


@pytest.mark.parametrize("all_", [False, True])
@pytest.mark.parametrize("i", range(len(HAND_FA)))
def test_fasta_hand_cases(i, all_):
    check_parity(HAND_FA[i], False, {"All": all_})


@pytest.mark.parametrize("all_", [False, True])
@pytest.mark.parametrize("seed", range(6))
def test_fastq_random_parity(seed, all_, monkeypatch):
    # tiny ranges: many anchors, records straddling every range and tile boundary
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", str([256, 1000, 4096, 65536][seed % 4]))
    rng = random.Random(seed)
    data = seqgen.random_fastq(rng, 3000, 0, [40, 300, 3000][seed % 3], final_newline=seed % 2 == 0,
                               trailing_blank=seed % 3)
    assert oracle.is_strict_4line_fastq(data)
    check_parity(data, True, {"All": all_})


@pytest.mark.parametrize("all_", [False, True])
@pytest.mark.parametrize("seed", range(6))
def test_fasta_random_parity(seed, all_, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", str([256, 1000, 4096, 65536][seed % 4]))
    rng = random.Random(100 + seed)
    data = seqgen.random_fasta(rng, 1500, 0, [100, 1500, 9000][seed % 3], width=[60, 70, 0, 13][seed % 4],
                               final_newline=seed % 2 == 0, trailing_blank=seed % 3, gt_in_header=True)
    check_parity(data, False, {"All": all_})


def test_shfl_scan_variant_gives_same_result(monkeypatch):
    rng = random.Random(9)
    data = seqgen.random_fastq(rng, 2000, 0, 200)
    monkeypatch.setenv("BSK_SCAN", "shfl")
    check_parity(data, True, {"All": True})


def test_newline_dense_input_takes_several_event_batches(monkeypatch):
    # more than 128 newlines per 4 KiB tile: records of 1-base reads
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "2048")
    rng = random.Random(4)
    data = seqgen.random_fastq(rng, 5000, 0, 2, name_space=False)
    check_parity(data, True, {"All": True})
    fa = b"".join(b">%d\nA\n\nC\n" % i for i in range(3000))
    check_parity(fa, False, {"All": True})


def test_long_reads_and_histogram_overflow():
    rng = random.Random(8)
    recs = []
    for L in [2047, 2048, 2049, 65535, 65536, 70000, 200000, 5, 200000]:
        s = "".join(rng.choice("ACGT") for _ in range(L))
        recs.append(f"@r{L}\n{s}\n+\n{'I' * L}\n")
    data = "".join(recs).encode()
    check_parity(data, True, {"All": True})
    fa = b">chr1 one line\n" + b"ACGT" * 100000 + b"\n>chr2\n" + (b"ACGTN" * 12 + b"\n") * 5000
    check_parity(fa, False, {"All": True})


def test_gap_letters_and_encoding_options():
    rng = random.Random(12)
    data = seqgen.random_fastq(rng, 500, 1, 100, qual_lo=64, qual_hi=104)
    check_parity(data, True, {"All": True, "FqEncoding": "solexa"})
    check_parity(data, True, {"All": True, "FqEncoding": "illumina-1.8+", "GapLetters": "N-"})
    check_parity(data, True, {"All": True, "Config": {"SeqType": "dna"}})


@pytest.mark.parametrize("gaps", ["-.*NnXx ~", "ACGTNacgtn-.", "".join(chr(c) for c in range(33, 127))])
def test_more_than_eight_gap_letters(gaps, monkeypatch):
    """the reference counts any number of gap letters (stats.go:36-43, 102); beyond the eight the streaming pass keeps in
    registers they are counted by a pass over the record table (round 4; before: refused)"""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(len(gaps))
    fq = seqgen.random_fastq(rng, 900, 0, 150, alphabet="ACGTNacgtn-.*Xx")
    fa = seqgen.random_fasta(rng, 200, 0, 900, width=60, alphabet="ACGTNacgtn-.*Xx")
    for on_device in (True, False):
        check_parity(fq, True, {"All": True, "GapLetters": gaps}, on_device=on_device)
        check_parity(fa, False, {"All": True, "GapLetters": gaps}, on_device=on_device)
    check_parity(fq, True, {"GapLetters": gaps})          # without -a nothing is counted
    one_line = b">chr1 one line\n" + b"ACGT-N.x" * 60000 + b"\n>chr2\n" + (b"AC-TN" * 12 + b"\n") * 3000
    check_parity(one_line, False, {"All": True, "GapLetters": gaps})


def test_host_resident_shard():
    rng = random.Random(13)
    data = seqgen.random_fastq(rng, 800, 0, 150)
    check_parity(data, True, {"All": True}, on_device=False)


def test_partitions_are_reduced():
    rng = random.Random(14)
    data = seqgen.random_fastq(rng, 1000, 0, 150)
    frame = bsk.ReadFASTQN(data, 5)
    assert len(frame.shards) >= 4
    m, op = bsk.stats_map(frame, bsk.SeqKitStatsOptions().All(True))
    op.close()
    assert m == oracle.stats_map(data, True, '{"All": true}', nparts=5)


@pytest.mark.parametrize("data,code,msg", [
    (b"@a\nACGT\n+\nIII\n@b\nA\n+\nI\n", _lib.BSK_ERR_FORMAT, "unmatched length"),
    (b"@a\nACGT\nAC\n+\nIIII\nIII\n", _lib.BSK_ERR_FORMAT, None),               # multi-line FASTQ, quality too long
    (b"\n@a\nACGT\n+\nIIII\n", _lib.BSK_ERR_UNSUPPORTED, "does not start"),
    (b"@a\nACGT\n+\nIIII\n@b\nAC\n", _lib.BSK_ERR_FORMAT, None),               # truncated
])
def test_malformed_fastq_is_an_error_not_a_wrong_answer(data, code, msg):
    assert not oracle.is_strict_4line_fastq(data)
    with pytest.raises(bsk.BskError) as e:
        gpu_map(data, True, {"All": True})
    assert e.value.code in (code, _lib.BSK_ERR_FORMAT, _lib.BSK_ERR_UNSUPPORTED)
    if msg:
        assert msg in str(e.value)


def test_stats_string_end_to_end():
    rng = random.Random(15)
    data = seqgen.random_fastq(rng, 1234, 0, 250)
    for opts in ({"All": True, "Tabular": True}, {"All": True}, {"Tabular": True}, {}):
        o = bsk.SeqKitStatsOptions()
        for k, v in opts.items():
            getattr(o, k)(v)
        got = bsk.StatsString("input0", "N/A", bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(data)]), o)
        assert got == oracle.stats_string(data, True, json.dumps(opts))
    prot = b">p1\nMKVLAAGIVGLLLAQW\n>p2\nMKV\n"
    got = bsk.StatsString("input0", "N/A", bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(prot)]),
                          bsk.SeqKitStatsOptions().Tabular(True))
    assert got == oracle.stats_string(prot, False, '{"Tabular": true}')


def test_synthetic_fastq150_full_size_properties():
    """BASELINE C2 layout at a size the oracle cannot touch: size-independent properties."""
    import torch
    rb = 317
    nrec = 20_000_000  # 6.3 GB in HBM
    n = rb * nrec - 1  # drop the final newline
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.bsk_synth_device(0, 42, 0, 0, C.c_void_p(t.data_ptr()), n, 0, None) == 0
    m, op = bsk.stats_map(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), bsk.SeqKitStatsOptions().All(True))
    op.close()
    assert m[150] == nrec and m[-3] == 0 and m[-4] == ord("D")
    assert set(m) == {150, -1, -2, -3, -4}
    # qual = '#' + (u8 * 39 >> 8): Phred >= 20 <=> u8 >= 119, Phred >= 30 <=> u8 >= 184
    tot = 150 * nrec
    assert abs(m[-1] / tot - 137 / 256) < 1e-4
    assert abs(m[-2] / tot - 72 / 256) < 1e-4
    # linearity: a prefix shard and the rest sum to the whole
    cut = rb * 7_000_001
    a, op = bsk.stats_map(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t[:cut], t[cut:]]), bsk.SeqKitStatsOptions().All(True))
    op.close()
    assert a == m
    # the oracle agrees on a 10 MB prefix
    head = bytes(t[:rb * 30000].cpu().numpy().tobytes())
    g = gpu_map(head, True, {"All": True})
    assert g == oracle.stats_map(head, True, '{"All": true}')


def test_stats_host_shard_chunked_pipeline_matches_device_path(monkeypatch):
    """on_device = 0: record-aligned chunks through two device buffers (copy of chunk i+1 overlaps the kernels of
    chunk i).  Tiny chunks force many cuts, in pinned (bsk_host_alloc) and pageable memory."""
    import ctypes as C
    import json
    import random
    import oracle
    import seqgen
    from bigseqkit_amd._lib import lib, check
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    monkeypatch.setenv("BSK_STAGE_BYTES", "30000")
    for fastq in (True, False):
        rng = random.Random(91 + fastq)
        data = seqgen.random_fastq(rng, 4000, 0, 200) if fastq else seqgen.random_fasta(rng, 1500, 0, 900)
        want = oracle.stats_map(data, fastq, json.dumps({"All": True}))
        p = lib.bsk_host_alloc(len(data))
        assert p
        try:
            C.memmove(p, data, len(data))
            for ptr in (p, C.cast(C.create_string_buffer(data, len(data)), C.c_void_p).value):
                with bsk.Operator("Stats", json.dumps({"All": True}), 0) as op:
                    keys, vals, n = (C.c_int64 * 8192)(), (C.c_int64 * 8192)(), C.c_size_t()
                    check(lib.bsk_stats_run(op.ctx, C.c_void_p(ptr), len(data), 0, bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA,
                                            0, None, None), op.ctx)
                    check(lib.bsk_stats_collect(op.ctx, None, keys, vals, 8192, C.byref(n)), op.ctx)
                    assert dict(zip(keys[:n.value], vals[:n.value])) == want
        finally:
            lib.bsk_host_free(p)


@pytest.mark.parametrize("all_", [False, True])
@pytest.mark.parametrize("final_newline", [True, False])
def test_fasta_records_much_longer_than_a_range(all_, final_newline, monkeypatch):
    """FASTA ranges begin on line starts, so a chromosome-sized record is cut into many ranges whose parts are added up
    by k_stats_stitch: lengths, record count, gap sums and the histogram overflow list must still be exact."""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(77 + all_)
    recs = []
    for k, L in enumerate([3, 250_000, 0, 70_001, 59, 60, 61, 1_200_000, 17, 66_000, 5000, 5000, 5000]):
        s = "".join(rng.choice("ACGTN-.") for _ in range(L))
        w = [60, 70, 80][k % 3]
        recs.append(f">chr{k} len={L}\n" + "".join(s[j:j + w] + "\n" for j in range(0, L, w)))
    data = "".join(recs)
    if not final_newline:
        data = data[:-1]
    check_parity(data.encode(), False, {"All": all_})


def test_text_that_is_not_fastq_fails_fast():
    """FASTA handed over with the FASTQ flag holds no FASTQ record start: the anchor search of every range boundary is
    bounded (anchor.hpp ANCHOR_SEARCH_BYTES) and the run ends with an error -- it used to walk to the end of the shard from
    every boundary (a quarter of an hour for 10 GB)."""
    import time
    import ctypes as C
    import torch
    rb, nrec = 5107, 40000   # ~200 MB of FASTA-5k
    t = torch.empty(rb * nrec, dtype=torch.uint8, device="cuda")
    assert _lib.lib.bsk_synth_device(2, 42, 0, 0, C.c_void_p(t.data_ptr()), rb * nrec, 0, None) == 0
    t0 = time.time()
    with pytest.raises(bsk.BskError):
        bsk.stats_map(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), bsk.SeqKitStatsOptions())
    with pytest.raises(bsk.BskError):
        bsk.Seq(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), bsk.SeqKitSeqOptions().Name(True))
    assert time.time() - t0 < 60


@pytest.mark.parametrize("gaps", ["- .", "N-", "-\x7f", "ACGT"])
@pytest.mark.parametrize("min_range", [256, 4096, 1 << 20])
def test_stats_a_line_role_counts_on_hostile_reads(gaps, min_range, monkeypatch):
    """`stats -a` counts Q20 / Q30 / gap by line role (stream_core_dev.hpp, sink_role_counts): reads with gap letters,
    empty reads (three and more newlines inside 16 bytes), bytes >= 128 in names and qualities, `@` / `+` leading
    qualities, a gap letter above the bases (no prefilter possible) and one at 127 (prefilter switched off)."""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", str(min_range))
    rng = random.Random(sum(map(ord, gaps)) * 31 + min_range)
    recs = []
    for i in range(4000):
        L = rng.choice([0, 0, 1, 2, 7, 15, 16, 17, 31, 33, 64, 150, 151, 400])
        s = bytes(rng.choice(b"ACGTNacgtn-. \x7f") for _ in range(L))
        q = bytes(rng.choice(b"!+5?@IJ~\x80\xff") for _ in range(L))
        name = b"r%d" % i + (b" caf\xc3\xa9 \xff" if i % 7 == 0 else b"")
        recs.append(b"@" + name + b"\n" + s + b"\n+" + (name if i % 11 == 0 else b"") + b"\n" + q + b"\n")
    data = b"".join(recs)
    assert oracle.is_strict_4line_fastq(data)
    check_parity(data, True, {"All": True, "GapLetters": gaps})
    check_parity(data[:-1], True, {"All": True, "GapLetters": gaps})  # no final newline


@pytest.mark.parametrize("seed", range(4))
def test_stats_a_line_roles_equal_the_dense_path(seed, monkeypatch):
    """Two unrelated device implementations of FASTQ `stats -a` -- line roles on the sparse path (default) and running
    counters on the dense path (BSK_STATS_A=dense) -- must give the same map, and the oracle's."""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", str([256, 4096, 65536, 1 << 20][seed]))
    rng = random.Random(40 + seed)
    data = seqgen.random_fastq(rng, 4000, 0, [40, 300, 3000, 150][seed], final_newline=seed % 2 == 0, trailing_blank=seed % 3)
    opts = {"All": True, "GapLetters": ["- .", "N-", "-", "ACGT"][seed]}
    roles = gpu_map(data, True, opts)
    monkeypatch.setenv("BSK_STATS_A", "dense")
    dense = gpu_map(data, True, opts)
    assert roles == dense == oracle.stats_map(data, True, json.dumps(opts))
