"""bsk_run_to_store: a host-resident partition through the chunked H2D / kernels / D2H + write pipeline into a FileStore.
The file must equal the oracle's output whatever the chunk size (tiny chunks force many record-aligned cuts), for chunked
operators and for the ones that see the whole partition; parts written out of order land in partition order."""
import ctypes as C
import json
import random

import pytest

import oracle
import seqgen
import bigseqkit_amd as bsk
from bigseqkit_amd import dist as bdist
from bigseqkit_amd._lib import lib, check

pytestmark = pytest.mark.gpu


def run_to_file(op_name, opts, parts, fmt, path, merge=1, order=None):
    s = C.c_void_p()
    assert lib.bsk_store_open(str(path).encode(), merge, C.byref(s)) == 0
    tot_b, tot_r = 0, 0
    with bsk.Operator(op_name, json.dumps(opts), 0) as op:
        for k in (order or range(len(parts))):
            data = parts[k]
            buf = C.create_string_buffer(data, len(data)) if data else None
            nb, nr = C.c_uint64(), C.c_uint64()
            check(lib.bsk_run_to_store(op.ctx, buf, len(data), fmt, k, s, k, C.byref(nb), C.byref(nr)), op.ctx)
            tot_b += nb.value
            tot_r += nr.value
    total = C.c_uint64()
    assert lib.bsk_store_close(s, C.byref(total)) == 0
    assert total.value == tot_b
    return tot_b, tot_r


CASES = [("SeqTransform", {"Reverse": True, "Complement": True}, oracle.seq),
         ("SeqTransform", {"Name": True}, oracle.seq),
         ("Grep", {"BySeq": True, "Pattern": ["ACGTTGCAAGCT", "GATTACAGATTA"]}, oracle.grep),
         ("Grep", {"Pattern": ["r1", "r22", "r333"], "InvertMatch": True}, oracle.grep),
         ("SubseqTransform", {"Region": "2:-3"}, oracle.subseq),
         ("Translate", {"Frame": ["6"]}, oracle.translate),
         ("Fq2Fa", {}, oracle.fq2fa),
         ("RmDup", {"BySeq": True}, oracle.rmdup),
         ("Sort", {"ByLength": True}, oracle.sort)]


@pytest.mark.parametrize("stage", ["3000", "50000", "1000000000"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_one_partition_any_chunk_size(case, stage, tmp_path, monkeypatch):
    monkeypatch.setenv("BSK_STAGE_BYTES", stage)
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    op_name, opts, orc = CASES[case]
    rng = random.Random(case)
    recs = []
    for i in range(900):
        L = rng.randint(20, 120)
        s = "".join(rng.choice("ACGT") for _ in range(L))
        if i % 9 == 0: s = s[:5] + "ACGTTGCAAGCT" + s[17:]
        if i % 11 == 0 and i: s = recs[rng.randrange(len(recs))][1]
        recs.append(("r%d" % i, s))
    data = "".join("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)) for n, s in recs).encode()
    want = orc(data, True, json.dumps(opts))
    out = tmp_path / "o.txt"
    nb, nr = run_to_file(op_name, opts, [data], bsk.FORMAT_FASTQ, out)
    assert out.read_bytes() == want and nb == len(want)


def test_locate_header_row_once_and_parts_in_order(tmp_path, monkeypatch):
    monkeypatch.setenv("BSK_STAGE_BYTES", "2000")
    rng = random.Random(4)
    data = seqgen.random_fasta(rng, 120, 30, 200, alphabet="ACGT")
    cuts = bdist.shard_bounds(data, 3, bsk.FORMAT_FASTA)
    parts = [data[a:b] for a, b in cuts]
    opts = {"Pattern": ["ACG"]}
    want = oracle.locate(data, False, json.dumps(opts))
    for order in ([0, 1, 2], [2, 1, 0], [1, 2, 0]):
        out = tmp_path / ("loc%d.tsv" % order[0])
        run_to_file("Locate", opts, parts, bsk.FORMAT_FASTA, out, order=order)
        assert out.read_bytes() == want
    d = tmp_path / "dir"
    run_to_file("Locate", opts, parts, bsk.FORMAT_FASTA, d, merge=0)
    assert b"".join((d / ("part%05d" % k)).read_bytes() for k in range(3)) == want


def test_alphabet_is_guessed_once_per_partition_not_per_chunk(tmp_path, monkeypatch):
    """SeqType auto: the reference guesses the alphabet from the FIRST record of a partition (helper.go:286-291).  A later
    chunk of the file -> file pipeline that starts with a protein-looking record must still be searched as DNA (both
    strands) -- the same answer as bsk_grep_run over the whole shard."""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(77)
    dna = ["".join(rng.choice("ACGT") for _ in range(60)) for _ in range(40)]
    dna[25] = dna[25][:10] + "AATGCCGGTTAC" + dna[25][22:]      # the reverse complement of the pattern: found only as DNA
    recs = [(">d%d" % i, s) for i, s in enumerate(dna)]
    recs.insert(20, (">p", "MKLVWFRESDEQILHPNMKLVWFRESDEQILHPNMKLVWFRESDEQILHPN"))  # guesses as protein
    data = "".join("%s\n%s\n" % r for r in recs).encode()
    opts = {"BySeq": True, "Pattern": ["GTAACCGGCATT"]}
    want = oracle.grep(data, False, json.dumps(opts))
    assert b">d25" in want
    whole = bsk.Grep(bsk.SeqFrame(bsk.FORMAT_FASTA, [data]), bsk.SeqKitGrepOptions().BySeq(True).Pattern(opts["Pattern"]))
    assert whole == want
    # cut so that one chunk begins with the protein record
    at = data.index(b">p\n")
    for stage in (str(at), "700", "100000"):
        monkeypatch.setenv("BSK_STAGE_BYTES", stage)
        out = tmp_path / ("g%s.fa" % stage)
        run_to_file("Grep", opts, [data], bsk.FORMAT_FASTA, out)
        assert out.read_bytes() == want, stage


def test_context_survives_a_failed_call_in_the_middle_of_a_partition(tmp_path, monkeypatch):
    """A chunk that fails (here: malformed FASTQ in the third chunk) must leave the context usable: the two output
    buffers handed back un-aliased, no event leaked -- the next call on the same context writes the right file."""
    monkeypatch.setenv("BSK_STAGE_BYTES", "2000")
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(9)
    good = seqgen.random_fastq(rng, 200, 20, 80)
    lines = good.split(b"\n")
    bad_lines = list(lines)
    bad_lines[4 * 150 + 3] = bad_lines[4 * 150 + 3] + b"II"      # quality longer than the sequence, far into the file
    bad = b"\n".join(bad_lines)
    opts = {"Reverse": True}
    want = oracle.seq(good, True, json.dumps(opts))
    s1, s2, s3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    with bsk.Operator("SeqTransform", json.dumps(opts), 0) as op:
        for k, (data, store, ok) in enumerate(((good, s1, True), (bad, s2, False), (good, s3, True), (good, None, True))):
            path = tmp_path / ("f%d.fq" % k)
            st = C.c_void_p()
            assert lib.bsk_store_open(str(path).encode(), 1, C.byref(st)) == 0
            buf = C.create_string_buffer(data, len(data))
            nb, nr = C.c_uint64(), C.c_uint64()
            rc = lib.bsk_run_to_store(op.ctx, buf, len(data), bsk.FORMAT_FASTQ, 0, st, 0, C.byref(nb), C.byref(nr))
            tot = C.c_uint64()
            lib.bsk_store_close(st, C.byref(tot))
            if ok:
                assert rc == 0, lib.bsk_last_error(op.ctx)
                assert path.read_bytes() == want
            else:
                assert rc != 0 and b"unmatched length" in lib.bsk_last_error(op.ctx)


def test_drain_of_parts_of_several_pieces_in_and_out_of_turn(tmp_path, monkeypatch):
    """parts of a few hundred MB: several 32 MiB pieces per chunk through the three pinned buffers, written at reserved
    offsets when it is the part's turn and kept in memory when it is not; a directory of part files likewise"""
    monkeypatch.setenv("BSK_STAGE_BYTES", str(64 << 20))
    import torch
    rb, nrec = 317, 900_000                                        # 285 MB: several 32 MiB pieces, three chunks
    t = torch.empty(rb * nrec, dtype=torch.uint8, device="cuda")
    from bigseqkit_amd import _lib
    assert lib.bsk_synth_device(0, 7, 0, 0, C.c_void_p(t.data_ptr()), rb * nrec, 0, None) == 0
    data = bytes(t.cpu().numpy().tobytes())
    parts = [data[:rb * 300_000], data[rb * 300_000:rb * 650_000], data[rb * 650_000:]]
    out = tmp_path / "big.fq"
    run_to_file("SeqTransform", {}, parts, bsk.FORMAT_FASTQ, out, order=[1, 0, 2])
    assert out.read_bytes() == data
    d = tmp_path / "dir"
    run_to_file("SeqTransform", {}, parts, bsk.FORMAT_FASTQ, d, merge=0, order=[2, 0, 1])
    assert b"".join((d / ("part%05d" % k)).read_bytes() for k in range(3)) == data
