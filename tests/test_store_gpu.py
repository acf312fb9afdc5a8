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
