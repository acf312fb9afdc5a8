"""The deferred whole-record sinks of the streaming skeleton (csrc/stream_core_dev.hpp: sink_records4 -- k_stats, k_names,
k_index on FASTQ): the newline events of several tiles wait in an LDS window and the sink takes 64 .. 128 WHOLE records at
a time.  What the window must survive: more newlines in one tile than it holds (records of a few bytes: the per-event rules
take over in mid-tile), ranges of one tile, records that end exactly at a range or window boundary, a file that stops inside
its last quality line, a truncated last record.  Everything against the oracle (SeqParser.Read + the operators,
/root/reference/bigseqkit-lib/helper.go:219-325, stats.go:48-117, seq.go:81-269)."""
import ctypes as C
import json
import random

import pytest

import oracle
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

pytestmark = pytest.mark.gpu


def run_seq(data, opts):
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    with bsk.Operator("SeqTransform", json.dumps(opts), 0) as op:
        out = _lib.Out()
        check(lib.bsk_seq_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out)), op.ctx)
        buf = C.create_string_buffer(max(1, out.len))
        check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
        return buf.raw[:out.len], out.records


def stats_row(data, all_):
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    o = bsk.SeqKitStatsOptions().Tabular(True).All(all_)
    return bsk.StatsString("x", "N/A", bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), o)


def dense(rng, n, lo, hi, ids=300):
    out = []
    for k in range(n):
        L = rng.randint(lo, hi)
        s = "".join(rng.choice("ACGTN") for _ in range(L))
        q = "".join(chr(rng.randint(33, 73)) for _ in range(L))
        out.append("@r%d n%d\n%s\n+\n%s\n" % (rng.randrange(ids), k, s, q))
    return "".join(out).encode()


@pytest.mark.parametrize("min_range", ["4096", "65536"])
@pytest.mark.parametrize("shape", ["len4", "len1_9", "len0_3", "len30_40", "len150", "mixed"])
def test_more_newlines_than_the_window_holds(shape, min_range, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", min_range)
    rng = random.Random(len(shape) + int(min_range))
    lo, hi, n = {"len4": (4, 4, 2000), "len1_9": (1, 9, 3000), "len0_3": (0, 3, 3000), "len30_40": (30, 40, 1500),
                 "len150": (150, 150, 1200), "mixed": (0, 400, 900)}[shape]
    data = dense(rng, n, lo, hi)
    for all_ in (False, True):
        assert stats_row(data, all_) == oracle.stats_string(data, True, json.dumps({"Tabular": True, "All": all_}), name="x")
    for opts in ({"Name": True}, {"Name": True, "OnlyId": True}, {"Reverse": True}, {"Seq": True, "MinLen": 2}):
        got, nrec = run_seq(data, opts)
        want = oracle.seq(data, True, json.dumps(opts))
        assert got == want and nrec == want.count(b"\n") // (4 if not (opts.get("Name") or opts.get("Seq")) else 1)


@pytest.mark.parametrize("cut", ["no_final_newline", "empty_last_quality", "after_plus", "after_bases", "after_header", "mid_quality_short"])
def test_the_last_record_of_a_file(cut, monkeypatch):
    """the virtual newline of a file that stops inside its last quality line completes that record inside the window; a
    truncated record takes the per-event rules and fails with the reference's kind of error"""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(len(cut))
    for n in (1, 3, 63, 64, 65, 127, 128, 129, 700):
        body = dense(rng, n, 5, 60)
        last = b"@last one\nACGTACGT\n+\nIIIIIIII\n"
        if cut == "no_final_newline": data = body + last[:-1]
        elif cut == "empty_last_quality": data = body + b"@e\n\n+\n"
        elif cut == "after_plus": data = body + last[:last.index(b"+") + 2]
        elif cut == "after_bases": data = body + last[:last.index(b"+")]
        elif cut == "after_header": data = body + last[:last.index(b"\n") + 1]
        else: data = body + last[:-4]
        if not oracle.is_strict_4line_fastq(data):   # (a record cut short: an error, never a different answer)
            for fn in (lambda: stats_row(data, True), lambda: stats_row(data, False), lambda: run_seq(data, {"Name": True}),
                       lambda: run_seq(data, {"Reverse": True})):
                with pytest.raises(_lib.BskError) as e:
                    fn()
                assert e.value.code in (_lib.BSK_ERR_FORMAT, _lib.BSK_ERR_UNSUPPORTED)
        else:
            assert stats_row(data, True) == oracle.stats_string(data, True, json.dumps({"Tabular": True, "All": True}), name="x")
            for opts in ({"Name": True}, {"Reverse": True}):
                assert run_seq(data, opts)[0] == oracle.seq(data, True, json.dumps(opts))
