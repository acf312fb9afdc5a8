"""`translate` on FASTA whose records all look alike (UniformLayout, csrc/ops_translate.hpp): no '>' pass, no record table --
record i begins at i * S, and k_translate_wide<G, true> verifies every byte of every record against the layout the host
proposed from the head of the shard.  One record that differs sends the call through the table paths.  Either way the
output is the oracle's (Translate.Call, /root/reference/bigseqkit-lib/translate.go:104-145)."""
import ctypes as C
import json
import random

import pytest

import oracle
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

pytestmark = pytest.mark.gpu


def translate(data, opts, sets=(), calls=2):
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    with bsk.Operator("Translate", json.dumps(opts), 0) as op:
        for k, v in sets:
            check(lib.bsk_ctx_set(op.ctx, k, v), op.ctx)
        outs = []
        for _ in range(calls):
            lib.bsk_profile_reset(op.ctx)
            lib.bsk_profile_enable(op.ctx, 1)
            out = _lib.Out()
            check(lib.bsk_translate_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTA, 0, None, C.byref(out)), op.ctx)
            buf = C.create_string_buffer(max(1, out.len))
            check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
            pb = C.create_string_buffer(4096)
            check(lib.bsk_profile_dump(op.ctx, pb, len(pb)), op.ctx)
            outs.append((buf.raw[:out.len], pb.value.decode(), out.records))
        return outs


def uniform(rng, nrec, L, W, name="r%05d some text", alphabet="ACGT", final_newline=True):
    recs = []
    for i in range(nrec):
        s = "".join(rng.choice(alphabet) for _ in range(L))
        body = s + "\n" if W == 0 else "".join(s[j:j + W] + "\n" for j in range(0, L, W))
        recs.append(">" + (name % i) + "\n" + body)
    text = "".join(recs)
    return (text if final_newline else text[:-1]).encode()


def synth(kind, nrec):
    rb = lib.bsk_synth_record_bytes(kind)
    buf = (C.c_uint8 * (rb * nrec))()
    check(lib.bsk_synth_host(kind, 42, 0, 0, buf, rb * nrec))
    return bytes(buf)


OPTS = [{"Frame": ["6"]}, {"Frame": ["1"], "Config": {"LineWidth": 0}}, {"Frame": ["-2", "3"], "Clean": True},
        {"Frame": ["6"], "TranslTable": 11, "AllowUnknownCodon": True, "Config": {"LineWidth": 70}}]
SHAPES = {
    "cds5k": lambda rng: synth(_lib.SYNTH_FASTA5K_CDS, 120),          # the C4 layout: 5 001 bases wrapped at 60, a wave per record
    "fasta1k": lambda rng: synth(_lib.SYNTH_FASTA1K, 500),             # 1 000 bases, 16 lanes per record
    "w70_L333": lambda rng: uniform(rng, 700, 333, 70),                # 4 lanes per record
    "one_line": lambda rng: uniform(rng, 600, 451, 0),
    "exact_lines": lambda rng: uniform(rng, 300, 600, 60),             # the last line is a full one
    "no_final_newline": lambda rng: uniform(rng, 300, 1234, 80, final_newline=False),
    "lower": lambda rng: uniform(rng, 300, 777, 60, alphabet="ACGTacgt"),
    "two_records": lambda rng: uniform(rng, 2, 4000, 100),
}


@pytest.mark.parametrize("o", range(len(OPTS)))
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_uniform_records_need_no_table(shape, o):
    data = SHAPES[shape](random.Random(len(shape) * 7 + o))
    want = oracle.translate(data, False, json.dumps(OPTS[o]))
    for got, stages, records in translate(data, OPTS[o]):
        assert got == want
        assert records == want.count(b">")
        assert "k_translate_uniform=" in stages and "k_fasta_starts" not in stages and "k_index" not in stages, stages


def _lines_of(rec_lines, widths):
    s = b"".join(rec_lines)
    out, at = [], 0
    for w in widths:
        out.append(s[at:at + w])
        at += w
    assert at == len(s)
    return out


@pytest.mark.parametrize("why", ["N", "break_moved", "header_longer", "header_break", "marker_gone", "two_breaks", "tab_in_line",
                                 "final_break_gone", "last_record_short", "extra_record_kind"])
@pytest.mark.parametrize("where", ["early", "late"])
def test_one_record_that_differs_sends_the_call_to_the_tables(why, where):
    """the probe looks at the first 256 KiB, the kernel at everything: a record that is not like record 0 -- at the same
    stride or not -- must not be translated from the proposed layout"""
    rng = random.Random(len(why) * 3 + len(where))
    nrec, L, W = 900, 420, 60
    data = uniform(rng, nrec, L, W)
    S = len(data) // nrec
    k = 5 if where == "early" else nrec - 7           # (early: inside the head sample; late: far behind it)
    rec = bytearray(data[k * S:(k + 1) * S])
    H = rec.index(b"\n")
    if why == "N":
        rec[H + 1 + 200] = ord("N")
    elif why == "break_moved":                          # lines of 60 / 50 / 70 / ...: same bytes, same stride, same length
        lines = bytes(rec[H + 1:]).split(b"\n")[:-1]
        widths = [len(x) for x in lines]
        widths[1] -= 10
        widths[2] += 10
        rec = bytearray(rec[:H + 1] + b"\n".join(_lines_of(lines, widths)) + b"\n")
    elif why == "header_longer":                        # one byte more in the header, one base less: same stride
        rec = bytearray(rec[:H] + b"x" + rec[H:-2] + b"\n")
    elif why == "header_break":                         # a line break inside what should be the header
        rec[3] = ord("\n")
    elif why == "marker_gone":
        rec[0] = ord("A")
    elif why == "two_breaks":                           # a base replaced by a second line break inside a window
        rec[H + 1 + 30] = ord("\n")
    elif why == "tab_in_line":
        rec[H + 1 + 100] = ord("\t")
    elif why == "final_break_gone":                     # the record's last byte is a base, the next '>' follows directly
        rec[-1] = ord("A")
    data2 = data[:k * S] + bytes(rec) + data[(k + 1) * S:]
    if why == "last_record_short":                      # the last record lacks three bases (and the size is no multiple of S)
        data2 = data[:-4] + b"\n"
    elif why == "extra_record_kind":                    # a whole number of strides, but the last "record" is two short ones
        la = S // 2
        a = b">a\n" + (b"ACGT" * 1000)[:la - 4] + b"\n"
        b2 = b">b\n" + b"C" * (S - la - 4) + b"\n"
        assert len(a) + len(b2) == S
        data2 = data + a + b2
    opts = {"Frame": ["6"]}
    try:
        want = oracle.translate(data2, False, json.dumps(opts))
    except oracle.OracleError as e:   # (text the reference itself rejects: the HIP path must reject it too)
        with pytest.raises(_lib.BskError):
            translate(data2, opts, calls=1)
        return
    outs = translate(data2, opts)
    assert outs[0][0] == want and outs[1][0] == want
    if where == "late" and why != "last_record_short":
        # the probe saw nothing wrong: the uniform pass ran, did not verify, and the call started over on the tables
        assert "k_translate_uniform=" in outs[0][1], outs[0][1]
        assert "k_translate_uniform=" not in outs[1][1], outs[1][1]     # the context does not try again
    assert "k_fasta_starts" in outs[0][1] or "k_index" in outs[0][1], outs[0][1]


def test_switches_keep_the_table_paths():
    data = synth(_lib.SYNTH_FASTA5K_CDS, 60)
    want = oracle.translate(data, False, json.dumps({"Frame": ["6"]}))
    for sets, stage in (((( b"translate_index", b"light"),), "k_fasta_starts"), (((b"translate_index", b"full"),), "k_index=")):
        for got, stages, _ in translate(data, {"Frame": ["6"]}, sets):
            assert got == want and stage in stages, stages
    # options whose output sizes depend on the text (--trim), or that need the ID split (-F), or -M: not this pass
    for extra in ({"Trim": True}, {"AppendFrame": True}, {"InitCodonAsM": True}):
        opts = dict({"Frame": ["6"]}, **extra)
        want = oracle.translate(data, False, json.dumps(opts))
        for got, stages, _ in translate(data, opts):
            assert got == want and ("k_fasta_starts" in stages or "k_index" in stages) and "uniform" not in stages, (extra, stages)
