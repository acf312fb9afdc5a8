"""`translate` on FASTA with the record table from the '>' bytes alone (stream_fasta_light.hip): starts and header lengths
exact, sequence lengths derived from "every line but the last is as long as the first" and VALIDATED by k_translate_wide,
which reads every byte anyway; whatever does not fit sends the call through the full index pass.  Either way the output is
the oracle's (Translate.Call, /root/reference/bigseqkit-lib/translate.go:104-145)."""
import ctypes as C
import json
import random
import zlib

import pytest

import oracle
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

pytestmark = pytest.mark.gpu


def translate(data, opts, sets=()):
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    with bsk.Operator("Translate", json.dumps(opts), 0) as op:
        for k, v in sets:
            check(lib.bsk_ctx_set(op.ctx, k, v), op.ctx)
        outs = []
        for _ in range(2):   # the second call: what the context remembered about the first
            lib.bsk_profile_reset(op.ctx)
            lib.bsk_profile_enable(op.ctx, 1)
            out = _lib.Out()
            check(lib.bsk_translate_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTA, 0, None, C.byref(out)), op.ctx)
            buf = C.create_string_buffer(max(1, out.len))
            check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
            pb = C.create_string_buffer(4096)
            check(lib.bsk_profile_dump(op.ctx, pb, len(pb)), op.ctx)
            outs.append((buf.raw[:out.len], pb.value.decode()))
        return outs


def fasta(rng, nrec, width, lens=(300, 900), alphabet="ACGT", newline_at_end=True):
    out = []
    for i in range(nrec):
        s = "".join(rng.choice(alphabet) for _ in range(rng.randint(*lens)))
        w = width(i) if callable(width) else width
        body = s + "\n" if w == 0 else "".join(s[j:j + w] + "\n" for j in range(0, len(s), w))
        out.append(">r%d some description %d\n%s" % (i, i, body if s else ""))
    text = "".join(out)
    return (text if newline_at_end else text[:-1]).encode()


OPTS = [{"Frame": ["6"]}, {"Frame": ["1"], "Trim": True}, {"Frame": ["-2", "3"], "AppendFrame": True, "Config": {"LineWidth": 0}},
        {"Frame": ["6"], "TranslTable": 11, "Clean": True}]


@pytest.mark.parametrize("o", range(len(OPTS)))
@pytest.mark.parametrize("shape", ["w60", "w70", "one_line", "own_width", "no_final_newline", "lower"])
def test_regular_text_takes_the_light_table(shape, o, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(zlib.crc32(shape.encode()) % 1000 + o)  # (str hashes differ from process to process)
    if shape == "w60": data = fasta(rng, 400, 60)
    elif shape == "w70": data = fasta(rng, 300, 70, lens=(0, 500))
    elif shape == "one_line": data = fasta(rng, 300, 0, lens=(1, 700))
    elif shape == "own_width": data = fasta(rng, 300, lambda i: (50, 60, 80, 0, 101)[i % 5])
    elif shape == "no_final_newline": data = fasta(rng, 200, 60, newline_at_end=False)
    else: data = fasta(rng, 300, 60, alphabet="ACGTacgt")
    want = oracle.translate(data, False, json.dumps(OPTS[o]))
    for got, stages in translate(data, OPTS[o]):
        assert got == want
        assert "k_fasta_starts" in stages and "k_index=" not in stages, stages


@pytest.mark.parametrize("why", ["N_late", "short_line_late", "blank_line", "narrow", "long_record", "switch"])
def test_text_that_does_not_fit_goes_through_the_full_index(why, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    monkeypatch.setenv("BSK_LONG_BYTES", "20000")
    rng = random.Random(len(why))
    data = fasta(rng, 900, 60)                       # > 256 KiB: the head sample sees nothing wrong
    lines = data.split(b"\n")
    sets = ()
    if why == "N_late":
        k = len(lines) - 30
        while lines[k].startswith(b">") or len(lines[k]) < 10: k -= 1
        lines[k] = lines[k][:5] + b"N" + lines[k][6:]
    elif why == "short_line_late":
        k = len(lines) - 40
        while lines[k].startswith(b">") or lines[k + 1].startswith(b">") or len(lines[k]) != 60: k -= 1
        lines[k] = lines[k][:31]                     # a short line in the middle of a record
    elif why == "blank_line":
        k = len(lines) - 50
        while not lines[k].startswith(b">"): k -= 1
        lines.insert(k, b"")                         # a blank line between two records
    elif why == "narrow":
        data = fasta(rng, 600, 20); lines = data.split(b"\n")   # lines too short for the wide kernel
    elif why == "long_record":
        lines.append(b">big"); lines.append(b"ACGT" * 10000)
    else:
        sets = ((b"translate_index", b"full"),)
    data = b"\n".join(lines)
    opts = {"Frame": ["6"]}
    want = oracle.translate(data, False, json.dumps(opts))
    outs = translate(data, opts, sets)
    assert outs[0][0] == want and outs[1][0] == want
    assert "k_index=" in outs[0][1]
    if why != "narrow":   # (the head sample of the narrow file already says no: the light pass is not even tried)
        assert ("k_fasta_starts" in outs[0][1]) == (why != "switch")
    assert "k_fasta_starts" not in outs[1][1]        # the context remembered


@pytest.mark.parametrize("lines_of", [(60, 50, 60, 60), (60, 55, 60, 5), (60, 60, 48, 60, 12), (70, 69, 70, 1)])
def test_a_break_in_the_right_window_but_the_wrong_place_is_not_trusted(lines_of, monkeypatch):
    """lines of 60 / 50 / 60 / 60 bases have the line COUNT and the LENGTH that the light table derives
    from "every line but the last is as long as the first" (60 / 60 / 60 / 50), and every window of k_translate_wide that
    expects a line break holds one -- a few bytes off.  The window check must look at the PLACE of the break: such a
    record goes through the full index pass and comes out as SeqParser joins it (helper.go:252-283)."""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(sum(lines_of))
    data = fasta(rng, 900, lines_of[0], lens=(300, 900))   # a regular file around the records in question
    recs = []
    for i in range(40):
        s = "".join(rng.choice("ACGT") for _ in range(sum(lines_of)))
        body, at = "", 0
        for w in lines_of:
            body += s[at:at + w] + "\n"
            at += w
        recs.append(">odd%d\n%s" % (i, body))
    data = data + "".join(recs).encode() + fasta(rng, 50, lines_of[0])
    opts = {"Frame": ["6"]}
    want = oracle.translate(data, False, json.dumps(opts))
    outs = translate(data, opts)
    assert outs[0][0] == want and outs[1][0] == want
    assert "k_fasta_starts" in outs[0][1] and "k_index=" in outs[0][1]   # tried light, fell back
    assert "k_fasta_starts" not in outs[1][1]


def test_a_shard_below_the_wide_kernels_minimum_is_not_left_to_the_light_table():
    """fuzz seed 1414 (round 6): 62 bytes, lines of 46, 5 and 1 letters, no final line break.  The light table derives 53
    bases from "every line but the last is as long as the first"; k_translate_wide, which would have found the line break
    among them, does not run on shards below 64 bytes -- and k_translate_frames4 validates nothing.  Such a shard takes the
    full index pass."""
    data = b">s0 a>b\ntgCCtCggaTtTCgAggcGGgcTAtTattAacAGTAccTggGcTGg\ngaGTt\nt"
    assert len(data) == 62
    for opts in ({"Config": {"LineWidth": 1}, "Frame": ["-3"], "TranslTable": 4, "AllowUnknownCodon": True, "Clean": True, "AppendFrame": True},
                 {"Frame": ["6"]}, {"Frame": ["1"], "Config": {"LineWidth": 0}}):
        want = oracle.translate(data, False, json.dumps(opts))
        for sets in ((), ((b"translate_index", b"light"),)):
            for got, _ in translate(data, opts, sets):
                assert got == want, (opts, sets)
    # ... and irregular lines in shards just above the minimum are caught by the wide kernel as before
    rng = random.Random(1414)
    for k in range(40):
        lines = ["".join(rng.choice("ACGTacgt") for _ in range(rng.choice((46, 30, 5, 1, 60)))) for _ in range(rng.randint(1, 4))]
        d = (">r x\n" + "\n".join(lines) + rng.choice(("", "\n"))).encode()
        want = oracle.translate(d, False, json.dumps({"Frame": ["6"]}))
        for got, _ in translate(d, {"Frame": ["6"]}):
            assert got == want, d
