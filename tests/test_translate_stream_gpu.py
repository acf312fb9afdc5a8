"""`translate` on FASTA records that do not all look alike, in ONE pass over the input (k_translate_stream, round 5): the
blocks find the record starts of small line-start ranges, derive lengths by the light table's rule, learn where their output
begins from a chain over the ranges, and translate -- every window verified, anything that does not fit sends the call
through the table paths.  Either way the output is the oracle's (Translate.Call,
/root/reference/bigseqkit-lib/translate.go:104-145)."""
import ctypes as C
import json
import random
import zlib

import pytest

import oracle
import bigseqkit_amd as bsk
from bigseqkit_amd._lib import lib
from test_translate_light_gpu import fasta, translate

pytestmark = pytest.mark.gpu
FORCE = ((b"translate_stream", b"force"),)
OPTS = [{"Frame": ["6"]}, {"Frame": ["-2", "3"], "Config": {"LineWidth": 0}}, {"Frame": ["6"], "TranslTable": 11, "Clean": True},
        {"Frame": ["1"], "AllowUnknownCodon": True, "Config": {"LineWidth": 70}}]


@pytest.mark.parametrize("min_range", ["4096", "65536"])
@pytest.mark.parametrize("o", range(len(OPTS)))
@pytest.mark.parametrize("shape", ["w60", "w70", "one_line", "own_width", "no_final_newline", "lower", "cds5k", "empty_records"])
def test_one_pass_translation(shape, o, min_range, monkeypatch):
    if shape == "empty_records" and min_range != "4096":
        pytest.skip("records of ~60 bytes: more than 256 per 64 KiB range -- the list overflows and the tables take over (tested below)")
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", min_range)
    rng = random.Random(zlib.crc32(shape.encode()) % 1000 + o)  # (str hashes differ from process to process)
    if shape == "w60": data = fasta(rng, 400, 60)
    elif shape == "w70": data = fasta(rng, 300, 70, lens=(0, 500))
    elif shape == "one_line": data = fasta(rng, 300, 0, lens=(1, 700))
    elif shape == "own_width": data = fasta(rng, 300, lambda i: (50, 60, 80, 0, 101)[i % 5])
    elif shape == "no_final_newline": data = fasta(rng, 200, 60, newline_at_end=False)
    elif shape == "lower": data = fasta(rng, 300, 60, alphabet="ACGTacgt")
    elif shape == "cds5k": data = fasta(rng, 60, 60, lens=(4000, 12000))      # records that span several ranges of 4 KiB
    else: data = b">first\nACGTTGCA\n" + fasta(rng, 500, 60, lens=(0, 40))       # many records without a sequence (not the first: below)
    want = oracle.translate(data, False, json.dumps(OPTS[o]))
    for got, stages in translate(data, OPTS[o], FORCE):
        assert got == want
        assert "k_translate_stream" in stages and "k_fasta_starts" not in stages and "k_index=" not in stages, stages


@pytest.mark.parametrize("sets", [FORCE, ((b"translate_stream", b"off"),)])
def test_a_first_record_without_a_sequence_is_not_dna(sets):
    """the alphabet is guessed from the FIRST record (translate.go:116-122 on SeqParser's guess): one without a sequence is
    not DNA, whatever follows -- the reference's error on every path, not a translation"""
    rng = random.Random(11)
    data = b">r0 nothing here\n" + fasta(rng, 300, 60)
    with pytest.raises(oracle.OracleError, match="only apply to DNA/RNA"):
        oracle.translate(data, False, json.dumps(OPTS[0]))
    with pytest.raises(bsk.BskError, match="only apply to DNA/RNA"):
        translate(data, OPTS[0], sets)


@pytest.mark.parametrize("opts", [{"Frame": ["1"], "Trim": True}, {"Frame": ["6"], "AppendFrame": True}, {"Frame": ["6"], "InitCodonAsM": True}])
def test_options_that_need_the_table_do_not_take_the_pass(opts, monkeypatch):
    rng = random.Random(3)
    data = fasta(rng, 200, 60)
    want = oracle.translate(data, False, json.dumps(opts))
    for got, stages in translate(data, opts, FORCE):
        assert got == want and "k_translate_stream" not in stages


@pytest.mark.parametrize("why", ["N_late", "short_line_late", "blank_line", "long_record", "many_records_in_a_range", "off"])
def test_text_that_does_not_fit_goes_through_the_tables(why, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "65536" if why == "many_records_in_a_range" else "4096")
    monkeypatch.setenv("BSK_LONG_BYTES", "20000")
    rng = random.Random(len(why))
    data = fasta(rng, 900, 60)
    lines = data.split(b"\n")
    sets = FORCE
    if why == "N_late":
        k = len(lines) - 30
        while lines[k].startswith(b">") or len(lines[k]) < 10: k -= 1
        lines[k] = lines[k][:5] + b"N" + lines[k][6:]
    elif why == "short_line_late":
        k = len(lines) - 40
        while lines[k].startswith(b">") or lines[k + 1].startswith(b">") or len(lines[k]) != 60: k -= 1
        lines[k] = lines[k][:31]
    elif why == "blank_line":
        k = len(lines) - 50
        while not lines[k].startswith(b">"): k -= 1
        lines.insert(k, b"")
    elif why == "long_record":
        lines.append(b">big"); lines.append(b"ACGT" * 10000)
    elif why == "many_records_in_a_range":
        data = fasta(rng, 3000, 60, lens=(20, 60)); lines = data.split(b"\n")   # ~ 700 records per 64 KiB: more than the list holds
    else:
        sets = ((b"translate_stream", b"off"),)
    data = b"\n".join(lines)
    opts = {"Frame": ["6"]}
    want = oracle.translate(data, False, json.dumps(opts))
    outs = translate(data, opts, sets)
    assert outs[0][0] == want and outs[1][0] == want
    assert ("k_translate_stream" in outs[0][1]) == (why != "off")
    assert "k_translate_stream" not in outs[1][1]        # the context remembered


def test_records_that_differ_at_the_size_of_the_bench_layout():
    """the synthetic layout of bench.py's second C4 leg (unpadded numbers in the header, 1 % of the records 3 bases shorter /
    longer): the default takes the one-pass translation (records of 5 kb) and equals the table path byte for byte"""
    import torch
    from bigseqkit_amd import _lib
    nrec = 20000
    n = lib.bsk_synth_offset(_lib.SYNTH_FASTA5K_VAR, nrec)
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.bsk_synth_device(_lib.SYNTH_FASTA5K_VAR, 42, 0, 0, C.c_void_p(t.data_ptr()), n, 0, None) == 0
    torch.cuda.synchronize()
    data = bytes(t.cpu().numpy().tobytes())
    outs = translate(data, {"Frame": ["6"]})
    assert "k_translate_stream" in outs[0][1] and "k_translate_stream" in outs[1][1]
    ref = translate(data, {"Frame": ["6"]}, ((b"translate_stream", b"off"),))
    assert "k_translate_stream" not in ref[0][1]
    assert outs[0][0] == ref[0][0] == outs[1][0]
    assert outs[0][0][:40] == oracle.translate(data[:lib.bsk_synth_offset(_lib.SYNTH_FASTA5K_VAR, 3)], False, '{"Frame": ["6"]}')[:40]
