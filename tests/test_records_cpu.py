"""fq2fa / range / head / duplicate without a GPU: the oracle against hand-written expectations (the reference has no
tests for them; the arithmetic follows bigseqkit/range.go:43-86 line by line, PARITY.md RNG) and the host-side option
handling of libbsk (bsk_create on device -1)."""
import json

import pytest

import oracle
import bigseqkit_amd as bsk

FA = b">a x\nACGT\nAC\n>b\nGG\n\n>c\nT"
FQ = b"@r1 d\nACGT\n+\nIIII\n@r2\nGG\n+r2\n##\n@r3\nT\n+\n@\n"
RECS = [b">%d\nA\n" % i for i in range(1, 21)]
MANY = b"".join(RECS)


def pick(lo, hi):  # 1-based inclusive
    return b"".join(RECS[lo - 1:hi])


def test_fq2fa_drops_the_quality_and_unwraps():
    assert oracle.fq2fa(FQ, True) == b">r1 d\nACGT\n>r2\nGG\n>r3\nT\n"
    assert oracle.fq2fa(FA, False) == b">a x\nACGTAC\n>b\nGG\n>c\nT\n"
    assert oracle.fq2fa(FA, False, json.dumps({"Config": {"LineWidth": 2}})) == b">a x\nACGTAC\n>b\nGG\n>c\nT\n"  # Format(0)
    with pytest.raises(oracle.OracleError):
        oracle.fq2fa(FA, False, json.dumps({"Config": {"SeqType": "bogus"}}))


@pytest.mark.parametrize("rng,lo,hi", [("1:12", 1, 12), ("5:5", 5, 5), ("3", 3, 20), ("19:40", 19, 20),
                                       ("-12:-1", 9, 20), ("-1:-1", 1, 20), ("-1:5", 1, 5), ("-10:-3", 11, 17),
                                       ("2:-2", 2, 18), ("-3", 18, 20)])
def test_range_arithmetic_as_written(rng, lo, hi):
    want = pick(lo, hi)
    assert oracle.range_(MANY, False, json.dumps({"Range": rng})) == want
    assert oracle.range_(MANY, False, json.dumps({"Range": rng}), nparts=3) == want  # the index is global


@pytest.mark.parametrize("rng,msg", [("", "flag -r (--range) needed"), ("0:3", "either start and end should not be 0"),
                                     ("3:0", "either start and end should not be 0"), ("9:3", "start must be > than end"),
                                     ("17:", 'strconv.ParseInt: parsing "": invalid syntax'),
                                     ("x:2", 'strconv.ParseInt: parsing "x": invalid syntax'),
                                     ("-3:-8", "start must be > than end")])
def test_range_errors(rng, msg):
    with pytest.raises(oracle.OracleError) as e:
        oracle.range_(MANY, False, json.dumps({"Range": rng}))
    assert msg in str(e.value)
    if rng != "-3:-8":  # needs the record count: reported by bsk_range_set_count, not by bsk_create
        with pytest.raises(bsk.BskError) as e2:
            bsk.Operator("Range", json.dumps({"Range": rng}), -1)
        assert msg in str(e2.value)


def test_head_is_range_1_to_n_and_duplicate_keeps_copies_adjacent():
    assert oracle.head(MANY, False) == pick(1, 10)
    assert oracle.head(MANY, False, '{"N": 3}') == pick(1, 3)
    assert oracle.head(FQ, True, '{"N": 2}') == b"@r1 d\nACGT\n+\nIIII\n@r2\nGG\n+r2\n##\n"
    assert oracle.duplicate(FA, False, '{"Times": 2}') == b">a x\nACGT\nAC\n>a x\nACGT\nAC\n>b\nGG\n\n>b\nGG\n\n>c\nT\n>c\nT\n"
    assert oracle.duplicate(FQ, True) == FQ
    assert oracle.duplicate(FQ, True, '{"Times": 0}') == b""
    with pytest.raises(oracle.OracleError):
        oracle.duplicate(FQ, True, '{"Times": -2}')


def test_libbsk_option_handling_of_the_record_operators():
    with bsk.Operator("Head", "{}", -1) as op:
        assert json.loads(op.opts_json())["N"] == 10
    with bsk.Operator("Duplicate", '{"Times": null}', -1) as op:
        assert json.loads(op.opts_json())["Times"] == 1
    with bsk.Operator("Range", '{"Range": "-12:-1"}', -1) as op:
        import ctypes as C
        needs = C.c_int()
        assert bsk.lib.bsk_range_needs_count(op.ctx, C.byref(needs)) == 0 and needs.value == 1
        assert bsk.lib.bsk_range_set_count(op.ctx, 5) == 0   # start -7: nothing before record 0 exists, all 5 are kept
        assert bsk.lib.bsk_range_needs_count(op.ctx, C.byref(needs)) == 0 and needs.value == 0
    with bsk.Operator("Range", '{"Range": "-3:-8"}', -1) as op:
        assert bsk.lib.bsk_range_set_count(op.ctx, 20) != 0  # start 17 >= end 12
        assert "start must be > than end" in bsk.lib.bsk_last_error(op.ctx).decode()
    with pytest.raises(bsk.BskError):
        bsk.Operator("Duplicate", '{"Times": -1}', -1)
    with pytest.raises(bsk.BskError):
        bsk.Operator("Fq2Fa", '{"Config": {"SeqType": "bogus"}}', -1)


def test_rename_numbers_further_records_of_an_id_in_file_order():
    fa = b">a x y\nACGT\n>b\nGG\n>a\tz\nTT\n>a\nC\n>b q\nA\n"
    assert oracle.rename(fa, False) == b">a x y\nACGT\n>b\nGG\n>a_1 z\nTT\n>a_2 \nC\n>b_1 q\nA\n"
    assert oracle.rename(fa, False, '{"ByName": true}') == fa   # all five headers differ
    fa2 = b">a x\nAC\n>a x\nGT\n>a y\nTT\n"
    assert oracle.rename(fa2, False, '{"ByName": true}') == b">a x\nAC\n>a_1 x\nGT\n>a y\nTT\n"
    fq = b"@r1 d\nAC\n+\nII\n@r1\nG\n+\n#\n"
    assert oracle.rename(fq, True) == b"@r1 d\nAC\n+\nII\n@r1_1 \nG\n+\n#\n"
    wrapped = b">s\nACGTACGT\n>s\nAAAA\n"
    assert oracle.rename(wrapped, False, '{"Config": {"LineWidth": 3}}') == b">s\nACG\nTAC\nGT\n>s_1 \nAAA\nA\n"
    with bsk.Operator("Rename", "{}", -1) as op:
        assert json.loads(op.opts_json())["ByName"] is False


def test_sort_keys_ties_and_errors():
    fa = b">b x\nACGT\n>A\nGG\n>c\nTTTTT\n>a\nC\n>B q\nA-\n"
    assert oracle.sort(fa, False) == b">A\nGG\n>B q\nA-\n>a\nC\n>b x\nACGT\n>c\nTTTTT\n"
    assert oracle.sort(fa, False, '{"IgnoreCase": true}') == b">A\nGG\n>a\nC\n>b x\nACGT\n>B q\nA-\n>c\nTTTTT\n"  # ties: file order
    assert oracle.sort(fa, False, '{"ByLength": true, "Reverse": true}') == b">c\nTTTTT\n>b x\nACGT\n>A\nGG\n>B q\nA-\n>a\nC\n"
    assert oracle.sort(fa, False, '{"ByBases": true}') == b">a\nC\n>B q\nA-\n>A\nGG\n>b x\nACGT\n>c\nTTTTT\n"
    assert oracle.sort(fa, False, '{"BySeq": true}') == b">B q\nA-\n>b x\nACGT\n>a\nC\n>A\nGG\n>c\nTTTTT\n"
    with pytest.raises(oracle.OracleError):
        oracle.sort(fa, False, '{"BySeq": true, "ByLength": true}')
    for o, msg in (('{"BySeq": true, "ByName": true}', "only one of the options"),):
        with pytest.raises(bsk.BskError) as e:
            bsk.Operator("Sort", o, -1)
        assert msg in str(e.value)
    with bsk.Operator("Sort", "{}", -1) as op:
        d = json.loads(op.opts_json())
        assert d["GapLetters"] == "- \t." and d["SeqPrefixLength"] == 10000 and d["Reverse"] is False


def test_faidx_rows_are_the_fai_columns_with_true_offsets():
    fa = b">a x\nACGTACGT\nACGTACGT\nACG\n>b\nGG\n>c\n>d q\nTTTT\nTT"
    assert oracle.faidx(fa, False) == b"a\t19\t5\t8\t9\nb\t2\t30\t2\t3\nc\t0\t36\t0\t0\nd\t6\t41\t4\t5\n"
    assert oracle.faidx(fa, False, nparts=3) == oracle.faidx(fa, False)   # partition offsets accumulate
    assert oracle.faidx(fa, False, '{"FullHead": true}').split(b"\n")[3] == b"d q\t6\t41\t4\t5"
    fq = b"@r1 d\nACGT\n+\nIIII\n@r2\nGG\n+r2\n##\n"
    assert oracle.faidx(fq, True) == b"r1\t4\t6\t4\t5\t13\nr2\t2\t22\t2\t3\t29\n"
    with pytest.raises(oracle.OracleError) as e:
        oracle.faidx(b">a\nACGT\nACGTAC\nAC\n", False)
    assert "different line length in sequence: a." in str(e.value)
    with pytest.raises(bsk.BskError):
        bsk.Operator("Faidx", '{"Regions": ["chr1("], "UseRegexp": true}', -1)
    fq = b">chr1 x\nACGTACGTAC\nGGGGGTTTTT\n>chr2\nAAAACCCC\n"
    assert oracle.faidx_query(fq, False, '{"Regions": ["chr1:2-5", "chr2:-3", "chr2:1-2", "chr1:5-2"]}') == b">chr1:2-5\nCGTA\n>chr2:1-3\nAAA\n"
    assert oracle.faidx_query(fq, False, '{"Regions": ["chr1:5-2"]}') == b">chr1:5-2\nTACG\n"
    assert oracle.faidx_query(fq, False, '{"Regions": ["chr1:15-"]}') == b">chr1\nGTTTTT\n"


def test_pair_kth_with_kth_and_the_rest_unpaired():
    a = b"@r1 1\nAC\n+\nII\n@r2 1\nGG\n+\nII\n@r1 1b\nTT\n+\nII\n@r5\nA\n+\nI\n"
    b = b"@r2 2\nCC\n+\nII\n@r9\nT\n+\nI\n@r1 2\nGT\n+\nII\n"
    p1, p2, u1, u2 = oracle.pair(a, b, True)
    assert p1 == b"@r1 1\nAC\n+\nII\n@r2 1\nGG\n+\nII\n" and p2 == b"@r1 2\nGT\n+\nII\n@r2 2\nCC\n+\nII\n"
    assert u1 == b"@r1 1b\nTT\n+\nII\n@r5\nA\n+\nI\n" and u2 == b"@r9\nT\n+\nI\n"
    with bsk.Operator("Pair", "{}", -1) as op:
        assert json.loads(op.opts_json())["SaveUnpaired"] is False


def test_common_first_file_records_present_in_every_file():
    a = b">x 1\nACGT\n>y\nGG\n>x 2\nTT\n>z\nAC\n"
    b = b">z q\nAA\n>x\nC\n"
    c = b">X\nA\n>z\nT\n"
    assert oracle.common([a, b], False) == b">x 1\nACGT\n>z\nAC\n"          # one record per key: the first x
    assert oracle.common([a, b, c], False) == b">z\nAC\n"
    assert oracle.common([a, b, c], False, '{"IgnoreCase": true}') == b">x 1\nACGT\n>z\nAC\n"
    assert oracle.common([a, b"" + b], False, '{"BySeq": true}') == b""
    assert oracle.common([a, b">q\nacgt\n"], False, '{"BySeq": true, "IgnoreCase": true}') == b">x 1\nACGT\n"
    with pytest.raises(bsk.BskError):
        bsk.Operator("Common", '{"BySeq": true, "ByName": true}', -1)


def test_concat_joins_every_a_with_every_b_of_an_id():
    a = b">x 1\nACGT\n>y\nGG\n>x 2\nTT\n"
    b = b">z q\nAA\n>x d\nCCC\n>x e\nG\n"
    assert oracle.concat(a, b, False) == b">x\nACGTCCC\n>x\nACGTG\n>x\nTTCCC\n>x\nTTG\n"
    assert oracle.concat(a, b, False, '{"Full": true, "Config": {"LineWidth": 3}}') == \
        b">x\nACG\nTCC\nC\n>x\nACG\nTG\n>y\nGG\n>x\nTTC\nCC\n>x\nTTG\n>z q\nAA\n"
    assert oracle.concat(b"@r1 a\nAC\n+\nII\n", b"@r1 b\nGGG\n+\n###\n", True) == b"@r1\nACGGG\n+\nII###\n"
    with bsk.Operator("Concat", "{}", -1) as op:
        assert json.loads(op.opts_json())["Separator"] == "|"
