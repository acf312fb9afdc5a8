"""The committed golden fixtures (tests/golden/fixtures.json, made by tests/golden/make_fixtures.py) against the
oracle as built here: a regression pin of the restatement, and a check that fixtures and oracle travel together."""
import json
import os

import pytest

import oracle

_G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIX = json.load(open(os.path.join(_G, "fixtures.json")))["cases"]
# expected values written BY HAND from the Go text (locate row formats, StatsString printf verbs): they pin the oracle,
# not the other way round ("source": "hand"; the derivation of every value is in the file)
HAND = json.load(open(os.path.join(_G, "hand_fixtures.json")))["cases"]
FN = {"seq": oracle.seq, "subseq": oracle.subseq, "translate": oracle.translate, "locate": oracle.locate, "grep": oracle.grep,
      "rmdup": oracle.rmdup, "fq2fa": oracle.fq2fa, "range": oracle.range_, "head": oracle.head, "duplicate": oracle.duplicate,
      "rename": oracle.rename, "sort": oracle.sort, "faidx": oracle.faidx, "faidx_query": oracle.faidx_query,
      "stats": lambda d, f, o: oracle.stats_string(d, f, o, name="input0").encode()}


@pytest.mark.parametrize("case", FIX, ids=[c["name"] for c in FIX])
def test_oracle_reproduces_the_golden_fixture(case):
    data = case["input"].encode("latin1")
    if "error" in case:
        with pytest.raises(oracle.OracleError) as e:
            FN[case["op"]](data, case["fastq"], json.dumps(case["opts"]))
        assert str(e.value) == case["error"]
    else:
        assert FN[case["op"]](data, case["fastq"], json.dumps(case["opts"])) == case["expected"].encode("latin1")


@pytest.mark.parametrize("case", HAND, ids=[c["name"] for c in HAND])
def test_oracle_reproduces_the_hand_derived_fixture(case):
    assert case["source"] == "hand" and case["derivation"]
    data = case["input"].encode("latin1")
    assert FN[case["op"]](data, case["fastq"], json.dumps(case["opts"])) == case["expected"].encode("latin1")


def test_fixture_set_covers_every_operator():
    assert {c["op"] for c in FIX} == set(FN)
    assert len(FIX) >= 84
