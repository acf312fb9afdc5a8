"""The HIP path against the committed golden fixtures (tests/golden/fixtures.json): inputs and expected outputs only --
nothing here calls the oracle."""
import json
import os

import pytest

import bigseqkit_amd as bsk

pytestmark = pytest.mark.gpu
_G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIX = json.load(open(os.path.join(_G, "fixtures.json")))["cases"]
FIX = FIX + json.load(open(os.path.join(_G, "hand_fixtures.json")))["cases"]  # hand-derived from the Go text ("source": "hand")


def dev(data):
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8) if len(data) else torch.empty(0, dtype=torch.uint8)
    return t.cuda()


class _Opts:
    def __init__(self, d):
        self.d = dict(d)
        self._v = self.d

    def to_json(self):
        return json.dumps(self.d)


def _stats(fr, o):
    return bsk.StatsString("input0", "N/A", fr, o).encode()


FN = {"seq": bsk.Seq, "subseq": bsk.Subseq, "translate": bsk.Translate, "locate": bsk.Locate, "grep": bsk.Grep, "rmdup": bsk.RmDup,
      "fq2fa": bsk.Fq2Fa, "range": bsk.Range, "head": bsk.Head, "duplicate": bsk.Duplicate, "rename": bsk.Rename, "sort": bsk.Sort,
      "faidx": bsk.Faidx, "faidx_query": bsk.FaidxQuery, "stats": _stats}


@pytest.mark.parametrize("case", FIX, ids=[c["name"] for c in FIX])
def test_hip_path_reproduces_the_golden_fixture(case):
    data = case["input"].encode("latin1")
    fr = bsk.SeqFrame(bsk.FORMAT_FASTQ if case["fastq"] else bsk.FORMAT_FASTA, [dev(data)])
    if "error" in case:
        with pytest.raises(bsk.BskError) as e:
            FN[case["op"]](fr, _Opts(case["opts"]))
        assert case["error"] in str(e.value)
    else:
        assert FN[case["op"]](fr, _Opts(case["opts"])) == case["expected"].encode("latin1")
