"""The segmented copy (ops_segcopy.hip: FASTQ records that leave unchanged are written by an output-driven copy) against
the oracle and against the record-wise emit kernel, on inputs built to hit every branch: records shorter than a 16-byte
chunk, records longer than several 4 KiB tiles, long runs without output, '+' lines that repeat the name (those records
stay with k_seq_emit), a shard without final newline."""
import json
import random

import pytest

import oracle
import bigseqkit_amd as bsk

pytestmark = pytest.mark.gpu


def dev(data):
    import torch
    return torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()


class _Opts:
    def __init__(self, d):
        self.d = dict(d)
        self._v = self.d

    def to_json(self):
        return json.dumps(self.d)


def fastq_mix(rng, nrec, final_newline=True, plus_names=0.05, dups=False):
    out = []
    for i in range(nrec):
        k = rng.random()
        if k < 0.15:
            L = rng.randint(0, 3)                     # whole record below 16 bytes
        elif k < 0.20:
            L = rng.randint(5000, 12000)              # a record over several output tiles
        else:
            L = rng.randint(20, 300)
        seq = "".join(rng.choice("ACGT") for _ in range(L))
        qual = "".join(chr(rng.randint(33, 73)) for _ in range(L))
        if L and rng.random() < 0.3:
            qual = rng.choice("@+") + qual[1:]
        name = "r%d" % i if rng.random() < 0.7 else "read_%d some longer description %d" % (i, rng.randint(0, 10 ** 9))
        plus = "+" + (name if rng.random() < plus_names else "")
        out.append("@%s\n%s\n%s\n%s\n" % (name, seq, plus, qual))
    if dups:  # every third of a stretch once more, under another name
        for j, r in enumerate(out[10:200:3]):
            out.append("@dup%d\n" % j + r.split("\n", 1)[1])
        rng.shuffle(out)
    s = "".join(out)
    if not final_newline:
        s = s[:-1]
    return s.encode()


CASES = [
    ("seq", {}),                                   # every record leaves
    ("seq", {"MinLen": 4}),                        # the tiny ones are dropped
    ("seq", {"MinLen": 100}),                      # ~ 60 % dropped, in runs
    ("seq", {"MaxLen": 2}),                        # almost everything dropped
    ("grep", {"Pattern": ["ACG"], "BySeq": True, "InvertMatch": True}),
    ("rmdup", {"BySeq": True}),
]


def run(cmd, data, opts):
    frame = bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(data)])
    if cmd == "seq":
        return bsk.Seq(frame, _Opts(opts))
    if cmd == "grep":
        return bsk.Grep(frame, _Opts(opts))
    return bsk.RmDup(frame, _Opts(opts))


def want_of(cmd, data, opts):
    j = json.dumps(opts)
    return {"seq": oracle.seq, "grep": oracle.grep}[cmd](data, True, j) if cmd != "rmdup" else oracle.rmdup(data, True, j)


@pytest.mark.parametrize("seed", range(3))
@pytest.mark.parametrize("k", range(len(CASES)))
def test_segmented_copy_equals_oracle_and_record_emit(seed, k, monkeypatch):
    cmd, opts = CASES[k]
    rng = random.Random(9000 + 17 * seed + k)
    data = fastq_mix(rng, 600, final_newline=seed != 1, plus_names=0.0 if seed == 2 else 0.05, dups=cmd == "rmdup")
    want = want_of(cmd, data, opts)
    monkeypatch.setenv("BSK_SEGCOPY", "force")
    assert run(cmd, data, opts) == want
    monkeypatch.setenv("BSK_SEGCOPY", "off")
    assert run(cmd, data, opts) == want


def test_segmented_copy_many_tiny_records_per_tile(monkeypatch):
    # > 64 records begin inside one 4 KiB output tile: the kernel's lookup-from-memory branch
    recs = ["@%d\nA\n+\nI\n" % i for i in range(5000)] + ["@x%d\n%s\n+\n%s\n" % (i, "ACGT" * 40, "I" * 160) for i in range(300)]
    random.Random(5).shuffle(recs)
    data = "".join(recs).encode()
    want = oracle.seq(data, True, "{}")
    monkeypatch.setenv("BSK_SEGCOPY", "force")
    assert run("seq", data, {}) == want
    want2 = oracle.seq(data, True, '{"MinLen": 2}')   # long runs without output between the kept records
    assert run("seq", data, {"MinLen": 2}) == want2


def test_segmented_copy_default_threshold_on_c5_layout():
    """rmdup on the C5 layout (20 % duplicates) at a size where the default policy takes the segmented copy"""
    import ctypes as C
    import torch
    from bigseqkit_amd import _lib
    rb, nrec = 317, 400000
    t = torch.empty(rb * nrec, dtype=torch.uint8, device="cuda")
    assert _lib.lib.bsk_synth_device(0, 42, 2, 0, C.c_void_p(t.data_ptr()), rb * nrec, 0, None) == 0
    got = bsk.RmDup(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), _Opts({"BySeq": True}))
    data = bytes(t.cpu().numpy().tobytes())
    assert got == oracle.rmdup(data, True, '{"BySeq": true}')
