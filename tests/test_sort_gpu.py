"""Parity of `sort` against the CPU oracle, through the C ABI."""
import json
import random

import pytest

import oracle
import bigseqkit_amd as bsk

pytestmark = pytest.mark.gpu


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


def make(rng, nrec, fastq, width=60, final_newline=True):
    out = []
    for i in range(nrec):
        k = rng.randrange(nrec // 2 + 1)
        name = rng.choice(["id", "ID", "Read", "read_with_a_long_common_prefix_"]) + str(k)
        if rng.random() < 0.3:
            name += " " + rng.choice(["desc", "Desc B", "x"])
        L = rng.choice([0, 1, 7, 8, 9, 16, 17]) if rng.random() < 0.3 else rng.randint(0, 150)
        s = "".join(rng.choice("ACGTacgtN-.") for _ in range(L))
        if i % 9 == 0 and out:
            s = prev  # equal sequences / lengths: ties keep file order
        prev = s
        if fastq:
            q = "".join(chr(rng.randint(33, 73)) for _ in range(len(s)))
            out.append(f"@{name}\n{s}\n+\n{q}\n")
        else:
            w = width if width > 0 else max(1, len(s))
            out.append(f">{name}\n" + "".join(s[j:j + w] + "\n" for j in range(0, len(s), w)))
    text = "".join(out)
    if not final_newline:
        text = text[:-1]
    return text.encode()


OPTS = [{"InNaturalOrder": True}, {"InNaturalOrder": True, "IgnoreCase": True, "Reverse": True}, {"InNaturalOrder": True, "ByName": True},
        {}, {"Reverse": True}, {"IgnoreCase": True}, {"ByName": True}, {"ByName": True, "IgnoreCase": True, "Reverse": True},
        {"BySeq": True}, {"BySeq": True, "IgnoreCase": True}, {"BySeq": True, "SeqPrefixLength": 5},
        {"BySeq": True, "SeqPrefixLength": 0, "Reverse": True}, {"ByLength": True}, {"ByLength": True, "Reverse": True},
        {"ByBases": True}, {"ByBases": True, "GapLetters": "-N", "Reverse": True}, {"Config": {"LineWidth": 13}, "ByLength": True},
        {"Config": {"IDNCBI": True}}]


@pytest.mark.parametrize("fastq,width", [(True, 60), (False, 60), (False, 0), (False, 7)])
def test_sort_matches_oracle(fastq, width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(17 + fastq * 100 + width)
    data = make(rng, 600, fastq, width, final_newline=width != 0)
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    for o in OPTS:
        want = oracle.sort(data, fastq, json.dumps(o))
        got = bsk.Sort(bsk.SeqFrame(fmt, [dev(data)]), _Opts(o))
        assert got == want, (o, got[:300], want[:300])


def test_sort_hand_cases_and_errors():
    fa = b">b x\nACGT\n>A\nGG\n>c\nTTTTT\n>a\nC\n>B q\nA-\n"
    fr = lambda: bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(fa)])
    assert bsk.Sort(fr(), _Opts({})) == b">A\nGG\n>B q\nA-\n>a\nC\n>b x\nACGT\n>c\nTTTTT\n"
    assert bsk.Sort(fr(), _Opts({"ByLength": True, "Reverse": True})) == b">c\nTTTTT\n>b x\nACGT\n>A\nGG\n>B q\nA-\n>a\nC\n"
    assert bsk.Sort(fr(), _Opts({"ByBases": True})) == b">a\nC\n>B q\nA-\n>A\nGG\n>b x\nACGT\n>c\nTTTTT\n"
    assert bsk.Sort(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(b"")]), _Opts({})) == b""
    for o, msg in (({"BySeq": True, "ByName": True}, "only one of the options"), ({"ByBases": True, "BySeq": True}, "only one of the options"),
                   ):
        with pytest.raises(bsk.BskError) as e:
            bsk.Sort(fr(), _Opts(o))
        assert msg in str(e.value)
    # keys longer than one 8-byte chunk that differ only at the end, and a key that is a prefix of another
    ids = ["abcdefgh_2", "abcdefgh_10", "abcdefgh", "abcdefgh_1", "abcdefghi", "abcdefg"]
    data = "".join(f">{i}\nA\n" for i in ids).encode()
    got = bsk.Sort(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)]), _Opts({}))
    assert got == "".join(f">{i}\nA\n" for i in sorted(ids)).encode() == oracle.sort(data, False)


def test_sort_long_records(monkeypatch):
    monkeypatch.setenv("BSK_LONG_BYTES", "20000")
    rng = random.Random(4)
    recs = []
    for k, L in enumerate([90_000, 4, 70_000, 90_000, 0]):
        s = "".join(rng.choice("ACGT") for _ in range(L))
        recs.append(f">chr{5 - k} x\n" + "".join(s[j:j + 60] + "\n" for j in range(0, L, 60)))
    data = "".join(recs).encode()
    for o in ({}, {"ByLength": True, "Reverse": True}, {"BySeq": True, "SeqPrefixLength": 100}):
        assert bsk.Sort(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)]), _Opts(o)) == oracle.sort(data, False, json.dumps(o)), o


def test_sort_natural_order_hand_case():
    ids = ["chr10", "chr2", "chr1", "chrX", "chr1_random", "chr01", "Chr3", "chr2a", "chr", "10", "9", "a-1", "a1", "s007x12", "s7x3", "s7x"]
    data = "".join(f">{i} d\n{'ACGT' * (k + 1)}\n" for k, i in enumerate(ids)).encode()
    fr = lambda: bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)])
    got = bsk.Sort(fr(), _Opts({"InNaturalOrder": True}))
    assert got == oracle.sort(data, False, '{"InNaturalOrder": true}')
    order = [l.split()[0][1:] for l in got.decode().split("\n") if l.startswith(">")]
    assert order == ["9", "10", "Chr3", "a1", "a-1", "chr", "chr1", "chr01", "chr1_random", "chr2", "chr2a", "chr10", "chrX", "s7x", "s7x3", "s007x12"]
    for o in ({"InNaturalOrder": True, "Reverse": True}, {"InNaturalOrder": True, "IgnoreCase": True}, {"InNaturalOrder": True, "ByName": True}):
        assert bsk.Sort(fr(), _Opts(o)) == oracle.sort(data, False, json.dumps(o)), o
    # -N is ignored when sorting by sequence or length (sort.go:130-133)
    for o in ({"InNaturalOrder": True, "BySeq": True}, {"InNaturalOrder": True, "ByLength": True}):
        assert bsk.Sort(fr(), _Opts(o)) == oracle.sort(data, False, json.dumps(o)), o


@pytest.mark.parametrize("opts", [{"BySeq": True}, {"BySeq": True, "Reverse": True}, {"BySeq": True, "IgnoreCase": True},
                                  {"ByName": True}, {"ByName": True, "Reverse": True}])
def test_long_keys_leading_chunks_then_ties_only(opts, monkeypatch):
    """keys of more than three 8-byte chunks: the order by the two leading chunks, then only the tied positions by the rest
    (ops_host_next.cpp sort_run_device).  Records that share 16, 24, 40 leading key bytes, exact duplicates (ties keep
    file order), keys that end inside a chunk; the full LSD sweep (BSK_SORT=lsd) must give the same bytes."""
    rng = random.Random(5151)
    stems = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (16, 16, 24, 40, 41, 7)]
    recs = []
    for i in range(3000):
        k = rng.random()
        if k < 0.5:
            s = rng.choice(stems) + "".join(rng.choice("ACGTacgt") for _ in range(rng.randint(0, 60)))
        elif k < 0.6 and recs:
            s = rng.choice(recs)[1]                     # an exact duplicate of an earlier sequence
        else:
            s = "".join(rng.choice("ACGTN") for _ in range(rng.randint(0, 150)))
        name = "%s_%d tail %d" % (rng.choice(["a_long_common_name_prefix_of_more_than_16_bytes", "b" * 30, "c"]), rng.randrange(40), i % 7)
        recs.append((name, s))
    data = "".join("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)) for n, s in recs).encode()
    want = oracle.sort(data, True, json.dumps(opts))
    fr = lambda: bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(data)])
    assert bsk.Sort(fr(), _Opts(opts)) == want
    monkeypatch.setenv("BSK_SORT", "lsd")
    assert bsk.Sort(fr(), _Opts(opts)) == want
