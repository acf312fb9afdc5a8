"""Parity of `translate` and `rmdup` against the CPU oracle, through the C ABI."""
import ctypes as C
import json
import random

import pytest

import oracle
import seqgen
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib

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


def frame(data, fastq):
    return bsk.SeqFrame(bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA, [dev(data)])


def check_translate(data, fastq, opts):
    want = oracle.translate(data, fastq, json.dumps(opts))
    got = bsk.Translate(frame(data, fastq), _Opts(opts))
    assert got == want, (opts, got[:200], want[:200])


TR_OPTS = [
    {},
    {"Frame": ["6"]},
    {"Frame": ["6"], "AppendFrame": True},
    {"Frame": ["2", "-3"], "Config": {"LineWidth": 0}},
    {"Frame": ["6"], "Trim": True},
    {"Frame": ["1", "-1"], "Clean": True, "Config": {"LineWidth": 17}},
    {"Frame": ["6"], "InitCodonAsM": True, "TranslTable": 11},
    {"Frame": ["6"], "TranslTable": 2, "Trim": True, "Clean": True},
    {"Frame": ["3"], "TranslTable": 4, "AppendFrame": True},
]


@pytest.mark.parametrize("alphabet", ["ACGT", "ACGTNRYacgt", "ACGUNacgu"])
@pytest.mark.parametrize("width", [60, 0, 11])
@pytest.mark.parametrize("i", range(len(TR_OPTS)))
def test_translate_fasta(i, width, alphabet, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(700 + i)
    data = seqgen.random_fasta(rng, 150, 0, 700, width=width, alphabet=alphabet, final_newline=i % 2 == 0)
    data = data.replace(b"-", b"A").replace(b".", b"C").replace(b" ", b"G")  # no gap letters here
    check_translate(data, False, TR_OPTS[i])


def test_translate_fastq_and_irregular_fasta():
    rng = random.Random(9)
    fq = seqgen.random_fastq(rng, 300, 0, 160, alphabet="ACGT")
    fq = fq  # gap letters may appear: handled below with -x
    check_translate(fq, True, {"Frame": ["6"], "AllowUnknownCodon": True})
    recs = []
    for k in range(80):
        L = rng.randint(0, 500)
        s = "".join(rng.choice("ACGTN") for _ in range(L))
        lines, j = [], 0
        while j < L:
            w = rng.randint(1, 40)
            lines.append(s[j:j + w])
            j += w
        recs.append(f">s{k} d\n" + "".join(l + "\n" for l in lines))
    check_translate("".join(recs).encode(), False, {"Frame": ["6"], "AppendFrame": True})


def test_translate_unknown_codon_and_errors():
    bad = b">a\nATG-CCTAA\n"
    with pytest.raises(bsk.BskError, match="unknown codon"):
        bsk.Translate(frame(bad, False), _Opts({}))
    with pytest.raises(oracle.OracleError, match="unknown codon"):
        oracle.translate(bad, False, "{}")
    check_translate(bad, False, {"AllowUnknownCodon": True})
    prot = b">p\nMKVLAAGIVGLLLAQ\n"
    with pytest.raises(bsk.BskError, match="only apply to DNA/RNA"):
        bsk.Translate(frame(prot, False), _Opts({}))
    with pytest.raises(oracle.OracleError, match="only apply to DNA/RNA"):
        oracle.translate(prot, False, "{}")
    for opts, msg in [({"TranslTable": 7}, "invalid translate table: 7"), ({"Frame": ["4"]}, "invalid frame: 4"),
                      ({"Frame": ["x"]}, "invalid frame(s): x")]:
        with pytest.raises(bsk.BskError) as e:
            bsk.Operator("Translate", json.dumps(opts), -1)
        assert msg in str(e.value)
        with pytest.raises(oracle.OracleError) as oe:
            oracle.translate(b">a\nATG\n", False, json.dumps(opts))
        assert msg in str(oe.value)


def test_translate_c4_synthetic_cds():
    """BASELINE C4 layout: every frame-1 protein is M + 1665 residues + '*'."""
    import torch
    rb, nrec = 5107, 20000
    t = torch.empty(rb * nrec, dtype=torch.uint8, device="cuda")
    assert _lib.lib.bsk_synth_device(2, 42, 0, 0, C.c_void_p(t.data_ptr()), rb * nrec, 0, None) == 0
    got = bsk.Translate(bsk.SeqFrame(bsk.FORMAT_FASTA, [t]), _Opts({"Frame": ["1"], "Config": {"LineWidth": 0}}))
    recs = got.split(b"\n")[:-1]
    assert len(recs) == 2 * nrec
    assert all(len(p) == 1667 and p[:1] == b"M" and p[-1:] == b"*" and b"*" not in p[:-1] for p in recs[1::2])
    head = bytes(t[:rb * 300].cpu().numpy().tobytes())
    o = {"Frame": ["6"]}
    assert bsk.Translate(frame(head, False), _Opts(o)) == oracle.translate(head, False, json.dumps(o))


def check_rmdup(data, fastq, opts):
    want = oracle.rmdup(data, fastq, json.dumps(opts))
    got = bsk.RmDup(frame(data, fastq), _Opts(opts))
    assert got == want, (opts, len(got), len(want))
    return got


def dup_fastq(rng, n, L=100):
    seqs, out = [], []
    for i in range(n):
        if i > 5 and rng.random() < 0.3:
            s = seqs[rng.randrange(len(seqs))]
            if rng.random() < 0.3:
                s = s.lower()
        else:
            s = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, L)))
        seqs.append(s)
        name = f"r{i % (n // 2) if rng.random() < 0.2 else i} c{rng.randint(0, 3)}"
        out.append(f"@{name}\n{s}\n+\n{'I' * len(s)}\n")
    return "".join(out).encode()


RMDUP_OPTS = [{"BySeq": True}, {"BySeq": True, "IgnoreCase": True}, {"BySeq": True, "OnlyPositiveStrand": True},
              {}, {"ByName": True}, {"ByName": True, "IgnoreCase": True}]


@pytest.mark.parametrize("i", range(len(RMDUP_OPTS)))
def test_rmdup_fastq(i, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(800 + i)
    data = dup_fastq(rng, 1500)
    got = check_rmdup(data, True, RMDUP_OPTS[i])
    assert 0 < got.count(b"\n") // 4 < 1500


@pytest.mark.parametrize("width", [60, 0])
def test_rmdup_fasta(width, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(31)
    seqs, recs = [], []
    for k in range(400):
        s = seqs[rng.randrange(len(seqs))] if (k > 3 and rng.random() < 0.4) else \
            "".join(rng.choice("ACGTacgt") for _ in range(rng.randint(0, 400)))
        seqs.append(s)
        w = width if width else max(1, len(s))
        if k % 5 == 0 and len(s) > 3:  # same sequence, different wrapping
            w = rng.randint(1, 30)
        recs.append(f">s{k}\n" + "".join(s[j:j + w] + "\n" for j in range(0, len(s), w)))
    data = "".join(recs).encode()
    for o in ({"BySeq": True}, {"BySeq": True, "IgnoreCase": True}):
        check_rmdup(data, False, o)


def test_rmdup_hash_matches_golden_xxh64():
    """The survivors of `-s` are decided by XXH64 keys: feed the golden vectors as sequences."""
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "xxh64_vectors.json")))
    seqs = [bytes.fromhex(v["hex"]) for v in g["vectors"]]
    fa = b"".join(b">v%d\n%s\n" % (i, s) for i, s in enumerate(seqs + seqs[::-1]))
    check_rmdup(fa, False, {"BySeq": True, "Config": {"LineWidth": 0}})


def test_rmdup_option_errors():
    for opts, msg in [({"BySeq": True, "ByName": True}, "only one/none of the flags"),
                      ({"OnlyPositiveStrand": True}, "flag -s (--by-seq) needed")]:
        with pytest.raises(bsk.BskError) as e:
            bsk.Operator("RmDup", json.dumps(opts), -1)
        assert msg in str(e.value)
        with pytest.raises(oracle.OracleError) as oe:
            oracle.rmdup(b">a\nA\n", False, json.dumps(opts))
        assert msg in str(oe.value)


def test_rmdup_c5_synthetic_duplicates():
    """BASELINE C5 rule: record i with i%5==4 copies an earlier sequence -> 20 % removed."""
    import torch
    rb, nrec = 317, 2_000_000
    t = torch.empty(rb * nrec, dtype=torch.uint8, device="cuda")
    assert _lib.lib.bsk_synth_device(0, 42, _lib.SYNTH_FLAG_DUPS, 0, C.c_void_p(t.data_ptr()), rb * nrec, 0, None) == 0
    got = bsk.RmDup(bsk.SeqFrame(bsk.FORMAT_FASTQ, [t]), _Opts({"BySeq": True}))
    kept = len(got) // rb
    assert len(got) == kept * rb and kept == nrec - nrec // 5
    # idempotence: a second pass removes nothing
    again = bsk.RmDup(frame(got[:rb * 50000], True), _Opts({"BySeq": True}))
    assert again == got[:rb * 50000]
    head = bytes(t[:rb * 30000].cpu().numpy().tobytes())
    assert bsk.RmDup(frame(head, True), _Opts({"BySeq": True})) == oracle.rmdup(head, True, '{"BySeq": true}')


@pytest.mark.parametrize("i", range(len(RMDUP_OPTS)))
def test_rmdup_side_files(tmp_path, i, monkeypatch):
    """-d / -D: text of the removed records and the duplicate-number lines (rmdup.go:179-186, 224-279)."""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(1800 + i)
    data = dup_fastq(rng, 1200)
    d, D = tmp_path / "dups" / "seqs", tmp_path / "nums"
    opts = dict(RMDUP_OPTS[i], DupSeqsFile=str(d), DupNumFile=str(D))
    check_rmdup(data, True, opts)
    assert (d / "0").read_bytes() == oracle.rmdup_side(data, True, json.dumps(opts), 1)
    nums = (D / "0").read_bytes()
    assert nums == oracle.rmdup_side(data, True, json.dumps(opts), 2)
    assert nums.count(b"\n") > 10 and all(int(l.split(b"\t")[0]) == l.count(b", ") + 1 for l in nums.splitlines())


def test_rmdup_side_files_fasta_and_nothing_removed(tmp_path):
    rng = random.Random(4)
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 300))) for _ in range(60)]
    fa = "".join(f">s{k} d\n" + "".join(s[j:j + 50] + "\n" for j in range(0, len(s), 50))
                 for k, s in enumerate(seqs + seqs[:25] + seqs[:5])).encode()
    d = tmp_path / "d"
    opts = {"BySeq": True, "DupSeqsFile": str(d), "Config": {"LineWidth": 70}}
    check_rmdup(fa, False, opts)
    assert (d / "0").read_bytes() == oracle.rmdup_side(fa, False, json.dumps(opts), 1)
    uniq = "".join(f">u{k}\n{s}\n" for k, s in enumerate(seqs)).encode()
    e = tmp_path / "e"
    check_rmdup(uniq, False, {"BySeq": True, "DupSeqsFile": str(e), "DupNumFile": str(e)})
    assert not e.exists()        # After() writes nothing when no record was removed (rmdup.go:245)


# ---------------------------------------------------------------- rmdup across ranks (device phases)
def _virtual_ranks(data, fastq, opts, world):
    """Drive `world` HipRmDupBackend contexts in ONE process on one GPU, doing by hand the exchanges that
    dist.rmdup_distributed does with all_gather / all_to_all_single (the collectives themselves are covered by the
    world_size-2 gloo test).  Returns the concatenated survivors."""
    import torch
    from bigseqkit_amd import dist as bdist
    fmt = bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA
    bounds = bdist.shard_bounds(data, world, fmt)
    shards = [dev(data[lo:hi]) for lo, hi in bounds]
    backs = [bdist.HipRmDupBackend(json.dumps(opts), 0) for _ in range(world)]
    try:
        ns = [b.keys(s, fmt) for b, s in zip(backs, shards)]
        bases = [sum(ns[:r]) for r in range(world)]
        packed = [b.pack(bases[r], world) for r, b in enumerate(backs)]
        assert all(sum(c) == n for (_, c), n in zip(packed, ns))
        # tuples for owner o: the o-th bucket of every sender, in sender order
        def bucket(r, o):
            send, counts = packed[r]
            a = sum(counts[:o])
            return send[a:a + counts[o]]
        keeps = []
        for o in range(world):
            recv = torch.cat([bucket(r, o) for r in range(world)]) if world > 1 else packed[0][0]
            keeps.append(backs[o].resolve(recv.contiguous()))
        outs = []
        for r in range(world):
            parts = []
            for o in range(world):
                a = sum(packed[s][1][o] for s in range(r))
                parts.append(keeps[o][a:a + packed[r][1][o]])
            reply = torch.cat(parts).contiguous() if parts else torch.empty(0, dtype=torch.uint8, device="cuda")
            outs.append(backs[r].emit(packed[r][0], reply, bases[r]))
        return b"".join(outs)
    finally:
        for b in backs:
            b.close()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("i", [0, 1, 3, 5])
def test_rmdup_across_virtual_ranks_fastq(i, world, monkeypatch):
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(2800 + i)
    data = dup_fastq(rng, 2000)
    want = oracle.rmdup(data, True, json.dumps(RMDUP_OPTS[i]))
    assert _virtual_ranks(data, True, RMDUP_OPTS[i], world) == want


@pytest.mark.parametrize("world", [1, 2, 5])
@pytest.mark.parametrize("bits", [16, 20])
def test_rmdup_across_virtual_ranks_with_colliding_first_keys(world, bits, monkeypatch):
    """With 16 / 20 bits of k1 many different sequences share their first key at the owner: its grouping (sort + bucket
    tables, round 4) keeps them apart by the second key through the overflow list -- the HBM table it replaced refused."""
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    monkeypatch.setenv("BSK_RMDUP_K1_BITS", str(bits))
    rng = random.Random(5100 + bits)
    data = dup_fastq(rng, 3000)
    for o in ({"BySeq": True}, {"BySeq": True, "IgnoreCase": True}):
        assert _virtual_ranks(data, True, o, world) == oracle.rmdup(data, True, json.dumps(o))


@pytest.mark.parametrize("world", [2, 5])
def test_rmdup_across_virtual_ranks_fasta(world):
    rng = random.Random(41)
    seqs, recs = [], []
    for k in range(500):
        s = seqs[rng.randrange(len(seqs))] if (k > 3 and rng.random() < 0.4) else \
            "".join(rng.choice("ACGTacgt") for _ in range(rng.randint(0, 300)))
        seqs.append(s)
        w = rng.choice([60, 60, 60, 17, max(1, len(s))])
        recs.append(f">s{k}\n" + "".join(s[j:j + w] + "\n" for j in range(0, len(s), w)))
    data = "".join(recs).encode()
    for o in ({"BySeq": True}, {"BySeq": True, "IgnoreCase": True}):
        assert _virtual_ranks(data, False, o, world) == oracle.rmdup(data, False, json.dumps(o))


def test_rmdup_distributed_single_rank_equals_rmdup():
    from bigseqkit_amd import dist as bdist
    rng = random.Random(9)
    data = dup_fastq(rng, 3000)
    b = bdist.HipRmDupBackend(json.dumps({"BySeq": True}), 0)
    try:
        got = bdist.rmdup_distributed(dev(data), bsk.FORMAT_FASTQ, b)
        local_pairs = b.local_pairs
    finally:
        b.close()
    want = oracle.rmdup(data, True, '{"BySeq": true}')
    assert got == bsk.RmDup(frame(data, True), _Opts({"BySeq": True})) == want
    # round 5: the owner's reply names the survivor; with one rank EVERY duplicate's survivor is in the same shard, so every
    # removed record went through the byte comparison of the single-GPU call (bsk_rmdup_dist_emit_ex)
    assert local_pairs == data.count(b"\n") // 4 - want.count(b"\n") // 4 > 0


def test_rmdup_distributed_compares_the_pairs_inside_a_shard():
    """three virtual ranks: the duplicates whose survivor lives in the same shard are byte-compared there (the rest is decided
    by the two keys); the sum over the ranks is what a walk over the oracle's survivors predicts"""
    import torch
    from bigseqkit_amd import dist as bdist
    rng = random.Random(19)
    data = dup_fastq(rng, 4000)
    world = 3
    bounds = bdist.shard_bounds(data, world, bsk.FORMAT_FASTQ)
    shards = [dev(data[lo:hi]) for lo, hi in bounds]
    backs = [bdist.HipRmDupBackend(json.dumps({"BySeq": True}), 0) for _ in range(world)]
    try:
        ns = [b.keys(s, bsk.FORMAT_FASTQ) for b, s in zip(backs, shards)]
        bases = [sum(ns[:r]) for r in range(world)]
        packed = [b.pack(bases[r], world) for r, b in enumerate(backs)]

        def bucket(t, counts, o):
            a = sum(counts[:o])
            return t[a:a + counts[o]]
        keeps, survs = [], []
        for o in range(world):
            recv = torch.cat([bucket(packed[r][0], packed[r][1], o) for r in range(world)])
            k, sv = backs[o].resolve_ex(recv)
            keeps.append(k)
            survs.append(sv)
        outs, pairs = [], 0
        for r in range(world):
            # what comes back to rank r: from every owner o, the slice of its answers that belongs to r's tuples, owner after owner
            rep, srep = [], []
            for o in range(world):
                a = sum(packed[q][1][o] for q in range(r))
                rep.append(keeps[o][a:a + packed[r][1][o]])
                srep.append(survs[o][a:a + packed[r][1][o]])
            outs.append(backs[r].emit(packed[r][0], torch.cat(rep), bases[r], surv_reply=torch.cat(srep)))
            pairs += backs[r].local_pairs
    finally:
        for b in backs:
            b.close()
    assert b"".join(outs) == oracle.rmdup(data, True, '{"BySeq": true}')
    # prediction: a removed record whose first occurrence (file order) lies in the same shard
    seqs = data.split(b"\n")[1::4]
    first, same = {}, 0
    starts = [sum(ns[:r]) for r in range(world)] + [sum(ns)]
    rank_of = lambda i: max(r for r in range(world) if starts[r] <= i)
    for i, s in enumerate(seqs):
        if s in first:
            same += rank_of(first[s]) == rank_of(i)
        else:
            first[s] = i
    assert pairs == same > 0


@pytest.mark.parametrize("i", range(len(TR_OPTS)))
def test_translate_long_records_take_the_block_per_chunk_kernel(i, monkeypatch):
    """records above BSK_LONG_BYTES bases are translated by k_translate_long (16 KiB of an element's body per block)"""
    import random
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    monkeypatch.setenv("BSK_LONG_BYTES", "20000")
    rng = random.Random(50 + i)
    recs = []
    for k, (L, w) in enumerate([(120_001, 60), (12, 60), (70_000, 0), (50_003, 11), (19_999, 60), (65_536, 80)]):
        s = "".join(rng.choice("ACGTacgtN") for _ in range(L))
        if k == 0:
            s = s[:-3] + "TAA"   # a stop codon at the very end (for --trim)
        body = "".join(s[j:j + w] + "\n" for j in range(0, L, w)) if w else s + "\n"
        recs.append(f">chr{k} some description\n{body}")
    data = "".join(recs).encode()
    o = dict(TR_OPTS[i], AllowUnknownCodon=True)
    got = bsk.Translate(bsk.SeqFrame(bsk.FORMAT_FASTA, [dev(data)]), _Opts(o))
    assert got == oracle.translate(data, False, json.dumps(o)), o


def test_rmdup_bucket_path_heavy_duplicates_and_table_cross_check(monkeypatch):
    """the radix-bucket grouping (default) against the oracle and against the one-table path (BSK_RMDUP=table): one
    sequence repeated 60 000 times (a single key: one bucket, one LDS slot), many small groups, and unique records"""
    rng = random.Random(5)
    recs = []
    for i in range(60000):
        recs.append("@same%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i)
    pool = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 40))) for _ in range(5000)]
    for i in range(40000):
        s = rng.choice(pool) if i % 2 else "".join(rng.choice("ACGT") for _ in range(50))
        recs.append("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    rng.shuffle(recs)
    data = "".join(recs).encode()
    want = oracle.rmdup(data, True, json.dumps({"BySeq": True}))
    got = bsk.RmDup(frame(data, True), _Opts({"BySeq": True}))
    assert got == want
    monkeypatch.setenv("BSK_RMDUP", "table")
    assert bsk.RmDup(frame(data, True), _Opts({"BySeq": True})) == want


def test_rmdup_by_seq_on_long_fasta_records_hashed_by_a_wave(monkeypatch):
    """sequences above BSK_LONG_BYTES on wrapped FASTA: XXH64 by k_rmdup_hash_long (4 KiB chunks through LDS, four lanes per
    seed run the accumulator chains) must give the keys of the one-lane code -- same groups, same survivors; lengths around
    the chunk and stripe sizes, duplicates that differ only in case (-i) or in their wrapping"""
    import json
    import random
    import oracle
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    monkeypatch.setenv("BSK_LONG_BYTES", "20000")
    rng = random.Random(777)
    seqs = []
    for L in (20000, 20001, 20479, 20480, 20481, 24575, 24576, 24607, 24608, 32768, 65536 + 31, 100003, 19999):
        seqs.append("".join(rng.choice("ACGTacgtN") for _ in range(L)))
    recs = []
    def put(name, s, w):
        recs.append(">%s\n%s" % (name, "".join(s[j:j + w] + "\n" for j in range(0, len(s), w))))
    for k, s in enumerate(seqs):
        put("a%d" % k, s, 60)
    for k, s in enumerate(seqs[::2]):
        put("dup%d same text other width" % k, s, 70)          # duplicates of a0, a2, ...
    for k, s in enumerate(seqs[1::3]):
        put("case%d" % k, s.swapcase(), 60)                      # duplicates only with -i
    put("last", seqs[3][:-1] + ("A" if seqs[3][-1] != "A" else "C"), 60)   # differs in the last base only
    data = "".join(recs).encode()
    t = dev(data)
    for o in ({"BySeq": True}, {"BySeq": True, "IgnoreCase": True}):
        got = bsk.RmDup(bsk.SeqFrame(bsk.FORMAT_FASTA, [t]), _Opts(o))
        assert got == oracle.rmdup(data, False, json.dumps(o))
    monkeypatch.setenv("BSK_LONG_BYTES", "100000000")           # nothing is long: the one-lane code on the same input
    for o in ({"BySeq": True}, {"BySeq": True, "IgnoreCase": True}):
        assert bsk.RmDup(bsk.SeqFrame(bsk.FORMAT_FASTA, [t]), _Opts(o)) == oracle.rmdup(data, False, json.dumps(o))
