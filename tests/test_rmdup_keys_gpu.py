"""The two 64-bit keys behind `rmdup` (csrc/hash_dev.hpp): k1 must be XXH64(seed 0) of the subject -- the reference's
grouping key, int64(xxhash.Sum64(subject)) (/root/reference/bigseqkit-lib/rmdup.go:67-84), held here to python-xxhash -- and
k2 the second hash, restated below.  Both come from the fused index + hash pass (stream_rmdup.hip: `-s` on FASTQ) and from
the per-record kernels (ops_rmdup.hip: names, IDs, FASTA, BSK_RMDUP_KEYS=off); the two must agree with each other and with
the Python side, whatever the tile / range / quad a sequence falls into.  Then the paths that only a key collision
reaches: BSK_RMDUP_K1_BITS keeps a few bits of k1, so that distinct sequences share a key and the overflow list decides."""
import ctypes as C
import json
import random

import pytest

import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
import oracle

pytestmark = pytest.mark.gpu

M64 = (1 << 64) - 1
Q = [0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0xD6E8FEB86659FD93]
QF1, QF2, QF3 = 0x9FB21C651E98DF25, 0xFF51AFD7ED558CCD, 0xC4CEB9FE1A85EC53


def rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def k2_py(s):
    b = [Q[(k + 1) & 3] for k in range(4)]
    nw = len(s) // 8
    for j in range(nw):
        w = int.from_bytes(s[8 * j:8 * j + 8], "little")
        k = j & 3
        b[k] = rotl(((b[k] ^ w) * Q[k]) & M64, 31)
    rest = int.from_bytes(s[8 * nw:], "little")
    t = b[0] ^ rotl(b[1], 16) ^ rotl(b[2], 32) ^ rotl(b[3], 48)
    t = ((t ^ rest) * QF1) & M64
    t ^= t >> 32
    t = ((t + len(s)) * QF2) & M64
    t ^= t >> 29
    t = (t * QF3) & M64
    t ^= t >> 32
    return t


def device_keys(data, fastq, opts, monkeypatch=None, env=None):
    import torch
    if env and monkeypatch:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    with bsk.Operator("RmDup", json.dumps(opts), 0) as op:
        n = C.c_uint64()
        check(lib.bsk_rmdup_dist_keys(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), bsk.FORMAT_FASTQ if fastq else bsk.FORMAT_FASTA,
                                      None, C.byref(n)), op.ctx)
        k1 = (C.c_uint64 * max(1, n.value))()
        k2 = (C.c_uint64 * max(1, n.value))()
        got = C.c_size_t()
        check(lib.bsk_selftest_rmdup_keys(op.ctx, k1, k2, n.value, C.byref(got)), op.ctx)
        assert got.value == n.value
        return list(k1)[:n.value], list(k2)[:n.value]


def fastq_of(seqs, rng):
    out = []
    for i, s in enumerate(seqs):
        q = bytes(rng.choice(b"#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHI") for _ in s)
        out.append(b"@r%d some text\n%s\n+\n%s\n" % (i, s, q))
    return b"".join(out)


def rand_seq(rng, n, alphabet=b"ACGTacgtN"):
    return bytes(rng.choice(alphabet) for _ in range(n))


@pytest.mark.parametrize("seed,lens", [(1, list(range(0, 200))), (2, [150] * 600), (3, [31, 32, 33, 63, 64, 65, 95, 96, 127, 128, 129] * 20),
                                       (4, [500, 511, 512, 513, 600, 1000, 4000, 4096, 5000, 9000, 20000])])
@pytest.mark.parametrize("fold", [False, True])
def test_fused_keys_are_xxh64_and_k2(seed, lens, fold, monkeypatch):
    xxhash = pytest.importorskip("xxhash")
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(seed)
    seqs = [rand_seq(rng, n) for n in lens]
    rng.shuffle(seqs)
    data = fastq_of(seqs, rng)
    opts = {"BySeq": True, "IgnoreCase": fold}
    k1, k2 = device_keys(data, True, opts)
    assert len(k1) == len(seqs)
    for i, s in enumerate(seqs):
        subj = s.lower() if fold else s
        assert k1[i] == xxhash.xxh64(subj).intdigest(), (i, len(s))
        assert k1[i] == oracle.xxh64(subj)
        assert k2[i] == k2_py(subj), (i, len(s))
    # the per-record kernels compute the same two functions
    monkeypatch.setenv("BSK_RMDUP_KEYS", "off")
    j1, j2 = device_keys(data, True, opts)
    assert j1 == k1 and j2 == k2


# ---- the grouping key of the byte-verifying mode (csrc/hash_dev.hpp "GROUPING key"), restated
P1, P2, P3 = 11400714785074694791, 14029467366897019727, 1609587929392839161


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def gkey_word(p, j):
    v = splitmix64((0x6b73625f67726f75 + p * 4 + (j >> 1)) & M64)
    return (v >> 32) if (j & 1) else (v & 0xFFFFFFFF)


GK = [[gkey_word(p, j) for j in range(8)] for p in range(64)]


def gkey_py(s):
    a1, a2 = [0, 0, 0, 0], [0, 0, 0, 0]       # the sums of the four lanes of a quad (lane k: chunks k, k + 4, ...)
    for c in range((len(s) + 15) // 16):
        ch = s[16 * c:16 * c + 16].ljust(16, b"\0")
        w = [int.from_bytes(ch[4 * d:4 * d + 4], "little") for d in range(4)]
        K, k = GK[c & 63], c & 3
        m32 = 0xFFFFFFFF
        a1[k] = (a1[k] + ((w[0] + K[0]) & m32) * ((w[1] + K[1]) & m32) + ((w[2] + K[2]) & m32) * ((w[3] + K[3]) & m32)) & M64
        a2[k] = (a2[k] + ((w[0] + K[4]) & m32) * ((w[1] + K[5]) & m32) + ((w[2] + K[6]) & m32) * ((w[3] + K[7]) & m32)) & M64
        if (c & 63) >= 60:
            a1[k] = (rotl(a1[k], 29) * P1) & M64
            a2[k] = (rotl(a2[k], 31) * P2) & M64
    s1, s2 = sum(a1) & M64, sum(a2) & M64
    h = (((s1 + len(s)) & M64) * P1 + rotl(s2, 32) * P2) & M64
    h ^= h >> 32
    h = (h * P3) & M64
    h ^= h >> 29
    return h


def group_keys(data, opts, env=None):
    """k1 of every record after a bsk_rmdup_run in the byte-verifying mode"""
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    with bsk.Operator("RmDup", json.dumps(opts), 0) as op:
        for k, v in (env or {}).items():
            check(lib.bsk_ctx_set(op.ctx, k.encode(), v.encode()), op.ctx)
        out = _lib.Out()
        check(lib.bsk_rmdup_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out)), op.ctx)
        buf = C.create_string_buffer(max(1, out.len))
        check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
        n = data.count(b"\n") // 4
        k1 = (C.c_uint64 * max(1, n))()
        got = C.c_size_t()
        check(lib.bsk_selftest_rmdup_keys(op.ctx, k1, None, n, C.byref(got)), op.ctx)
        assert got.value == n
        return list(k1)[:n], buf.raw[:out.len]


@pytest.mark.parametrize("seed,lens", [(1, list(range(0, 200))), (2, [150] * 600), (3, [15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129] * 20),
                                       (4, [500, 511, 512, 513, 600, 1000, 1023, 1024, 1025, 4000, 4096, 5000, 9000, 20000])])
@pytest.mark.parametrize("fold", [False, True])
def test_grouping_key_of_the_verify_mode(seed, lens, fold, monkeypatch):
    """`rmdup -s` (bytes verified) groups by the chain-free key of hash_dev.hpp, whatever tile / range / quad a sequence falls
    into, from LDS or (long and cut lines) from global memory; `rmdup_hash=xxh64` keeps XXH64 there; same survivors."""
    xxhash = pytest.importorskip("xxhash")
    monkeypatch.setenv("BSK_MIN_RANGE_BYTES", "4096")
    rng = random.Random(seed)
    seqs = [rand_seq(rng, n) for n in lens]
    seqs += [rng.choice(seqs) for _ in range(len(seqs) // 3)]
    rng.shuffle(seqs)
    data = fastq_of(seqs, rng)
    opts = {"BySeq": True, "IgnoreCase": fold}
    want = oracle.rmdup(data, True, json.dumps(opts))
    k1, out = group_keys(data, opts)
    assert out == want
    for i, s in enumerate(seqs):
        assert k1[i] == gkey_py(s.lower() if fold else s), (i, len(s))
    x1, out = group_keys(data, opts, {"rmdup_hash": "xxh64"})
    assert out == want
    for i, s in enumerate(seqs):
        assert x1[i] == xxhash.xxh64(s.lower() if fold else s).intdigest(), (i, len(s))


@pytest.mark.parametrize("opts", [{"ByName": True}, {}, {"ByName": True, "IgnoreCase": True}])
def test_name_and_id_keys(opts):
    xxhash = pytest.importorskip("xxhash")
    rng = random.Random(7)
    seqs = [rand_seq(rng, rng.randint(1, 80)) for _ in range(300)]
    data = fastq_of(seqs, rng)
    k1, k2 = device_keys(data, True, opts)
    for i in range(len(seqs)):
        subj = (b"r%d some text" % i) if opts.get("ByName") else (b"r%d" % i)
        if opts.get("IgnoreCase"):
            subj = subj.lower()
        assert k1[i] == xxhash.xxh64(subj).intdigest() and k2[i] == k2_py(subj)


def test_fasta_sequence_keys_wrapped_and_long(monkeypatch):
    xxhash = pytest.importorskip("xxhash")
    monkeypatch.setenv("BSK_LONG_BYTES", "3000")
    rng = random.Random(11)
    seqs = [rand_seq(rng, n) for n in (0, 1, 59, 60, 61, 150, 1000, 2999, 3000, 5000, 20000, 70000)]
    data = b"".join(b">s%d\n" % i + b"".join(s[j:j + 60] + b"\n" for j in range(0, len(s), 60)) for i, s in enumerate(seqs))
    k1, k2 = device_keys(data, False, {"BySeq": True})
    for i, s in enumerate(seqs):
        assert k1[i] == xxhash.xxh64(s).intdigest() and k2[i] == k2_py(s), (i, len(s))


def run_rmdup(data, opts):
    return bsk.RmDup(bsk.SeqFrame(bsk.FORMAT_FASTQ, [data]), type("O", (), {"to_json": lambda self: json.dumps(opts)})())


@pytest.mark.parametrize("bits", [16, 18, 24])
@pytest.mark.parametrize("fold", [False, True])
def test_distinct_sequences_under_one_key_are_kept_apart(bits, fold, monkeypatch):
    """With 16..24 bits of k1 nearly every record shares its key with an unrelated earlier one: the second key must keep
    them apart, the overflow list must still find the real duplicates among them -- the output is the oracle's."""
    rng = random.Random(100 + bits)
    uniq = [rand_seq(rng, rng.choice((36, 150, 151, 250))) for _ in range(1500)]
    seqs = uniq + [rng.choice(uniq) for _ in range(700)] + [rng.choice(uniq).lower() for _ in range(100)]
    rng.shuffle(seqs)
    data = fastq_of(seqs, rng)
    opts = {"BySeq": True, "IgnoreCase": fold}
    want = oracle.rmdup(data, True, json.dumps(opts))
    assert run_rmdup(data, opts) == want
    monkeypatch.setenv("BSK_RMDUP_K1_BITS", str(bits))
    assert run_rmdup(data, opts) == want
    monkeypatch.setenv("BSK_RMDUP_KEYS", "verify")  # ... and the byte comparison of every duplicate agrees
    assert run_rmdup(data, opts) == want
    monkeypatch.setenv("BSK_RMDUP_KEYS", "two-key")  # ... round 3's default: the two keys alone
    assert run_rmdup(data, opts) == want


def test_verify_mode_takes_one_key_and_goes_round_again_on_a_collision():
    """The default groups by k1 ALONE and compares the bytes; a shard on which the comparison meets two different sequences
    under one k1 (forced here with 16 bits of k1) runs once more with both keys and the overflow list -- seen in the
    context's profile as a second streaming pass -- and still gives the oracle's output."""
    import torch
    rng = random.Random(77)
    uniq = [rand_seq(rng, 150, b"ACGT") for _ in range(2500)]
    seqs = uniq + [rng.choice(uniq) for _ in range(900)]
    rng.shuffle(seqs)
    data = fastq_of(seqs, rng)
    want = oracle.rmdup(data, True, json.dumps({"BySeq": True}))
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    for bits, passes in ((None, 1), (b"16", 2)):
        with bsk.Operator("RmDup", json.dumps({"BySeq": True}), 0) as op:
            if bits:
                check(lib.bsk_ctx_set(op.ctx, b"rmdup_k1_bits", bits), op.ctx)
            out = _lib.Out()
            lib.bsk_profile_reset(op.ctx)
            lib.bsk_profile_enable(op.ctx, 1)
            check(lib.bsk_rmdup_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out)), op.ctx)
            buf = C.create_string_buffer(max(1, out.len))
            check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
            pb = C.create_string_buffer(4096)
            check(lib.bsk_profile_dump(op.ctx, pb, len(pb)), op.ctx)
            assert buf.raw[:out.len] == want
            prof = dict(kv.split("=") for kv in pb.value.decode().split(";") if kv)  # name=ms/launches;
            assert int(prof["k_rmdup_stream"].split("/")[1]) == passes, prof


@pytest.mark.parametrize("nrec", [1, 7, 2047, 2048, 2049, 4096, 30011])
def test_one_pass_placement_equals_the_separate_passes(nrec, monkeypatch):
    """k_rmdup_place (sizes + byte comparison + output offsets by decoupled look-back + segment list in one pass) against
    round 4's verify / scan / segment-build passes (`rmdup_place=off`) and the oracle: block boundaries at 2 048 records,
    '+' lines that repeat the name and a last record without a newline (left to the record-wise emit), a record above the
    'long' threshold (the general size pass takes over)."""
    rng = random.Random(nrec)
    uniq = [rand_seq(rng, rng.choice((20, 36, 150, 151)), b"ACGT") for _ in range(max(1, nrec * 2 // 3))]
    seqs = [rng.choice(uniq) for _ in range(nrec)]
    recs = []
    for i, s in enumerate(seqs):
        q = bytes(rng.choice(b"#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHI") for _ in s)
        plus = (b"+r%d x" % i) if (i % 97 == 5) else b"+"
        recs.append(b"@r%d x\n%s\n%s\n%s\n" % (i, s, plus, q))
    data = b"".join(recs)
    for variant in (data, data[:-1]):
        want = oracle.rmdup(variant, True, '{"BySeq": true}')
        assert run_rmdup(variant, {"BySeq": True}) == want
        monkeypatch.setenv("BSK_RMDUP_PLACE", "off")
        assert run_rmdup(variant, {"BySeq": True}) == want
        monkeypatch.delenv("BSK_RMDUP_PLACE")
    if nrec >= 2048:
        monkeypatch.setenv("BSK_LONG_BYTES", "300")   # records of 150 bases are "long": the list-writing size pass runs
        assert run_rmdup(data, {"BySeq": True}) == oracle.rmdup(data, True, '{"BySeq": true}')


@pytest.mark.parametrize("keys", ["", "off"])
def test_bucket_pass_by_hand_equals_the_library_sort(keys, monkeypatch):
    """BSK_RMDUP_BUCKETS=hand: one 16-bit histogram + an unstable scatter instead of rocPRIM's two digit passes (slower, kept
    as a measurement: ops_rmdup.hip) -- the LDS tables take a bucket's pairs in any order, so the answer is the same"""
    rng = random.Random(4242)
    uniq = [rand_seq(rng, rng.choice((36, 150, 151))) for _ in range(3000)]
    seqs = uniq + [rng.choice(uniq) for _ in range(1500)]
    rng.shuffle(seqs)
    data = fastq_of(seqs, rng)
    if keys:
        monkeypatch.setenv("BSK_RMDUP_KEYS", keys)
    want = oracle.rmdup(data, True, '{"BySeq": true}')
    assert run_rmdup(data, {"BySeq": True}) == want
    monkeypatch.setenv("BSK_RMDUP_BUCKETS", "hand")
    assert run_rmdup(data, {"BySeq": True}) == want
    assert bsk.RmDup(bsk.SeqFrame(bsk.FORMAT_FASTQ, [data]), type("O", (), {"to_json": lambda self: '{"ByName": true}'})()) == \
        oracle.rmdup(data, True, '{"ByName": true}')


def test_key_path_equals_byte_path_on_c5_layout(monkeypatch):
    import torch
    rb, nrec = 317, 400_000
    t = torch.empty(rb * nrec, dtype=torch.uint8, device="cuda")
    assert lib.bsk_synth_device(0, 42, _lib.SYNTH_FLAG_DUPS, 0, C.c_void_p(t.data_ptr()), rb * nrec, 0, None) == 0
    data = bytes(t.cpu().numpy().tobytes())
    a = run_rmdup(data, {"BySeq": True})
    assert len(a) == rb * (nrec - nrec // 5)
    monkeypatch.setenv("BSK_RMDUP_KEYS", "off")
    assert run_rmdup(data, {"BySeq": True}) == a
    monkeypatch.setenv("BSK_RMDUP_KEYS", "verify")
    assert run_rmdup(data, {"BySeq": True}) == a


def test_switches_belong_to_a_context_not_to_the_process():
    """bsk_ctx_set: two contexts of one process take different paths at the same time (the byte-comparing path and the
    key path), no environment variable involved, same survivors"""
    import torch
    rng = random.Random(3)
    uniq = [rand_seq(rng, 150, b"ACGT") for _ in range(3000)]
    seqs = uniq + [rng.choice(uniq) for _ in range(1500)]
    rng.shuffle(seqs)
    data = fastq_of(seqs, rng)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    outs = []
    with bsk.Operator("RmDup", json.dumps({"BySeq": True}), 0) as a, bsk.Operator("RmDup", json.dumps({"BySeq": True}), 0) as b:
        check(lib.bsk_ctx_set(b.ctx, b"rmdup_keys", b"off"), b.ctx)
        check(lib.bsk_ctx_set(b.ctx, b"segcopy", b"off"), b.ctx)
        for op in (a, b, a):
            out = _lib.Out()
            lib.bsk_profile_reset(op.ctx)
            lib.bsk_profile_enable(op.ctx, 1)
            check(lib.bsk_rmdup_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out)), op.ctx)
            buf = C.create_string_buffer(max(1, out.len))
            check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
            pb = C.create_string_buffer(4096)
            check(lib.bsk_profile_dump(op.ctx, pb, len(pb)), op.ctx)
            outs.append((buf.raw[:out.len], pb.value.decode()))
    assert outs[0][0] == outs[1][0] == outs[2][0] == oracle.rmdup(data, True, json.dumps({"BySeq": True}))
    assert "k_rmdup_stream" in outs[0][1] and "k_rmdup_stream" in outs[2][1]      # context a: the fused pass
    assert "k_rmdup_stream" not in outs[1][1] and "k_rmdup_hash" in outs[1][1]     # context b: its own switches
