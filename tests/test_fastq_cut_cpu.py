"""Where a host-resident FASTQ text may be cut (bsk_find_record_start -- ReadFixer's job, /root/reference/bigseqkit-lib/
helper.go:41-66, for shards and staging chunks): strict 4-line files AND files whose sequence / quality text is wrapped over
several lines (SeqParser reads those, helper.go:252-269; round 4).  The answer must be the first TRUE record start at or
after the asked position -- true = reached by reading the file from its beginning under the grammar of PARITY.md SPLIT-FQ."""
import ctypes as C
import random

import pytest

import bigseqkit_amd as bsk
from bigseqkit_amd._lib import lib, check


def cut(data, pos):
    out = C.c_size_t()
    check(lib.bsk_find_record_start(data, len(data), pos, bsk.FORMAT_FASTQ, C.byref(out)))
    return out.value


def make(rng, nrec, width, strict_every=0):
    """records and their start offsets; qualities begin with '@' / '+' now and then, continuation lines with '+'"""
    out, starts, at = [], [], 0
    for i in range(nrec):
        w = 10 ** 9 if (strict_every and i % strict_every == 0) else width
        b = min(width, 80)
        L = rng.choice([0, 1, b - 1, b, b + 1, 3 * b, rng.randint(0, 8 * b)])
        seq = "".join(rng.choice("ACGTN") for _ in range(L))
        qual = [chr(rng.randint(33, 126)) for _ in range(L)]
        for k in range(0, L, w):
            if k and qual[k] == "@":
                qual[k] = "A"          # ('@' at the start of a continuation line begins a record under the grammar)
            elif k and rng.random() < 0.3:
                qual[k] = "+"
        if L and rng.random() < 0.3:
            qual[0] = rng.choice("@+")
        qual = "".join(qual)
        sl = [seq[j:j + w] for j in range(0, L, w)] or [""]
        ql = [qual[j:j + w] for j in range(0, L, w)] or [""]
        rec = "@r%d x\n%s\n+%s\n%s\n" % (i, "\n".join(sl), "r%d x" % i if rng.random() < 0.3 else "", "\n".join(ql))
        starts.append(at)
        out.append(rec)
        at += len(rec)
    return "".join(out).encode(), starts


@pytest.mark.parametrize("width,strict_every", [(10 ** 9, 0), (60, 0), (80, 0), (7, 0), (25, 3), (13, 50)])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_cut_is_the_next_true_record_start(width, strict_every, seed):
    rng = random.Random(seed * 1000 + min(width, 999))
    data, starts = make(rng, 400, width, strict_every)
    n = len(data)
    import bisect
    positions = [0, 1, n - 1, n] + [rng.randrange(n) for _ in range(300)] + starts[:40] + [s + 1 for s in starts[:40]]
    for pos in positions:
        got = cut(data, pos)
        k = bisect.bisect_left(starts, pos)
        true_next = starts[k] if k < len(starts) else n
        if strict_every == 0 or width == 10 ** 9:
            assert got == true_next, (pos, got, true_next)
        else:
            # a file that mixes one-line and wrapped records: the strict rule may answer with a LATER record that is on one
            # line; any true start at or after `pos` within reach is a valid cut
            assert got == n or (got >= pos and got in set(starts)), (pos, got, true_next)
    # without a final line break, and with blank lines behind the last record
    for tail in (data[:-1], data + b"\n\n"):
        for pos in [rng.randrange(len(tail)) for _ in range(60)]:
            got = cut(tail, pos)
            k = bisect.bisect_left(starts, pos)
            assert got == (starts[k] if k < len(starts) else len(tail)) or (strict_every and got in set(starts)), (pos, got)


def test_shards_of_a_wrapped_file_cover_it_record_by_record():
    from bigseqkit_amd import dist as bdist
    rng = random.Random(99)
    data, starts = make(rng, 2000, 60)
    for world in (2, 3, 7):
        b = bdist.shard_bounds(data, world, bsk.FORMAT_FASTQ)
        assert b[0][0] == 0 and b[-1][1] == len(data)
        for (lo, hi), (lo2, _) in zip(b, b[1:]):
            assert hi == lo2
        assert all(lo in set(starts) or lo == len(data) for lo, _ in b)
        assert len({lo for lo, _ in b}) == world   # (real cuts: the wrapped reading found record starts, not the end)


def _accident():
    """a wrapped record whose quality lines read, from the second of them on, as a record of their own: '@'-led quality line,
    two lines taken for bases, a '+'-led quality line, two lines of 'qualities' -- then genuine records"""
    w = 10
    seq = "ACGTACGTAC" * 7
    qual_lines = ["IIIIIIIIII", "@IIIIIIIII", "IIIIIIIIII", "IIIIIIIIII", "+IIIIIIIII", "IIIIIIIIII", "IIIIIIIIII"]
    victim = "@victim\n" + "\n".join(seq[j:j + w] for j in range(0, len(seq), w)) + "\n+\n" + "\n".join(qual_lines) + "\n"
    plain = "".join("@r%d\nACGTACGTAC\nACGTA\n+\nIIIIIIIIII\nIIIII\n" % i for i in range(6))
    before = "".join("@p%d\nACGTACGTAC\nACG\n+\nIIIIIIIIII\nIII\n" % i for i in range(4))
    return before, victim, plain


def test_a_quality_line_that_reads_as_a_record_is_no_place_to_cut():
    before, victim, plain = _accident()
    data = (before + victim + plain).encode()
    v0 = len(before)
    false_start = v0 + victim.index("\n@IIIIIIIII") + 1
    true_next = v0 + len(victim)
    # the accident is real: read from the '@'-led quality line on, the text IS three records in a row
    assert cut(data[false_start:], 0) == 0
    # with the text before it in view the record that spans the line refutes it, wherever the search begins inside the victim
    for pos in (v0 + 1, v0 + 20, false_start - 3, false_start):
        assert cut(data, pos) == true_next, pos
    assert cut(data, v0) == v0 and cut(data, true_next) == true_next
    # every genuine start still stands, also the first one of the text and the ones a window begins inside of
    starts = [m for m in range(len(data)) if data[m:m + 1] == b"@" and (m == 0 or data[m - 1:m] == b"\n")
              and not (v0 < m < true_next)]
    for s in starts:
        assert cut(data, s) == s
        for a in (s - 1, s - 7, max(0, s - 40)):
            if a >= 0:
                assert cut(data[a:], s - a) == s - a, (s, a)
