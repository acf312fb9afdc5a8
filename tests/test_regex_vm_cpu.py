"""The position-reporting matcher (regex_vm.hpp: Thompson program + Pike VM, leftmost-first like Go's regexp) against
Python's `re` on the syntax both share: bounds of the match and of capture group 1, searched from several offsets.
Host build of the same routine the device runs."""
import ctypes as C
import random
import zlib
import re

import pytest

from bigseqkit_amd._lib import lib

NONE = 0xFFFFFFFF


def find(expr, text, start=0):
    caps = (C.c_uint32 * 4)()
    ng = C.c_uint32()
    r = lib.bsk_selftest_regex_find(expr.encode(), text, len(text), start, caps, C.byref(ng))
    assert r >= 0, lib.bsk_global_error()
    return (tuple(caps), ng.value) if r else (None, ng.value)


def py_find(expr, text, start):
    flags = 0
    m = re.match(r"^\(\?([ismU]+)\)", expr)
    body = expr
    if m:
        body = expr[m.end():]
        if "i" in m.group(1): flags |= re.I
        if "s" in m.group(1): flags |= re.S
        if "m" in m.group(1): flags |= re.M  # (the texts of this test hold no line break: the targets of the library never do)
    body = body.replace("[[:digit:]]", "[0-9]").replace("[[:alpha:]]", "[A-Za-z]")  # POSIX names are RE2 syntax, not Python's
    # ^ / $ : Go without (?m) anchors at the ends of the text only; search from `start` keeps position 0 as the beginning
    rx = re.compile(body.encode(), flags)
    mm = rx.search(text, start)
    if not mm:
        return None
    g1 = mm.span(1) if rx.groups >= 1 and mm.group(1) is not None else (NONE, NONE)
    return (mm.start(), mm.end(), g1[0] if g1[0] >= 0 else NONE, g1[1] if g1[1] >= 0 else NONE)


EXPRS = [r"^(\S+)\s?", r"\|([^\|]+)\| ", r"^([^ ]+) ", r"(\d+)$", r"id=(\w+)", r"^gi\|(\d+)\|", r"(a|ab)(c|bcd)", r"(a+)(a*)", r"(a+?)(a*)",
         r"x*", r"(x*)y", r"(?i)ac(g+)t", r"[^ ]+ (.+)$", r"^(.*?)_", r"^(.*)_", r"(AC|ACG|ACGT)T?", r"(A{2,4})C", r"(A{2,4}?)C", r"A(C|G){0,2}T",
         r"(?:AB)+(C)", r"(?P<name>[A-Z]+)\d", r"\.(\w+)$", r"([ACGT]+)N+([ACGT]+)", r"(GA|G)(AT|A)T", r"(T[AG]A)", r"AC+G", r"^A.*T$",
         r"(AC|GT){2}", r"()", r"(a|b)*c", r"(\w+)\s(\w+)", r"[[:digit:]]+([[:alpha:]]*)",
         # ASCII word boundaries (RE2 \b \B; round 4)
         r"\bid=(\w+)\b", r"(\w+)\b", r"\B(a+)", r"\b(AC|GT)+\b", r"x\b", r"\b", r"a\Bb", r"\b(\d+)\b", r"(?i)\bacg(t*)\b", r"\B", r"(\S+)\b ", r"(?m)^(\w+)$", r"(?im)a(c+)$"]


@pytest.mark.parametrize("expr", EXPRS)
def test_same_spans_as_python_re(expr):
    rng = random.Random(zlib.crc32(repr(expr).encode()) & 0xFFFF)  # (str hashes differ from process to process)
    alpha = "ACGTNacgt|_ .=xyab01239d-"
    texts = [b"", b"a", b"aaa", b"gi|110645304|ref|NC_002516.2| Pseudomonas", b"seq1 desc id=ab_9 end", b"ACGTACGT", b"xxxy", b"abcd",
             b"read_12_x 77", b"AAAAC", b"GAAT", b"TAA TGA"]
    for _ in range(120):
        texts.append("".join(rng.choice(alpha) for _ in range(rng.randint(0, 40))).encode())
    for t in texts:
        for start in {0, min(1, len(t)), len(t) // 2, len(t)}:
            got, ng = find(expr, t, start)
            want = py_find(expr, t, start)
            if expr == r"\B" and t == b"":  # Python < 3.14 never matches \B in an empty text; RE2 / Go do (no boundary there)
                want = (0, 0, NONE, NONE)
            assert got == want, (expr, t, start, got, want)


def test_group_count_and_rejections():
    assert find(r"(a)(b)(?:c)(?P<x>d)", b"abcd")[1] == 3
    assert find(r"abc", b"abc")[1] == 0
    for bad in [r"[\b]", r"a(?m)b", r"\pL", "[ab]{40}[cd]{40}"]:
        caps = (C.c_uint32 * 4)()
        assert lib.bsk_selftest_regex_find(bad.encode(), b"x", 1, 0, caps, None) == -1
