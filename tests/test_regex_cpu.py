"""The regular-expression compiler of the HIP path (Go regexp / RE2 syntax subset -> position automaton,
bigseqkit_amd/csrc/regex_nfa.cpp) checked on the host against Python's `re` (an independent engine that agrees with RE2
on this subset for targets without newlines).  Reference call site: re.Match(target), bigseqkit-lib/grep.go:459-468."""
import ctypes as C
import random
import re

import pytest

from bigseqkit_amd._lib import lib


def bsk_match(expr, text):
    m = C.c_int()
    buf = (C.c_ubyte * max(1, len(text))).from_buffer_copy(text or b"\0")
    rc = lib.bsk_regex_match(expr.encode(), C.cast(buf, C.c_void_p), len(text), C.byref(m))
    if rc != 0:
        raise ValueError(lib.bsk_global_error().decode())
    return bool(m.value)


CASES = [
    ("ACGT", [b"ACGT", b"TTACGTT", b"ACG", b"", b"acgt"]),
    ("^ACG", [b"ACGT", b"TACG", b""]),
    ("ACG$", [b"TTACG", b"ACGT", b""]),
    ("^$", [b"", b"A"]),
    ("^A*$", [b"", b"AAAA", b"AAB"]),
    ("A[CG]T", [b"ACT", b"AGT", b"ATT", b"xxAGTxx"]),
    ("A[^CG]T", [b"ACT", b"AAT", b"ATT"]),
    ("A.T", [b"AxT", b"AT", b"A\tT"]),
    ("(AC|GT)+T", [b"ACT", b"GTACT", b"T", b"ACAC"]),
    ("A{3}", [b"AA", b"AAA", b"TAAAAT"]),
    ("^A{2,3}$", [b"A", b"AA", b"AAA", b"AAAA"]),
    ("^A{2,}$", [b"A", b"AA", b"AAAAAAAA"]),
    ("A[TU]G(?:.{3})+?[TU](?:AG|AA|GA)", [b"ATGCCCTAG", b"ATGTAG", b"AUGCCCAAAUGA", b"ATGCCTAG"]),
    (r"^chr\d+$", [b"chr1", b"chr22", b"chrX", b"xchr1"]),
    (r"\w+\s\w+", [b"ab cd", b"abcd", b"a b"]),
    (r"gi\|(\d+)\|", [b"gi|12345|ref", b"gi||ref"]),
    (r"[[:alpha:]]+[[:digit:]]", [b"abc1", b"1abc", b"a1"]),
    ("(?i)acgt", [b"ACGT", b"AcGt", b"ACGA"]),
    ("(?i)[a-c]x[^d]", [b"BXE", b"bxd", b"BXD", b"cxe"]),
    (r"\.fa$", [b"x.fa", b"xfa", b"x.fa.gz"]),
    ("a|", [b"", b"b"]),
    ("(a|b)*abb", [b"abb", b"aabb", b"babb", b"ab"]),
    ("x{0}y", [b"y", b"xy"]),
    ("[]a]+", [b"]a]", b"b"]),
    (r"\x41\x2a", [b"A*", b"AA"]),
]


@pytest.mark.parametrize("expr,texts", CASES)
def test_regex_cases_equal_python_re(expr, texts):
    pat = re.compile(expr.replace("[[:alpha:]]", "[A-Za-z]").replace("[[:digit:]]", "[0-9]").encode())
    for t in texts:
        assert bsk_match(expr, t) == (pat.search(t) is not None), (expr, t)


def test_re2_specific_readings():
    # "{,n}" is not a repetition in RE2 (Python reads it as {0,n}): the braces are literals
    assert bsk_match("a{,2}", b"a{,2}") and not bsk_match("^a{,2}$", b"aa")
    # $ without (?m) is the end of the text only
    assert not bsk_match("A$", b"A\n")


def _rand_regex(rng, depth=0):
    r = rng.random()
    if depth > 3 or r < 0.35:
        k = rng.random()
        if k < 0.5:
            return rng.choice("ACGT")
        if k < 0.7:
            s = "".join(sorted(set(rng.choice("ACGTN") for _ in range(rng.randint(1, 3)))))
            return "[" + ("^" if rng.random() < 0.3 else "") + s + "]"
        if k < 0.8:
            return "."
        return rng.choice(["AC", "GT", "TTT"])
    if r < 0.6:
        return _rand_regex(rng, depth + 1) + _rand_regex(rng, depth + 1)
    if r < 0.75:
        return "(" + _rand_regex(rng, depth + 1) + "|" + _rand_regex(rng, depth + 1) + ")"
    q = rng.choice(["*", "+", "?", "{2}", "{1,3}", "{2,}", "*?", "+?"])
    return "(?:" + _rand_regex(rng, depth + 1) + ")" + q


def test_random_regexes_equal_python_re():
    rng = random.Random(12345)
    checked = 0
    for _ in range(400):
        body = _rand_regex(rng)
        expr = ("^" if rng.random() < 0.2 else "") + body + ("$" if rng.random() < 0.2 else "")
        try:
            pat = re.compile(expr.encode())
        except re.error:
            continue
        try:
            bsk_match(expr, b"")
        except ValueError as e:
            assert "64 positions" in str(e), (expr, e)   # only the size limit may reject these
            continue
        for _ in range(25):
            t = "".join(rng.choice("ACGTN") for _ in range(rng.randint(0, 30))).encode()
            assert bsk_match(expr, t) == (pat.search(t) is not None), (expr, t)
            checked += 1
    assert checked > 5000


def test_regex_errors_are_explicit():
    for expr, msg in [("a(b", "missing closing )"), ("a[b", "missing closing ]"), ("*a", "missing argument to repetition operator"),
                      (r"a\b", "word boundary"), (r"\pL", "Unicode class"), ("a{2000}", "invalid repeat count"),
                      ("(?:ACGT){20}", "more than 64 positions"), ("a(?=b)", "not supported")]:
        with pytest.raises(ValueError) as e:
            bsk_match(expr, b"x")
        assert msg in str(e.value), (expr, str(e.value))
