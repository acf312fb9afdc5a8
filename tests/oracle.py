"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE."""
import ctypes as C
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))


class KitConfig(C.Structure):
    _fields_ = [("SeqType", C.c_char_p), ("LineWidth", C.c_int), ("IDRegexp", C.c_char_p), ("IDNCBI", C.c_int),
                ("Quiet", C.c_int), ("AlphabetGuessSeqLength", C.c_int), ("ValidateSeqLength", C.c_int)]


class StatsOpts(C.Structure):
    _fields_ = [("Config", KitConfig), ("Tabular", C.c_int), ("GapLetters", C.c_char_p), ("All", C.c_int),
                ("SkipErr", C.c_int), ("FqEncoding", C.c_char_p), ("Basename", C.c_int)]


class SeqOpts(C.Structure):
    _fields_ = [("Config", KitConfig)] + [(k, C.c_int) for k in
                ("Reverse", "Complement", "Name", "Seq", "Qual", "OnlyId", "RemoveGaps")] + \
               [("GapLetters", C.c_char_p)] + [(k, C.c_int) for k in
                ("LowerCase", "UpperCase", "Dna2rna", "Rna2dna", "ValidateSeq", "ValidateSeqLength", "MaxLen",
                 "MinLen", "QualAsciiBase")] + [("MinQual", C.c_double), ("MaxQual", C.c_double)]


class GrepOpts(C.Structure):
    _fields_ = [("Config", KitConfig), ("Pattern", C.POINTER(C.c_char_p)), ("npattern", C.c_int)] + \
               [(k, C.c_int) for k in ("InvertMatch", "ByName", "BySeq", "OnlyPositiveStrand", "IgnoreCase")] + \
               [("Region", C.c_char_p)] + [(k, C.c_int) for k in
                ("Circular", "Count", "UseRegexp", "Degenerate", "MaxMismatch", "DeleteMatched")] + \
               [("PatternFile", C.c_char_p)]


class SubseqOpts(C.Structure):
    _fields_ = [("Config", KitConfig), ("Region", C.c_char_p), ("UpStream", C.c_int), ("DownStream", C.c_int),
                ("OnlyFlank", C.c_int), ("Gtf", C.c_char_p), ("Bed", C.c_char_p), ("Chr", C.c_char_p),
                ("Feature", C.c_char_p), ("GtfTag", C.c_char_p)]


class LocateOpts(C.Structure):
    _fields_ = [("Config", KitConfig), ("Pattern", C.POINTER(C.c_char_p)), ("npattern", C.c_int)] + \
               [(k, C.c_int) for k in ("IgnoreCase", "OnlyPositiveStrand", "NonGreedy", "Gtf", "Bed", "HideMatched",
                                       "Circular", "Degenerate", "UseRegexp", "UseFmi", "MaxMismatch")] + \
               [("PatternFile", C.c_char_p)]


class TranslateOpts(C.Structure):
    _fields_ = [("Config", KitConfig), ("TranslTable", C.c_int), ("Frame", C.POINTER(C.c_char_p)), ("nframe", C.c_int)] + \
               [(k, C.c_int) for k in ("Trim", "Clean", "AllowUnknownCodon", "InitCodonAsM", "ListTranslTable",
                                       "ListTranslTableWithAmbCodons", "AppendFrame")]


class RmDupOpts(C.Structure):
    _fields_ = [("Config", KitConfig)] + [(k, C.c_int) for k in ("ByName", "BySeq", "IgnoreCase", "OnlyPositiveStrand")]


class OracleError(RuntimeError):
    pass


def _cfg(d):
    c = d.get("Config") or {}
    g = lambda k, dv: dv if c.get(k) is None else c[k]
    return KitConfig(g("SeqType", "auto").encode(), g("LineWidth", 60), g("IDRegexp", r"^(\S+)\s?").encode(),
                     int(g("IDNCBI", False)), int(g("Quiet", False)), g("AlphabetGuessSeqLength", 10000),
                     g("ValidateSeqLength", 10000))


def stats_opts(opts_json):
    """Fill the oracle's plain struct from the same JSON the product eats (defaults per
    /root/reference/bigseqkit/stats.go:28-38)."""
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    g = lambda k, dv: dv if d.get(k) is None else d[k]
    return StatsOpts(_cfg(d), int(g("Tabular", False)), g("GapLetters", "- .").encode(), int(g("All", False)),
                     int(g("SkipErr", False)), g("FqEncoding", "sanger").encode(), int(g("Basename", False)))


def seq_opts(opts_json):
    """defaults per /root/reference/bigseqkit/seq.go:32-55"""
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    g = lambda k, dv: dv if d.get(k) is None else d[k]
    b = lambda k: int(bool(g(k, False)))
    return SeqOpts(_cfg(d), b("Reverse"), b("Complement"), b("Name"), b("Seq"), b("Qual"), b("OnlyId"),
                   b("RemoveGaps"), g("GapLetters", "- \t.").encode(), b("LowerCase"), b("UpperCase"), b("Dna2rna"),
                   b("Rna2dna"), b("ValidateSeq"), g("ValidateSeqLength", 10000), g("MaxLen", -1), g("MinLen", -1),
                   g("QualAsciiBase", 33), float(g("MinQual", -1)), float(g("MaxQual", -1)))


def _deep(call):
    """run `call` on a thread with a 1 GiB stack: libstdc++'s std::regex (the stand-in for Go's regexp in `grep -r` /
    `locate -r` / --id-regexp) matches recursively -- `^A.*T$` on a read of 17 821 bases overflowed the 8 MiB of the main
    thread (fuzz seed 423, round 6: the CHECKER crashed, not the product)"""
    import threading
    res = []

    def body():
        try:
            res.append((True, call()))
        except BaseException as e:  # noqa: BLE001 -- handed to the caller's thread
            res.append((False, e))

    old = threading.stack_size(1 << 30)
    try:
        t = threading.Thread(target=body)
        t.start()
    finally:
        threading.stack_size(old)
    t.join()
    ok, v = res[0]
    if not ok:
        raise v
    return v


def _run_text(fn, data, fastq, o, nparts):
    cap = 4 * len(data) + 4096
    while True:
        out, n, nrec, err = C.create_string_buffer(cap), C.c_size_t(), C.c_uint64(), C.create_string_buffer(_ERR)
        call = lambda: fn(_buf(data), C.c_size_t(len(data)), int(fastq), C.byref(o), nparts, out, C.c_size_t(cap), C.byref(n),
                          C.byref(nrec), err, _ERR)
        rc = _deep(call) if len(data) > 2000 else call()
        if rc == 2:
            cap = n.value + 16
            continue
        if rc:
            raise OracleError(err.value.decode())
        return out.raw[:n.value], nrec.value


def seq(data, fastq, opts_json="{}", nparts=1):
    """SeqTransform -> the bytes FileStore would write (each element + newline)."""
    return _run_text(_lib.orc_seq, data, fastq, seq_opts(opts_json), nparts)[0]


def grep_opts(opts_json):
    """defaults per /root/reference/bigseqkit/grep.go:31-49"""
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    g = lambda k, dv: dv if d.get(k) is None else d[k]
    b = lambda k: int(bool(g(k, False)))
    pats = [p.encode() for p in g("Pattern", [""])]
    arr = (C.c_char_p * max(1, len(pats)))(*pats)
    o = GrepOpts(_cfg(d), arr, len(pats), b("InvertMatch"), b("ByName"), b("BySeq"), b("OnlyPositiveStrand"),
                 b("IgnoreCase"), g("Region", "").encode(), b("Circular"), b("Count"), b("UseRegexp"),
                 b("Degenerate"), g("MaxMismatch", 0), b("DeleteMatched"), g("PatternFile", "").encode())
    o._keep = (arr, pats)
    return o


def grep(data, fastq, opts_json="{}", nparts=1):
    return _run_text(_lib.orc_grep, data, fastq, grep_opts(opts_json), nparts)[0]


def subseq_opts(opts_json):
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    g = lambda k, dv: dv if d.get(k) is None else d[k]
    return SubseqOpts(_cfg(d), g("Region", "").encode(), g("UpStream", 0), g("DownStream", 0),
                      int(bool(g("OnlyFlank", False))), g("Gtf", "").encode(), g("Bed", "").encode(),
                      "\n".join(g("Chr", [])).encode(), "\n".join(g("Feature", [])).encode(),
                      g("GtfTag", "").encode())


def subseq(data, fastq, opts_json="{}", nparts=1):
    return _run_text(_lib.orc_subseq, data, fastq, subseq_opts(opts_json), nparts)[0]


def locate_opts(opts_json):
    """defaults per /root/reference/bigseqkit/locate.go:27-45"""
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    g = lambda k, dv: dv if d.get(k) is None else d[k]
    b = lambda k: int(bool(g(k, False)))
    pats = [p.encode() for p in g("Pattern", [""])]
    arr = (C.c_char_p * max(1, len(pats)))(*pats)
    o = LocateOpts(_cfg(d), arr, len(pats), b("IgnoreCase"), b("OnlyPositiveStrand"), b("NonGreedy"), b("Gtf"),
                   b("Bed"), b("HideMatched"), b("Circular"), b("Degenerate"), b("UseRegexp"), b("UseFmi"),
                   g("MaxMismatch", 0), g("PatternFile", "").encode())
    o._keep = (arr, pats)
    return o


def locate(data, fastq, opts_json="{}", nparts=1):
    return _run_text(_lib.orc_locate, data, fastq, locate_opts(opts_json), nparts)[0]


def translate_opts(opts_json):
    """defaults per /root/reference/bigseqkit/translate.go:22-35"""
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    g = lambda k, dv: dv if d.get(k) is None else d[k]
    b = lambda k: int(bool(g(k, False)))
    fr = [f.encode() for f in g("Frame", ["1"])]
    arr = (C.c_char_p * max(1, len(fr)))(*fr)
    o = TranslateOpts(_cfg(d), g("TranslTable", 1), arr, len(fr), b("Trim"), b("Clean"), b("AllowUnknownCodon"),
                      b("InitCodonAsM"), g("ListTranslTable", -1), g("ListTranslTableWithAmbCodons", -1),
                      b("AppendFrame"))
    o._keep = (arr, fr)
    return o


def translate(data, fastq, opts_json="{}", nparts=1):
    return _run_text(_lib.orc_translate, data, fastq, translate_opts(opts_json), nparts)[0]


def translate_seq(seq, table=1, frame=1, trim=False, clean=False, allow_unknown=False, init_m=False):
    out = C.create_string_buffer(len(seq) + 8)
    rc = _lib.orc_translate_seq(seq.encode(), table, frame, int(trim), int(clean), int(allow_unknown), int(init_m),
                                out, C.c_size_t(len(out)))
    if rc == 3:
        raise OracleError("unknown codon")
    if rc:
        raise OracleError("translate")
    return out.value.decode()


def genetic_code(table):
    """(ncbieaa, sncbieaa) as the oracle derives them from the standard code + NCBI's documented differences"""
    a, b = C.create_string_buffer(65), C.create_string_buffer(65)
    if not (_lib.orc_genetic_code(table, 0, a) and _lib.orc_genetic_code(table, 1, b)):
        return None
    return a.value.decode(), b.value.decode()


def rmdup_opts(opts_json):
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    b = lambda k: int(bool(d.get(k)))
    return RmDupOpts(_cfg(d), b("ByName"), b("BySeq"), b("IgnoreCase"), b("OnlyPositiveStrand"))


def rmdup(data, fastq, opts_json="{}"):
    return _run_text(_lib.orc_rmdup, data, fastq, rmdup_opts(opts_json), 1)[0]


def rmdup_mt(data, fastq, opts_json="{}", threads=4):
    """rmdup on `threads` host threads (rmdup_call_mt: bench.py's all-cores baseline); the nparts slot carries the thread count"""
    return _run_text(_lib.orc_rmdup_mt, data, fastq, rmdup_opts(opts_json), threads)[0]


def rmdup_side(data, fastq, opts_json="{}", which=1):
    """which=1: text of the removed records (-d); which=2: the duplicate-number lines (-D)"""
    o = rmdup_opts(opts_json)
    cap = 4 * len(data) + 4096
    out, n, err = C.create_string_buffer(cap), C.c_size_t(), C.create_string_buffer(_ERR)
    rc = _lib.orc_rmdup_side(_buf(data), C.c_size_t(len(data)), int(fastq), C.byref(o), which, out, C.c_size_t(cap),
                             C.byref(n), err, _ERR)
    if rc:
        raise OracleError(err.value.decode() or "rmdup_side failed")
    return out.raw[:n.value]


_lib.orc_xxh64.restype = C.c_uint64


def xxh64(b):
    return _lib.orc_xxh64(_buf(b), C.c_size_t(len(b)))


def sub_location(length, start, end):
    b, e = C.c_size_t(), C.c_size_t()
    _lib.orc_sub_location(C.c_size_t(length), start, end, C.byref(b), C.byref(e))
    return b.value, e.value


def run_ptr(name, ptr, n, fastq, opts_json, out_cap, threads=1):
    """Timing entry of bench.py's cpu_baseline legs: operator `name` ("seq", "grep", "subseq", "translate", "rmdup") on the n
    bytes at address `ptr` -- no copy of the input, the output written into an UNINITIALISED buffer of out_cap bytes (grown
    once if the operator needs more) and discarded.  Returns (output bytes, output records).  The test entries above copy
    the input, zero-fill four times its size for the output and copy the result: fine for parity tests, but with 256
    callers at once the page faults of those buffers, not the restatement, were what got timed."""
    import numpy as np
    fn, mk = {"seq": (_lib.orc_seq, seq_opts), "grep": (_lib.orc_grep, grep_opts), "subseq": (_lib.orc_subseq, subseq_opts),
              "translate": (_lib.orc_translate, translate_opts), "rmdup": (_lib.orc_rmdup, rmdup_opts),
              "rmdup_mt": (_lib.orc_rmdup_mt, rmdup_opts)}[name]
    o = mk(opts_json)
    cap = int(out_cap)
    while True:
        out = np.empty(max(16, cap), dtype=np.uint8)
        nout, nrec, err = C.c_size_t(), C.c_uint64(), C.create_string_buffer(_ERR)
        rc = fn(C.c_void_p(ptr), C.c_size_t(n), int(fastq), C.byref(o), int(threads), C.c_void_p(out.ctypes.data), C.c_size_t(out.size), C.byref(nout),
                C.byref(nrec), err, _ERR)
        if rc == 2:
            cap = nout.value + 16
            continue
        if rc:
            raise OracleError(err.value.decode())
        return nout.value, nrec.value


def _buf(data):
    return (C.c_char * len(data)).from_buffer_copy(data) if len(data) else (C.c_char * 1)()


_ERR = 1024


def count_records(data, fastq):
    out, err = C.c_uint64(), C.create_string_buffer(_ERR)
    if _lib.orc_count_records(_buf(data), C.c_size_t(len(data)), int(fastq), C.byref(out), err, _ERR):
        raise OracleError(err.value.decode())
    return out.value


def is_strict_4line_fastq(data):
    return bool(_lib.orc_is_strict_4line_fastq(_buf(data), C.c_size_t(len(data))))


def record_spans(data, fastq):
    cap = max(16, len(data) // 2 + 2)
    st, ln = (C.c_uint64 * cap)(), (C.c_uint64 * cap)()
    n, err = C.c_size_t(), C.create_string_buffer(_ERR)
    if _lib.orc_record_spans(_buf(data), C.c_size_t(len(data)), int(fastq), st, ln, C.c_size_t(cap), C.byref(n), err, _ERR):
        raise OracleError(err.value.decode())
    return list(zip(st[:n.value], ln[:n.value]))


def stats_map(data, fastq, opts_json="{}", nparts=1):
    o = stats_opts(opts_json)
    cap = 1 << 20
    keys, vals = (C.c_int64 * cap)(), (C.c_int64 * cap)()
    n, err = C.c_size_t(), C.create_string_buffer(_ERR)
    if _lib.orc_stats_map(_buf(data), C.c_size_t(len(data)), int(fastq), C.byref(o), nparts, keys, vals,
                          C.c_size_t(cap), C.byref(n), err, _ERR):
        raise OracleError(err.value.decode())
    return dict(zip(keys[:n.value], vals[:n.value]))


def stats_map_ptr(ptr, n, fastq, opts_json="{}", nparts=1):
    """Same as stats_map but on a raw host pointer (no copy; bench.py's cpu_baseline leg)."""
    o = stats_opts(opts_json)
    cap = 1 << 16
    keys, vals = (C.c_int64 * cap)(), (C.c_int64 * cap)()
    nn, err = C.c_size_t(), C.create_string_buffer(_ERR)
    if _lib.orc_stats_map(C.c_void_p(ptr), C.c_size_t(n), int(fastq), C.byref(o), nparts, keys, vals,
                          C.c_size_t(cap), C.byref(nn), err, _ERR):
        raise OracleError(err.value.decode())
    return dict(zip(keys[:nn.value], vals[:nn.value]))


def stats_string(data, fastq, opts_json="{}", nparts=1, name="input0", fmt="N/A"):
    o = stats_opts(opts_json)
    out, err = C.create_string_buffer(1 << 16), C.create_string_buffer(_ERR)
    if _lib.orc_stats_string(_buf(data), C.c_size_t(len(data)), int(fastq), C.byref(o), nparts, name.encode(),
                             fmt.encode(), out, C.c_size_t(len(out)), err, _ERR):
        raise OracleError(err.value.decode())
    return out.value.decode()


def stats_string_from_map(m, first_record, opts_json="{}", name="input0", fmt="N/A"):
    o = stats_opts(opts_json)
    ks = sorted(m)
    keys, vals = (C.c_int64 * len(ks))(*ks), (C.c_int64 * len(ks))(*[m[k] for k in ks])
    out, err = C.create_string_buffer(1 << 16), C.create_string_buffer(_ERR)
    if _lib.orc_stats_string_from_map(keys, vals, C.c_size_t(len(ks)), _buf(first_record),
                                      C.c_size_t(len(first_record)), C.byref(o), name.encode(), fmt.encode(), out,
                                      C.c_size_t(len(out)), err, _ERR):
        raise OracleError(err.value.decode())
    return out.value.decode()


def wrap(s, width):
    out = C.create_string_buffer(2 * len(s) + 16)
    n = C.c_size_t()
    _lib.orc_wrap(_buf(s), C.c_size_t(len(s)), width, out, C.c_size_t(len(out)), C.byref(n))
    return out.raw[:n.value]


def parse_head(head, regexp=""):
    i, d = C.create_string_buffer(4096), C.create_string_buffer(4096)
    if _lib.orc_parse_head(head.encode(), regexp.encode(), i, C.c_size_t(4096), d, C.c_size_t(4096)):
        raise OracleError("parse_head")
    return i.value.decode(), d.value.decode()


_lib.orc_go_round.restype = C.c_double
_lib.orc_go_round.argtypes = [C.c_double, C.c_int]


def go_round(f, n):
    return _lib.orc_go_round(f, n)


def _records(data, fastq, opts_json, which, nparts):
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    g = lambda k, dv: dv if d.get(k) is None else d[k]
    cfg = _cfg(d)
    num = {0: 0, 1: 0, 2: g("N", 10), 3: g("Times", 1)}[which]
    rng = g("Range", "").encode()
    cap = 4 * len(data) * max(1, num if which == 3 else 1) + 4096
    while True:
        out, n, nrec, err = C.create_string_buffer(cap), C.c_size_t(), C.c_uint64(), C.create_string_buffer(_ERR)
        rc = _lib.orc_records(_buf(data), C.c_size_t(len(data)), int(fastq), C.byref(cfg), which, rng,
                              C.c_longlong(num), nparts, out, C.c_size_t(cap), C.byref(n), C.byref(nrec), err, _ERR)
        if rc == 2:
            cap = n.value + 16
            continue
        if rc:
            raise OracleError(err.value.decode())
        return out.raw[:n.value]


def fq2fa(data, fastq, opts_json="{}", nparts=1):
    return _records(data, fastq, opts_json, 0, nparts)


def range_(data, fastq, opts_json, nparts=1):
    return _records(data, fastq, opts_json, 1, nparts)


def head(data, fastq, opts_json="{}", nparts=1):
    return _records(data, fastq, opts_json, 2, nparts)


def duplicate(data, fastq, opts_json="{}", nparts=1):
    return _records(data, fastq, opts_json, 3, nparts)


def rename(data, fastq, opts_json="{}"):
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    cfg = _cfg(d)
    cap = 2 * len(data) + 4096
    while True:
        out, n, nrec, err = C.create_string_buffer(cap), C.c_size_t(), C.c_uint64(), C.create_string_buffer(_ERR)
        rc = _lib.orc_rename(_buf(data), C.c_size_t(len(data)), int(fastq), C.byref(cfg), int(bool(d.get("ByName"))), out,
                             C.c_size_t(cap), C.byref(n), C.byref(nrec), err, _ERR)
        if rc == 2:
            cap = n.value + 16
            continue
        if rc:
            raise OracleError(err.value.decode())
        return out.raw[:n.value]


class SortOpts(C.Structure):
    _fields_ = [("Config", KitConfig)] + [(k, C.c_int) for k in ("InNaturalOrder", "BySeq", "ByName", "ByLength", "ByBases")] + \
               [("GapLetters", C.c_char_p), ("Reverse", C.c_int), ("IgnoreCase", C.c_int), ("SeqPrefixLength", C.c_longlong)]


def sort(data, fastq, opts_json="{}"):
    """defaults per /root/reference/bigseqkit/sort.go:26-39"""
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    g = lambda k, dv: dv if d.get(k) is None else d[k]
    b = lambda k: int(bool(g(k, False)))
    o = SortOpts(_cfg(d), b("InNaturalOrder"), b("BySeq"), b("ByName"), b("ByLength"), b("ByBases"),
                 g("GapLetters", "- \t.").encode(), b("Reverse"), b("IgnoreCase"), g("SeqPrefixLength", 10000))
    cap = 2 * len(data) + 4096
    while True:
        out, n, nrec, err = C.create_string_buffer(cap), C.c_size_t(), C.c_uint64(), C.create_string_buffer(_ERR)
        rc = _lib.orc_sort(_buf(data), C.c_size_t(len(data)), int(fastq), C.byref(o), out, C.c_size_t(cap), C.byref(n),
                           C.byref(nrec), err, _ERR)
        if rc == 2:
            cap = n.value + 16
            continue
        if rc:
            raise OracleError(err.value.decode())
        return out.raw[:n.value]


def faidx(data, fastq, opts_json="{}", nparts=1):
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    cfg = _cfg(d)
    cap = 2 * len(data) + 4096
    while True:
        out, n, nrec, err = C.create_string_buffer(cap), C.c_size_t(), C.c_uint64(), C.create_string_buffer(_ERR)
        rc = _lib.orc_faidx(_buf(data), C.c_size_t(len(data)), int(fastq), C.byref(cfg), int(bool(d.get("FullHead"))), nparts,
                            out, C.c_size_t(cap), C.byref(n), C.byref(nrec), err, _ERR)
        if rc == 2:
            cap = n.value + 16
            continue
        if rc:
            raise OracleError(err.value.decode())
        return out.raw[:n.value]


def pair(a, b, fastq, opts_json="{}"):
    """-> (paired.1, paired.2, unpaired.1, unpaired.2)"""
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    cfg = _cfg(d)
    res = []
    for which in range(4):
        cap = 2 * (len(a) + len(b)) + 4096
        out, n, nrec, err = C.create_string_buffer(cap), C.c_size_t(), C.c_uint64(), C.create_string_buffer(_ERR)
        rc = _lib.orc_pair(_buf(a), C.c_size_t(len(a)), _buf(b), C.c_size_t(len(b)), int(fastq), C.byref(cfg), which, out,
                           C.c_size_t(cap), C.byref(n), C.byref(nrec), err, _ERR)
        if rc:
            raise OracleError(err.value.decode())
        res.append(out.raw[:n.value])
    return tuple(res)


def common(files, fastq, opts_json="{}"):
    """files: list of bytes (every one ending in a newline) -> records of files[0] common to all"""
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    cfg = _cfg(d)
    data = b"".join(files)
    ends, at = [], 0
    for f in files:
        at += len(f)
        ends.append(at)
    arr = (C.c_uint64 * len(ends))(*ends)
    cap = 2 * len(data) + 4096
    out, n, nrec, err = C.create_string_buffer(cap), C.c_size_t(), C.c_uint64(), C.create_string_buffer(_ERR)
    b = lambda k: int(bool(d.get(k)))
    rc = _lib.orc_common(_buf(data), arr, len(files), int(fastq), C.byref(cfg), b("ByName"), b("BySeq"), b("IgnoreCase"),
                         b("OnlyPositiveStrand"), out, C.c_size_t(cap), C.byref(n), C.byref(nrec), err, _ERR)
    if rc:
        raise OracleError(err.value.decode())
    return out.raw[:n.value]


def concat(a, b, fastq, opts_json="{}"):
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    cfg = _cfg(d)
    cap = 4 * (len(a) + len(b)) + 4096
    while True:
        out, n, nrec, err = C.create_string_buffer(cap), C.c_size_t(), C.c_uint64(), C.create_string_buffer(_ERR)
        rc = _lib.orc_concat(_buf(a), C.c_size_t(len(a)), _buf(b), C.c_size_t(len(b)), int(fastq), C.byref(cfg),
                             int(bool(d.get("Full"))), out, C.c_size_t(cap), C.byref(n), C.byref(nrec), err, _ERR)
        if rc == 2:
            cap = n.value + 16
            continue
        if rc:
            raise OracleError(err.value.decode())
        return out.raw[:n.value]


def faidx_query(data, fastq, opts_json):
    d = json.loads(opts_json) if isinstance(opts_json, (str, bytes)) else dict(opts_json or {})
    cfg = _cfg(d)
    qs = []
    if d.get("RegionFile"):
        qs += [l.rstrip("\r") for l in open(d["RegionFile"]).read().split("\n") if l.strip("\r")]
    qs += list(d.get("Regions") or [])
    cap = 2 * len(data) + 4096
    out, n, nrec, err = C.create_string_buffer(cap), C.c_size_t(), C.c_uint64(), C.create_string_buffer(_ERR)
    rc = _lib.orc_faidx_query(_buf(data), C.c_size_t(len(data)), int(fastq), C.byref(cfg), "\n".join(qs).encode(),
                              int(bool(d.get("IgnoreCase"))), int(bool(d.get("UseRegexp"))), out, C.c_size_t(cap), C.byref(n),
                              C.byref(nrec), err, _ERR)
    if rc:
        raise OracleError(err.value.decode())
    return out.raw[:n.value]
