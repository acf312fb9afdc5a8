"""The `bigseqkit` command line (cli/bigseqkit.cpp): cobra flag tables -> option JSON (CPU, --dry-run) and
end-to-end runs on files compared with the CPU oracle driven by the SAME option JSON (GPU).
Flag tables: /root/reference/bigseqkit-cli/{helper.go:161-173,seq.go:54-73,stats.go:61-65,grep.go:81-96,
locate.go:61-74,subseq.go:56-67,translate.go:86-94,rmdup.go:45-51}."""
import json
import os
import random
import subprocess
import zlib

import pytest

import oracle
import seqgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bigseqkit_amd", "bin", "bigseqkit")


def run(*args, ok=True):
    p = subprocess.run([CLI, *args], capture_output=True, timeout=600)
    if ok:
        assert p.returncode == 0, p.stderr.decode()
    return p


def dry(*args):
    out = run(*args, "--dry-run").stdout.decode().split("\n")
    return out[0], json.loads(out[1]), [f for f in out[2:] if f]


def test_persistent_flag_defaults_match_reference_cli():
    op, js, files = dry("seq", "a.fa", "b.fa")
    assert op == "SeqTransform" and files == ["a.fa", "b.fa"]
    assert js["Config"] == {"SeqType": "auto", "LineWidth": 60, "IDRegexp": "^(\\S+)\\s?", "IDNCBI": False,
                            "Quiet": False, "AlphabetGuessSeqLength": 10000}
    assert js["GapLetters"] == "- \t." and js["ValidateSeqLength"] == 10000 and js["QualAsciiBase"] == 33
    assert js["MinLen"] == -1 and js["MaxQual"] == -1


def test_shorthand_clusters_values_and_slices():
    op, js, files = dry("grep", "-sic", "-p", "ACGT,TTT", "-p", '"A,C"', "-R", "1:30", "-m1", "--max-mismatch=2", "x.fq")
    assert op == "Grep" and files == ["x.fq"]
    assert js["Pattern"] == ["ACGT", "TTT", "A,C"]
    assert js["BySeq"] and js["IgnoreCase"] and js["Circular"] and not js["ByName"]
    assert js["Region"] == "1:30" and js["MaxMismatch"] == 2
    op, js, _ = dry("translate", "-f", "1,2,-1", "-T11", "-xMF", "--trim")
    assert op == "Translate" and js["Frame"] == ["1", "2", "-1"] and js["TranslTable"] == 11
    assert js["AllowUnknownCodon"] and js["InitCodonAsM"] and js["AppendFrame"] and js["Trim"] and not js["Clean"]
    assert dry("translate")[1]["Frame"] == ["1"]
    op, js, _ = dry("subseq", "-r", "2:-3", "-u", "5", "-f")
    assert op == "SubseqTransform" and js["Region"] == "2:-3" and js["UpStream"] == 5 and js["OnlyFlank"]
    assert js["GtfTag"] == "gene_id" and js["Chr"] == []
    op, js, _ = dry("stats", "-aT", "-G", "-.", "-E", "illumina-1.3+")
    assert op == "Stats" and js["All"] and js["Tabular"] and js["GapLetters"] == "-." and js["FqEncoding"] == "illumina-1.3+"
    op, js, _ = dry("rmdup", "-s", "-i", "-P")
    assert op == "RmDup" and js["BySeq"] and js["IgnoreCase"] and js["OnlyPositiveStrand"] and not js["ByName"]
    assert dry("fq2fa", "x.fq")[0] == "Fq2Fa" and dry("range", "-r", "1:12")[1]["Range"] == "1:12"
    assert dry("head")[0:2] == ("Head", dict(dry("head")[1], N=10)) and dry("head", "-n", "3")[1]["N"] == 3
    assert dry("dup", "-n", "4")[0] == "Duplicate" and dry("duplicate", "--times", "4")[1]["Times"] == 4
    op, js, _ = dry("locate", "-p", "AA", "-GMP", "-w", "0")
    assert op == "Locate" and js["NonGreedy"] and js["HideMatched"] and js["OnlyPositiveStrand"] and js["Config"]["LineWidth"] == 0


def test_id_ncbi_replaces_the_id_regexp():
    assert dry("seq", "--id-ncbi")[1]["Config"]["IDRegexp"] == r"\|([^\|]+)\| "


def test_flag_value_checks_use_the_reference_messages(tmp_path):
    for args, msg in [(("seq", "-w", "-1"), "value of flag --line-width should be greater than 0"),
                      (("seq", "-V", "10"), "value of flag --validate-seq-length too small, should >= 1000"),
                      (("seq", "-b", "0"), "value of flag --qual-ascii-base should be greater than 0"),
                      (("translate", "-T", "0"), "value of flag --transl-table should be greater than 0"),
                      (("grep", "-m", "-2"), "value of flag --max-mismatch should be greater than 0"),
                      (("seq", "--alphabet-guess-seq-length", "5"), "value of flag --alphabet-guess-seq-length too small, should >= 1000"),
                      (("seq", "--nope"), "unknown flag: --nope"),
                      (("seq", "-w", "x"), 'invalid argument "x" for "--line-width" flag'),
                      (("frobnicate",), 'unknown command "frobnicate" for "bigseqkit"')]:
        p = run(*args, "--dry-run", ok=False)
        assert p.returncode == 1 and msg in p.stderr.decode(), (args, p.stderr)


def test_infile_list(tmp_path):
    lst = tmp_path / "files.txt"
    lst.write_text("c.fa\nd.fq\n")
    assert dry("seq", "a.fa", "--infile-list", str(lst))[2] == ["a.fa", "c.fa", "d.fq"]


# ------------------------------------------------------------------ GPU: files in, files out
def _write(tmp_path, name, data):
    p = tmp_path / name
    p.write_bytes(data)
    return str(p)


CASES = [
    ("seq", ["-rp", "-w", "50"], "fa"), ("seq", ["-n", "-i"], "fq"), ("seq", ["-m", "100", "-M", "400", "-u"], "fa"),
    ("seq", ["-Q", "20"], "fq"),
    ("grep", ["-s", "-p", "ACGTAC,GGATCC", "-i"], "fa"), ("grep", ["-n", "-p", "r3 x", "-v"], "fq"),
    ("subseq", ["-r", "3:-4"], "fa"), ("subseq", ["-r", "-20:-1"], "fq"),
    ("locate", ["-p", "ACG,TTGA", "-i"], "fa"), ("locate", ["-p", "GATC", "-P", "-M"], "fq"),
    ("translate", ["-f", "1,-2", "-x"], "fa"), ("translate", ["-T", "11", "-F", "--trim", "-x"], "fa"),
    ("rmdup", ["-s"], "fa"), ("rmdup", ["-n"], "fq"),
    ("grep", ["-r", "-p", "^r1\\d$", "-i"], "fq"), ("grep", ["-s", "-d", "-p", "ACGNNT"], "fa"),
    ("grep", ["-s", "-m", "1", "-p", "ACGTACGTAC"], "fa"), ("locate", ["-d", "-p", "ACNNT", "--bed"], "fa"),
    ("locate", ["-m", "1", "-p", "ACGTAC", "-i"], "fq"), ("locate", ["-F", "-p", "ACG"], "fa"),
    ("fq2fa", [], "fq"), ("fq2fa", [], "fa"), ("range", ["-r", "5:40"], "fq"), ("range", ["-r", "-30:-1"], "fa"),
    ("head", [], "fa"), ("head", ["-n", "77"], "fq"), ("duplicate", ["-n", "3"], "fq"), ("dup", [], "fa"),
    ("rename", [], "fq"), ("rename", ["-n"], "fa"), ("sort", ["-l", "-r"], "fa"), ("sort", ["-s", "-i"], "fq"), ("sort", [], "fa"),
    ("faidx", [], "fa"), ("faidx", ["-f"], "fq"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("cmd,flags,kind", CASES)
def test_cli_output_equals_oracle(tmp_path, cmd, flags, kind):
    rng = random.Random(zlib.crc32(repr((cmd, flags)).encode()))
    fastq = kind == "fq"
    data = seqgen.random_fastq(rng, 300, min_len=1) if fastq else seqgen.random_fasta(rng, 300, min_len=1)
    if cmd in ("rmdup", "rename"):
        data = data + data[:len(data)]  # every record twice
    path = _write(tmp_path, "in." + kind, data)
    op, js, _ = dry(cmd, *flags, path)
    fn = {"seq": oracle.seq, "grep": oracle.grep, "subseq": oracle.subseq, "locate": oracle.locate,
          "translate": oracle.translate, "rmdup": oracle.rmdup, "fq2fa": oracle.fq2fa, "range": oracle.range_,
          "head": oracle.head, "duplicate": oracle.duplicate, "dup": oracle.duplicate, "rename": oracle.rename, "sort": oracle.sort, "faidx": oracle.faidx}[cmd]
    want = fn(data, fastq, json.dumps(js))
    want = want[0] if isinstance(want, tuple) else want
    assert run(cmd, *flags, path, "-o", "-").stdout == want
    # default store: <file>-out directory of parts; --merge: one file
    run(cmd, *flags, path, "--partitions", "3")
    parts = sorted(os.listdir(path + "-out"))
    assert parts == ["part00000", "part00001", "part00002"]
    assert b"".join(open(os.path.join(path + "-out", p), "rb").read() for p in parts) == want
    merged = str(tmp_path / "merged.out")
    run(cmd, *flags, path, "--merge", "-o", merged)
    assert open(merged, "rb").read() == want


@pytest.mark.gpu
def test_cli_stats_two_inputs_and_grep_count(tmp_path):
    rng = random.Random(5)
    fq = seqgen.random_fastq(rng, 500, min_len=1)
    fa = seqgen.random_fasta(rng, 200, min_len=1)
    a, b = _write(tmp_path, "a.fastq", fq), _write(tmp_path, "b.fna", fa)
    for flags in (["-a"], ["-T"], ["-a", "-T"], []):
        _, js, _ = dry("stats", *flags)
        t0 = oracle.stats_string(fq, True, json.dumps(js), name="input0")
        t1 = oracle.stats_string(fa, False, json.dumps(js), name="input1")
        head = t1.split("\n")[0] + "\n"
        body = "\n".join(t0.split("\n")[1:]) + "\n" + "\n".join(t1.split("\n")[1:]) + "\n"
        assert run("stats", *flags, a, b).stdout.decode() == head + body
    _, js, _ = dry("grep", "-s", "-p", "ACG", "-C")
    js["Count"] = True
    want = int(oracle.grep(fa, False, json.dumps(js)).strip() or 0)
    assert run("grep", "-s", "-p", "ACG", "-C", b).stdout == str(want).encode()


@pytest.mark.gpu
def test_cli_sniffs_format_from_first_byte_and_reports_kernel_errors(tmp_path):
    rng = random.Random(9)
    fa = seqgen.random_fasta(rng, 50, min_len=1)
    p = _write(tmp_path, "noext", fa)
    assert run("seq", p, "-o", "-").stdout == oracle.seq(fa, False, json.dumps(dry("seq")[1]))
    bad = _write(tmp_path, "bad.txt", b"hello\n")
    r = run("seq", bad, ok=False)
    assert r.returncode == 1 and b"must be fasta or fastq" in r.stderr
    r = run("grep", p, ok=False)
    assert r.returncode == 1 and b"one of flags -p (--pattern) and -f (--pattern-file) needed" in r.stderr


@pytest.mark.gpu
def test_cli_pipe_chains_commands_in_hbm(tmp_path):
    """`pipe --job job.json` (bigseqkit-cli/pipe.go): outputs of the jobs under "pipe" feed "cmd" without leaving HBM.
    seq (length filter) -> grep -s -> rmdup -s, plus a second branch, checked against the oracle applied step by step."""
    rng = random.Random(31)
    recs = []
    for i in range(600):
        s = "".join(rng.choice("ACGT") for _ in range(rng.randint(20, 200)))
        if i % 4 == 0 and i > 8:
            s = recs[rng.randrange(len(recs))][1]
        recs.append((f"r{i} d", s))
    fq1 = "".join(f"@{n}\n{s}\n+\n{'I' * len(s)}\n" for n, s in recs[:350]).encode()
    fq2 = "".join(f"@{n}\n{s}\n+\n{'I' * len(s)}\n" for n, s in recs[350:]).encode()
    a, b = _write(tmp_path, "a.fq", fq1), _write(tmp_path, "b.fq", fq2)
    job = {"pipe": [{"pipe": [{"cmd": ["seq", "-m", "60", a]}], "cmd": ["grep", "-s", "-p", "ACG"]},
                    {"cmd": ["seq", "-M", "120", b]}],
           "cmd": ["rmdup", "-s"]}
    jf = tmp_path / "job.json"
    jf.write_text(json.dumps(job))
    got = run("pipe", "--job", str(jf), "-o", "-").stdout
    o_seq = json.dumps(dry("seq", "-m", "60")[1])
    o_grep = json.dumps(dry("grep", "-s", "-p", "ACG")[1])
    o_seq2 = json.dumps(dry("seq", "-M", "120")[1])
    o_rm = json.dumps(dry("rmdup", "-s")[1])
    step1 = oracle.grep(oracle.seq(fq1, True, o_seq), True, o_grep)
    step2 = oracle.seq(fq2, True, o_seq2)
    want = oracle.rmdup(step1 + step2, True, o_rm)
    assert got == want and 0 < len(want) < len(fq1) + len(fq2)
    # a stats table at the end of a chain
    job2 = {"pipe": [{"cmd": ["seq", "-m", "60", a]}], "cmd": ["stats", "-T"]}
    jf.write_text(json.dumps(job2))
    t = oracle.stats_string(oracle.seq(fq1, True, o_seq), True, json.dumps(dry("stats", "-T")[1]), name="input0")
    assert run("pipe", "--job", str(jf)).stdout.decode() == t.split("\n")[0] + "\n" + "\n".join(t.split("\n")[1:]) + "\n"


@pytest.mark.gpu
def test_cli_global_commands_join_files_that_lack_a_final_newline(tmp_path):
    a = _write(tmp_path, "a.fa", b">x 1\nACGT\n>y\nGG")          # no newline at the end
    b = _write(tmp_path, "b.fa", b">x 2\nTTTT\n>z\nC\n")
    union = b">x 1\nACGT\n>y\nGG\n>x 2\nTTTT\n>z\nC\n"
    assert run("rename", a, b, "-o", "-").stdout == oracle.rename(union, False) == b">x 1\nACGT\n>y\nGG\n>x_1 2\nTTTT\n>z\nC\n"
    assert run("sort", "-l", a, b, "-o", "-").stdout == oracle.sort(union, False, '{"ByLength": true}')
    assert run("rmdup", a, b, "-o", "-").stdout == oracle.rmdup(union, False)


@pytest.mark.gpu
def test_cli_faidx_rows_and_region_queries(tmp_path):
    fa = b">chr1 x\nACGTACGTAC\nGGGGGTTTTT\n>chr2\nAAAACCCC\n"
    path = _write(tmp_path, "g.fa", fa)
    assert run("faidx", path, "-o", "-").stdout == oracle.faidx(fa, False) == b"chr1\t20\t8\t10\t11\nchr2\t8\t36\t8\t9\n"
    got = run("faidx", path, "chr1:2-5", "chr2:-3", "chr1:9-1", "-o", "-").stdout
    assert got == b">chr1:2-5\nCGTA\n>chr2:1-3\nAAA\n" == oracle.faidx_query(fa, False, '{"Regions": ["chr1:2-5", "chr2:-3", "chr1:9-1"]}')
    rf = _write(tmp_path, "r.txt", b"chr2:5-\nCHR1:12-15\n")
    assert run("faidx", "-i", "-l", rf, path, "-w", "2", "-o", "-").stdout == b">chr1:12-15\nGG\nGG\n>chr2\nCC\nCC\n"


@pytest.mark.gpu
def test_cli_range_indexes_the_union_of_its_inputs_and_chains_in_a_pipe(tmp_path):
    rng = random.Random(8)
    d1, d2 = seqgen.random_fastq(rng, 120, min_len=1), seqgen.random_fastq(rng, 90, min_len=1)
    a, b = _write(tmp_path, "a.fq", d1), _write(tmp_path, "b.fq", d2)
    for r in ("100:150", "-100:-1", "1:5", "130"):
        want = oracle.range_(d1 + d2, True, json.dumps({"Range": r}))
        assert run("range", "-r", r, a, b, "-o", "-").stdout == want, r
    assert run("head", "-n", "125", a, b, "-o", "-").stdout == oracle.head(d1 + d2, True, '{"N": 125}')
    p = run("range", "-r", "9:3", a, ok=False)
    assert p.returncode != 0 and b"start must be > than end" in p.stderr
    # fq2fa -> duplicate -> head without leaving HBM
    job = {"pipe": [{"pipe": [{"cmd": ["fq2fa", a]}], "cmd": ["dup", "-n", "2"]}], "cmd": ["head", "-n", "7"]}
    jf = tmp_path / "job.json"
    jf.write_text(json.dumps(job))
    want = oracle.head(oracle.duplicate(oracle.fq2fa(d1, True), False, '{"Times": 2}'), False, '{"N": 7}')
    assert run("pipe", "--job", str(jf), "-o", "-").stdout == want and want.count(b">") == 7


@pytest.mark.gpu
def test_cli_rmdup_is_global_over_several_input_files_and_parts_never_split_records(tmp_path):
    rng = random.Random(32)
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(30, 90))) for _ in range(200)]
    f1 = "".join(f"@a{i}\n{s}\n+\n{'@' * len(s)}\n" for i, s in enumerate(seqs)).encode()
    f2 = "".join(f"@b{i}\n{s}\n+\n{'@' * len(s)}\n" for i, s in enumerate(seqs[::2])).encode()   # all duplicates of file 1
    a, b = _write(tmp_path, "a.fq", f1), _write(tmp_path, "b.fq", f2)
    out = tmp_path / "o"
    run("rmdup", "-s", a, b, "-o", str(out), "--partitions", "4")
    parts = sorted(os.listdir(out))
    assert len(parts) == 4
    texts = [open(os.path.join(out, p), "rb").read() for p in parts]
    assert b"".join(texts) == f1                     # every record of b.fq is a duplicate of one in a.fq
    for t in texts:                                  # quality lines start with '@': a line-based cut would split records
        assert t == b"" or (t.startswith(b"@a") and oracle.is_strict_4line_fastq(t))
