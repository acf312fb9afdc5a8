"""`python -m bigseqkit_amd.run --devices ...` / `bigseqkit <cmd> ... --devices ...`: the user-facing entry point for several
GPUs (one worker per device, the file cut on record starts in 1 MiB windows, collectives where the command reduces or
exchanges -- /root/reference/bigseqkit/helper.go:148-195, bigseqkit-cli/helper.go:87-141).  Two workers share the one GPU of
the test box (gloo): their files must equal what the single-device command line writes."""
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bigseqkit_amd", "bin", "bigseqkit")


def fastq(nrec, seed):
    rng = random.Random(seed)
    seqs, out = [], []
    for i in range(nrec):
        if i > 20 and rng.random() < 0.25:
            s = seqs[rng.randrange(len(seqs))]
        else:
            s = "".join(rng.choice("ACGT") for _ in range(rng.randint(40, 180)))
            if rng.random() < 0.05:
                k = rng.randrange(len(s) - 12)
                s = s[:k] + "ACGTTGCAAGCT" + s[k + 12:]
        seqs.append(s)
        q = "".join(chr(rng.randint(35, 73)) for _ in s)
        if i % 97 == 5:
            q = "@" + q[1:]                       # a quality line that begins like a header
        out.append("@read%d some description\n%s\n+\n%s\n" % (i, s, q))
    return "".join(out).encode()


def fasta(nrec, seed):
    rng = random.Random(seed)
    out = []
    for i in range(nrec):
        s = "".join(rng.choice("ACGT") for _ in range(rng.randint(200, 2000)))
        out.append(">cds%d len=%d\n" % (i, len(s)) + "".join(s[j:j + 60] + "\n" for j in range(0, len(s), 60)))
    return "".join(out).encode()


def read_out(path):
    if os.path.isdir(path):
        return b"".join(open(os.path.join(path, f), "rb").read() for f in sorted(os.listdir(path)))
    return open(path, "rb").read()


def run(cmd, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run(cmd, capture_output=True, cwd=ROOT, env=env, timeout=600)
    assert p.returncode == 0, (cmd, p.stderr.decode()[-3000:])
    return p.stdout


CASES = [("seq", ["seq", "-r", "-p", "-t", "dna", "--quiet"], "fq"),
         ("grep", ["grep", "-s", "-p", "ACGTTGCAAGCT"], "fq"),
         ("subseq", ["subseq", "-r", "5:-5"], "fq"),
         ("rmdup", ["rmdup", "-s"], "fq"),
         ("locate", ["locate", "-p", "ACGTTGCA"], "fq"),
         ("translate", ["translate", "-f", "6", "-x"], "fa"),
         ("seqfa", ["seq", "-w", "70"], "fa")]


@pytest.mark.parametrize("merge", [False, True])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_two_workers_write_what_one_device_writes(case, merge, tmp_path):
    name, args, kind = case
    data = fastq(30000, 11) if kind == "fq" else fasta(3000, 12)
    src = str(tmp_path / ("in." + kind))
    open(src, "wb").write(data)
    one, two = str(tmp_path / "one.out"), str(tmp_path / "two.out")
    extra = ["--merge"] if merge else []
    run([CLI] + args + [src, "-o", one] + extra)
    run([sys.executable, "-m", "bigseqkit_amd.run", "--devices", "0,0", "--share-gpu", "--"] + args + [src, "-o", two] + extra)
    want, got = read_out(one), read_out(two)
    assert len(want) > 0 and got == want
    if not merge:
        assert sorted(os.listdir(two)) == ["part00000", "part00001"]
    else:
        assert os.path.isfile(two) and not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]


def wrapped_fastq(nrec, seed, width=60):
    """the same kind of reads with bases and qualities wrapped at `width` (SeqParser reads them, helper.go:252-269); first
    quality characters '@' / '+' and continuation lines that begin with '+' included"""
    rng = random.Random(seed)
    seqs, out = [], []
    for i in range(nrec):
        if i > 20 and rng.random() < 0.25:
            s = seqs[rng.randrange(len(seqs))]
        else:
            s = "".join(rng.choice("ACGT") for _ in range(rng.randint(40, 400)))
            if rng.random() < 0.05:
                k = rng.randrange(len(s) - 12)
                s = s[:k] + "ACGTTGCAAGCT" + s[k + 12:]
        seqs.append(s)
        q = [chr(rng.randint(35, 73)) for _ in s]
        for k in range(width, len(s), width):
            q[k] = "+" if rng.random() < 0.2 else ("A" if q[k] == "@" else q[k])
        if i % 37 == 5:
            q[0] = "@"
        q = "".join(q)
        out.append("@read%d some description\n%s\n+\n%s\n" % (i, "\n".join(s[j:j + width] for j in range(0, len(s), width)),
                                                              "\n".join(q[j:j + width] for j in range(0, len(q), width))))
    return "".join(out).encode()


@pytest.mark.parametrize("case", [c for c in CASES if c[2] == "fq"], ids=[c[0] for c in CASES if c[2] == "fq"])
def test_two_workers_on_a_wrapped_fastq_file(case, tmp_path):
    """Round 4: a FASTQ file whose records are wrapped over several lines is cut on record starts too (bsk_find_record_start
    reads the wrapped grammar), every worker rewrites its shard to four lines per record on the device as a single shard
    always did, and the rmdup exchange takes such shards (bsk_rmdup_dist_keys used to refuse them)."""
    name, args, kind = case
    data = wrapped_fastq(12000, 21)
    src = str(tmp_path / "in.fq")
    open(src, "wb").write(data)
    one, two = str(tmp_path / "one.out"), str(tmp_path / "two.out")
    run([CLI] + args + [src, "-o", one, "--merge"])
    run([sys.executable, "-m", "bigseqkit_amd.run", "--devices", "0,0,0", "--share-gpu", "--"] + args + [src, "-o", two, "--merge"])
    want, got = read_out(one), read_out(two)
    assert len(want) > 0 and got == want
    if name == "stats":
        return
    for sargs in (["stats", "-a", "-T"],):
        assert run([sys.executable, "-m", "bigseqkit_amd.run", "--devices", "0,0", "--share-gpu", "--"] + sargs + [src]) == run([CLI] + sargs + [src])


@pytest.mark.parametrize("args", [["rmdup", "-s"], ["stats", "-a", "-T"], ["grep", "-s", "-p", "ACGTTGCAAGCT", "-C"], ["seq", "-n", "-i"]],
                         ids=["rmdup", "stats", "grep-count", "seq"])
def test_one_worker_over_rccl(args, tmp_path):
    """BSK_DIST_SINGLE_RANK_COLLECTIVES=1: one worker, backend "nccl" -- the process group, the barrier, the all-reduces of
    stats / grep -C, the all_gather + all_to_all exchange of rmdup and the size scan of --merge run over RCCL on the one GPU
    of the box (round 4; with more ranks than GPUs they run over gloo)."""
    data = fastq(20000, 31)
    src = str(tmp_path / "in.fq")
    open(src, "wb").write(data)
    env = {"BSK_DIST_SINGLE_RANK_COLLECTIVES": "1", "BSK_A2A_MAX_BYTES": "100000"}
    if args[0] in ("stats",) or args[-1] == "-C":
        want = run([CLI] + args + [src])
        got = run([sys.executable, "-m", "bigseqkit_amd.run", "--devices", "0", "--"] + args + [src], env)
        assert got == want and len(want) > 0
        return
    one, two = str(tmp_path / "one.out"), str(tmp_path / "two.out")
    run([CLI] + args + [src, "-o", one, "--merge"])
    run([sys.executable, "-m", "bigseqkit_amd.run", "--devices", "0", "--"] + args + [src, "-o", two, "--merge"], env)
    assert len(read_out(one)) > 0 and read_out(two) == read_out(one)


def test_stats_and_grep_count_reduce_over_the_workers(tmp_path):
    data = fastq(30000, 13)
    src = str(tmp_path / "in.fq")
    open(src, "wb").write(data)
    for args in (["stats", "-a", "-T"], ["stats"], ["grep", "-s", "-p", "ACGTTGCAAGCT", "-C"]):
        want = run([CLI] + args + [src])
        got = run([sys.executable, "-m", "bigseqkit_amd.run", "--devices", "0,0,0", "--share-gpu", "--"] + args + [src])
        assert got == want and len(want) > 0, (args, got, want)


def test_the_command_line_hands_over_with_devices(tmp_path):
    """`bigseqkit <cmd> ... --devices 0,0`: the C++ command line starts the workers itself"""
    data = fastq(12000, 14)
    src = str(tmp_path / "in.fq")
    open(src, "wb").write(data)
    one, two = str(tmp_path / "one.out"), str(tmp_path / "two.out")
    run([CLI, "rmdup", "-s", src, "-o", one, "--merge"])
    run([CLI, "rmdup", "-s", src, "-o", two, "--merge", "--devices", "0,0"], {"BSK_RUN_SHARE_GPU": "1"})
    assert read_out(two) == read_out(one)
