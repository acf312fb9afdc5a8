"""Host logic of the multi-GPU entry point (bigseqkit_amd/run.py): shard bounds from windows of the mapped file, the device
list, and the plan the C++ command line hands to the launcher (no second flag parser)."""
import json
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bigseqkit_amd", "bin", "bigseqkit")
sys.path.insert(0, ROOT)


def fastq(nrec, seed):
    rng = random.Random(seed)
    out = []
    for i in range(nrec):
        s = "".join(rng.choice("ACGT") for _ in range(rng.randint(40, 180)))
        q = "".join(chr(rng.randint(35, 73)) for _ in s)
        if i % 97 == 5:
            q = "@" + q[1:]                       # a quality line that begins like a header
        out.append("@read%d some description\n%s\n+\n%s\n" % (i, s, q))
    return "".join(out).encode()


def test_cuts_fall_on_record_starts_without_reading_the_file(tmp_path):
    """the shard bounds of N workers: record starts found in windows of the mapped file; the pieces tile the file"""
    import mmap
    from bigseqkit_amd import run as brun
    from bigseqkit_amd._lib import lib, check
    import oracle
    data = fastq(20000, 15)
    src = str(tmp_path / "in.fq")
    open(src, "wb").write(data)
    starts = {a for a, _ in oracle.record_spans(data, True)} | {len(data)}
    with open(src, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        for world in (1, 2, 3, 7, 16):
            cuts = brun.cut_points(mm, len(data), world, 1, lib, check)
            assert cuts[0] == 0 and cuts[-1] == len(data) and len(cuts) == world + 1
            assert all(a <= b for a, b in zip(cuts, cuts[1:])) and all(c in starts for c in cuts)
        mm.close()


def test_fasta_cuts_and_long_lines(tmp_path):
    """FASTA: '>' at a line start; a window that holds no record start (a chromosome line of 3 MB) grows"""
    import mmap
    from bigseqkit_amd import run as brun
    from bigseqkit_amd._lib import lib, check
    rng = random.Random(4)
    recs = [">c%d\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(3_000_000 if i == 2 else 500))) for i in range(6)]
    data = "".join(recs).encode()
    src = str(tmp_path / "in.fa")
    open(src, "wb").write(data)
    starts, at = {len(data)}, 0
    for r in recs:
        starts.add(at)
        at += len(r)
    with open(src, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        for world in (2, 3, 5):
            cuts = brun.cut_points(mm, len(data), world, 0, lib, check)
            assert len(cuts) == world + 1 and all(c in starts for c in cuts) and all(a <= b for a, b in zip(cuts, cuts[1:]))
        mm.close()


def test_device_lists():
    from bigseqkit_amd import run as brun
    assert brun.parse_devices("0,1,2") == [0, 1, 2]
    assert brun.parse_devices("0-3") == [0, 1, 2, 3]
    assert brun.parse_devices("0-1, 4,6-7") == [0, 1, 4, 6, 7]
    assert brun.parse_devices("0,0") == [0, 0]


def test_the_command_line_plans_for_the_launcher(tmp_path):
    src = str(tmp_path / "x.fq")
    open(src, "wb").write(b"@r\nACGT\n+\nIIII\n")
    p = subprocess.run([CLI, "grep", "-s", "-p", "ACG", "-i", src, "-o", str(tmp_path / "o"), "--merge", "--plan"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    d = json.loads(p.stdout)
    assert d["use"] == "grep" and d["op"] == "Grep" and d["files"] == [src] and d["merge"] is True
    assert d["opts"]["BySeq"] is True and d["opts"]["IgnoreCase"] is True and d["opts"]["Pattern"] == ["ACG"]
    assert d["out_file"] == str(tmp_path / "o")
    # a flag error is reported by the same parser, before any worker starts
    p = subprocess.run([CLI, "grep", "--no-such-flag", src, "--plan"], capture_output=True, text=True)
    assert p.returncode != 0
