#!/usr/bin/env python3
"""End-to-end soak of the command line on a file of a few GB (synthetic FASTQ-150 written to /tmp): every command reads the
file from disk, runs on the GPU and stores its result; sizes and counts are checked against what the generator implies."""
import ctypes as C, hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bigseqkit_amd._lib import lib, check
import bigseqkit_amd._lib as _lib

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
CLI = os.path.join(ROOT, "bigseqkit_amd", "bin", "bigseqkit")
rb = lib.bsk_synth_record_bytes(0)
nrec = int(gb * 1e9) // rb
buf = C.create_string_buffer(nrec * rb)
check(lib.bsk_synth_host(0, 42, _lib.SYNTH_FLAG_DUPS, 0, buf, nrec * rb))
path = "/tmp/soak.fq"
with open(path, "wb") as f:
    f.write(buf.raw)
del buf
print("file", path, nrec, "records", nrec * rb, "bytes")


def run(*args):
    t0 = time.time()
    p = subprocess.run([CLI, *args], capture_output=True)
    assert p.returncode == 0, (args, p.stderr.decode()[-500:])
    return p.stdout, time.time() - t0


out, dt = run("stats", "-T", path)
row = out.decode().splitlines()[1].split("\t")
assert row[3] == str(nrec) and row[4] == str(nrec * 150), row
print("stats ok %.1fs" % dt, row[3:8])
out, dt = run("seq", "-n", "-i", path, "-o", "-")
assert len(out) == 12 * nrec and out[:12] == b"S0000000000\n"
print("seq -n ok %.1fs" % dt)
out, dt = run("grep", "-s", "-p", "ACGTTGCAAGCT", "-C", path)
print("grep -C ok %.1fs" % dt, out.decode())
out, dt = run("rmdup", "-s", path, "-o", "/tmp/soak.rmdup.fq", "--merge")
kept = os.path.getsize("/tmp/soak.rmdup.fq") // rb
assert kept == nrec - nrec // 5, (kept, nrec)
print("rmdup ok %.1fs" % dt, kept)
out, dt = run("sort", "-l", "-r", path, "-o", "/tmp/soak.sorted.fq", "--merge")
assert os.path.getsize("/tmp/soak.sorted.fq") == nrec * rb
print("sort -l ok %.1fs" % dt)
out, dt = run("head", "-n", "1000", path, "-o", "-")
assert len(out) == 1000 * rb
out, dt = run("range", "-r", "-1000:-1", path, "-o", "-")
assert len(out) == 1000 * rb and out.startswith(b"@S%010d" % (nrec - 1000))
print("head / range ok")
os.makedirs("/tmp/soak.pair", exist_ok=True)
out, dt = run("pair", "-O", "/tmp/soak.pair", path, "/tmp/soak.rmdup.fq")
assert os.path.getsize("/tmp/soak.pair/paired.1") == kept * rb == os.path.getsize("/tmp/soak.pair/paired.2")
print("pair ok %.1fs" % dt)
out, dt = run("faidx", path, "-o", "-")
assert out.count(b"\n") == nrec and out.startswith(b"S0000000000\t150\t13\t150\t151\t166\n")
print("faidx ok %.1fs" % dt)
for f in (path, "/tmp/soak.rmdup.fq", "/tmp/soak.sorted.fq", "/tmp/soak.pair/paired.1", "/tmp/soak.pair/paired.2"):
    os.remove(f)
print("cli soak ok")
