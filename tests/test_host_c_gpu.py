"""tests/host_c/ranks.c on the GPU box: the Go shim's call sequence (go/bigseqkit/bsk_cgo.go StatsN / GrepCountN / RmDupN --
bsk_comm_init_all, one OS thread per rank, `agree` before every collective that carries data) compiled as plain C11 against
include/bsk.h and run with N pthreads.  The Go file itself cannot be compiled here (no toolchain); this is its twin, and the
one other consumer of that exact sequence.  Ranks that share the GPU take the "local" backend; one rank
with BSK_COMM=rccl goes through librccl."""
import json
import os
import subprocess

import pytest

import oracle
from test_rmdup_xcheck_gpu import dup_fastq

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bigseqkit_amd", "bin", "host_c_ranks")


def run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    env.update(env_extra or {})
    return subprocess.run([BIN] + args, capture_output=True, cwd=ROOT, env=env, timeout=timeout)


@pytest.mark.parametrize("devices,env", [("0,0,0", {}), ("0,0", {"BSK_A2A_MAX_BYTES": "4096"}), ("0", {"BSK_COMM": "rccl"})])
def test_the_go_call_sequence_from_pthreads(devices, env, tmp_path):
    data = b"@planted\nTTACGTTGCATT\n+\nIIIIIIIIIIII\n" + dup_fastq(71, 12000).replace(b"ACGTTGCA", b"ACGTTGCT")
    src = tmp_path / "in.fq"
    src.write_bytes(data)
    p = run([str(src), devices, str(tmp_path)], env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    world = devices.count(",") + 1
    assert (tmp_path / "stats.txt").read_text() == oracle.stats_string(data, True, json.dumps({"All": True, "Tabular": True}))
    want_c = oracle.grep(data, True, json.dumps({"BySeq": True, "Pattern": ["ACGTTGCA"]})).count(b"\n") // 4
    assert int((tmp_path / "grepc.txt").read_text()) == want_c >= 1
    want = oracle.rmdup(data, True, json.dumps({"BySeq": True}))
    assert b"".join((tmp_path / ("rmdup.%d" % r)).read_bytes() for r in range(world)) == want
    pairs = [[int(x) for x in (tmp_path / ("pairs.%d" % r)).read_text().split()] for r in range(world)]
    assert sum(a + b for a, b, f, k in pairs) == data.count(b"\n") // 4 - want.count(b"\n") // 4    # every duplicate was byte-compared
    assert sum(f for a, b, f, k in pairs) == 0


def test_a_rank_that_fails_takes_the_others_with_it_and_nobody_hangs(tmp_path):
    data = dup_fastq(72, 4000)
    src = tmp_path / "in.fq"
    src.write_bytes(data)
    p = run([str(src), "0,0,0", str(tmp_path), "1"], timeout=120)
    assert p.returncode == 1 and b"failed=3" in p.stdout, (p.stdout, p.stderr[-1000:])
    assert b"rank 1: bsk_stats" in p.stderr and b"another rank failed" in p.stderr
