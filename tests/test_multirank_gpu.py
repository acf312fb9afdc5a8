"""The multi-rank operators with REAL processes on the GPU box: N ranks (torch.distributed.run), each with a
record-aligned shard in HBM and the HIP device phases of libbsk; the collectives are RCCL when every rank has a GPU of its
own and gloo through the host when the ranks share the GPU of a one-GPU box (bigseqkit_amd/dist.py coll_device).
Results are compared with the oracle on the whole text (StatsReduce bigseqkit/stats.go:91, GrepReduceCount grep.go:175,
GroupByKey of rmdup rmdup.go:97, MapWithIndex of range range.go:69-103, FileStore helper.go:378-460)."""
import json
import os
import random
import subprocess
import sys

import pytest

import oracle
import seqgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_on_the_gpu_box(world, tmp_path, record_property):
    rng = random.Random(4200 + world)
    data = seqgen.random_fastq(rng, 3000, 0, 200)
    # plant duplicates across the future shard cuts: the same records again at the end
    body = data.decode()
    lines = body.split("\n")
    dup = []
    for i in range(0, 400 * 4, 8):   # every other of the first 400 records (4 lines each)
        dup += ["@dup%d" % i] + lines[i + 1:i + 4]
    data = (body + "\n".join(dup) + "\n").encode()
    assert oracle.is_strict_4line_fastq(data)
    path = tmp_path / "in.fq"
    path.write_bytes(data)
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + world), os.path.join(ROOT, "tests", "_mr_gpu_worker.py"), str(tmp_path), str(path), "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]

    def cat(name):
        return b"".join((tmp_path / ("%s.%d" % (name, r))).read_bytes() for r in range(world))

    want_stats = oracle.stats_map(data, True, '{"All": true}')
    got_stats = dict((int(k), int(v)) for k, v in json.loads((tmp_path / "stats.0").read_bytes()))
    assert got_stats == want_stats
    for r in range(world):
        assert int((tmp_path / ("grepc.%d" % r)).read_bytes()) == oracle.grep(data, True, '{"Pattern": ["ACG"], "BySeq": true}').count(b"\n") // 4
    want_rmdup = oracle.rmdup(data, True, '{"BySeq": true}')
    assert cat("rmdup") == want_rmdup
    # round 6: EVERY duplicate was byte-compared with its survivor -- inside its shard or on the survivor's rank
    pairs = [json.loads((tmp_path / ("rmdup_pairs.%d" % r)).read_bytes()) for r in range(world)]
    n_records = data.count(b"\n") // 4
    assert sum(p[3] for p in pairs) == n_records
    assert sum(p[0] + p[1] for p in pairs) == n_records - want_rmdup.count(b"\n") // 4 > 0
    assert sum(p[1] for p in pairs) > 0 and sum(p[2] for p in pairs) == 0           # pairs crossed ranks; none differed
    # ... and with masked keys different sequences under one pair of keys are told apart by their text, across ranks too
    assert cat("rmdupx") == want_rmdup
    assert sum(json.loads((tmp_path / ("rmdupx_pairs.%d" % r)).read_bytes())[2] for r in range(world)) > 0
    assert cat("range") == oracle.range_(data, True, '{"Range": "3:-3"}')
    assert (tmp_path / "merged.fq").read_bytes() == oracle.seq(data, True, '{"Reverse": true}')
    # which collectives carried this run: RCCL when the box gave every rank a GPU, gloo through the host otherwise --
    # written next to the test log (gpurun_out/) and into the junit properties, so that a GPUTEST record says which
    backends = {(tmp_path / ("backend.%d" % r)).read_text() for r in range(world)}
    assert len(backends) == 1
    backend = backends.pop()
    import torch
    assert backend == ("nccl" if torch.cuda.device_count() >= world else "gloo")
    record_property("collective_backend", backend)
    print("multirank world=%d backend=%s gpus=%d" % (world, backend, torch.cuda.device_count()))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "multirank_backend_world%d.json" % world), "w") as f:
            json.dump({"world": world, "backend": backend, "gpus_visible": torch.cuda.device_count()}, f)
    except OSError:
        pass
