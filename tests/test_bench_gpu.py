"""bench.py end to end on the GPU box: the single-GPU line at a small size, and the N-rank path (self-launch, record-aligned
shards of the synthetic file, one sum all-reduce of the stats vector = StatsReduce, bigseqkit/stats.go:91) with two ranks
that share the one GPU of the box (BSK_BENCH_SHARE_GPU=1: gloo collectives through the host; a functional check, not a
scaling measurement)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_single_gpu_line_small():
    d = run_bench(["--gb", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["bit_exact_vs_expected_row"] is True
    assert d["stats_all"]["verified"] is True
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["avg_launch_ms"] > 0
    assert d["unit"] == "M records/s" and d["higher_is_better"] is True


@pytest.mark.parametrize("n", [2, 3])
def test_bench_n_ranks_sharing_one_gpu(n):
    d = run_bench(["--gpus", str(n), "--gb", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                  {"BSK_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == n
    assert d["bit_exact_vs_expected_row"] is True and d["stats_all"]["verified"] is True
    assert d["allreduce_ms_per_step"] is not None
    # (on a box with >= n GPUs the ranks get a GPU each and the collective is RCCL)
    assert d["backend"] in ("gloo", "nccl")
    assert abs(d["shard_bytes_per_rank"] * n - d["config"]["bytes"]) <= 317 * n
