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
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines   # ONE JSON line on stdout (the collective libraries' greetings go to stderr)
    return json.loads(lines[0])


def test_bench_single_gpu_line_small():
    d = run_bench(["--gb", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["bit_exact_vs_expected_row"] is True
    assert d["stats_all"]["verified"] is True
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["avg_launch_ms"] > 0
    assert d["unit"] == "M records/s" and d["higher_is_better"] is True


@pytest.mark.parametrize("n", [2, 3])
def test_bench_n_ranks_sharing_one_gpu(n):
    d = run_bench(["--gpus", str(n), "--gb", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--ops-scale", "0.02",
                   "--ops-calls", "2"], {"BSK_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == n
    assert d["bit_exact_vs_expected_row"] is True and d["stats_all"]["verified"] is True
    assert d["allreduce_ms_per_step"] is not None
    # (on a box with >= n GPUs the ranks get a GPU each and the collective is RCCL)
    assert d["backend"] in ("gloo", "nccl")
    assert abs(d["shard_bytes_per_rank"] * n - d["config"]["bytes"]) <= 317 * n
    # the configs that are defined on several GPUs: grep @ C3 (sum all-reduce of the counts) and rmdup @ C5 (tuple
    # all-to-all, a duplicate's first occurrence may sit on the rank before), survivors resident in HBM
    ops = d["ops"]
    assert "error" not in ops, ops
    g, r = ops["grep -s -p @ C3"], ops["rmdup -s @ C5"]
    assert g["exact"] is True and g["n_gpus"] == n and g["hits"] >= g["planted"] > 0 and g["out_bytes"] == 317 * g["hits"], g
    assert r["exact"] is True and r["n_gpus"] == n and r["survivors"] == r["records"] - r["records"] // 5, r
    assert set(r["phases_ms_per_rank"]) == {"keys", "pack", "all_to_all", "resolve", "reply", "xpack", "xchange", "xcompare", "xreply", "xapply", "emit"}
    # round 6: every duplicate was byte-compared with its survivor, most of them across ranks (a duplicate's original is up to
    # 1 001 records earlier: with shards this small that is often the rank before)
    assert r["compared_pairs"] == r["duplicates"] == r["records"] // 5 and r["flagged_records"] == 0
    assert sum(r["pairs_byte_compared_across_ranks_per_rank"]) > 0 and sum(r["subject_bytes_sent_per_rank"]) == 150 * sum(r["pairs_byte_compared_across_ranks_per_rank"])
    assert r["without_cross_rank_comparison"]["exact"] is True
    assert len(r["per_rank_own_ms"]) == n and all(b > 0 for b in r["tuple_bytes_sent_per_rank"])
    assert sum(r["tuple_bytes_sent_per_rank"]) == 24 * r["records"]


def test_bench_one_rank_over_rccl():
    """BSK_BENCH_DIST_SINGLE=1: the N-rank path with ONE rank -- process group "nccl", the all-reduce of every step, the
    grep count all-reduce, the rmdup tuple exchange (all_gather + three all_to_all_single) -- so that the RCCL calls of
    that path run on a one-GPU box at all.  A small message limit forces the exchange into rounds."""
    d = run_bench(["--gpus", "1", "--gb", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--ops-scale", "0.02",
                   "--ops-calls", "2"], {"BSK_BENCH_DIST_SINGLE": "1", "BSK_A2A_MAX_BYTES": str(8 << 20)})
    assert d["single_rank_dist_check"] is True and d["backend"] == "nccl" and d["n_gpus"] == 1
    assert d["bit_exact_vs_expected_row"] is True and d["allreduce_ms_per_step"] is not None
    ops = d["ops"]
    assert "error" not in ops, ops
    g, r = ops["grep -s -p @ C3"], ops["rmdup -s @ C5"]
    assert g["exact"] is True and g["backend"] == "nccl", g
    assert r["exact"] is True and r["survivors"] == r["records"] - r["records"] // 5, r
    assert r["phases_ms_per_rank"]["all_to_all"][0] > 0   # the exchange ran (24 bytes per record to "the owner")


def test_bench_ops_object_small():
    """the 'ops' object of the driver-run line (seq -n @ C2, grep @ C3 shard, translate @ C4, rmdup @ C5 shard) at 1 / 50
    of the BASELINE sizes: every entry must carry its timing, its algorithmic bytes and an exact full-output check"""
    d = run_bench(["--gb", "2", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1", "--ops-scale", "0.02", "--ops-calls", "2"])
    ops = d["ops"]
    assert "error" not in ops, ops
    stats_legs = {"stats @ FASTA-1k (C1 layout, 20 GB)", "stats -a @ FASTA-1k (C1 layout, 20 GB)", "stats @ C4 input (50 GB FASTA-5k)"}
    assert set(ops) == {"seq -n @ C2", "subseq -r 1:50 (25 GB)", "grep -s -p @ C3 shard", "translate -f 6 @ C4",
                        "translate -f 6 @ C4, records that differ", "rmdup -s @ C5 shard"} | stats_legs
    for name, e in ops.items():
        assert e["exact"] is True, (name, e)
        assert e["ms"] > 0 and e["algorithmic_bytes"] >= e["in_bytes"] and 0 < e["frac"] < 1, (name, e)
        assert e["kernels_ms_per_call"], (name, e)
    g = ops["grep -s -p @ C3 shard"]
    assert g["hits"] >= g["planted"] > 0 and g["out_bytes"] == 317 * g["hits"]
    r = ops["rmdup -s @ C5 shard"]
    assert r["survivors"] == r["records"] - r["records"] // 5
    t = ops["translate -f 6 @ C4"]
    assert t["out_records"] == 6 * t["records"] and t["out_bytes"] == 10298 * t["records"]
    assert "k_translate_uniform" in t["kernels_ms_per_call"]       # the C4 layout needs no table
    tv = ops["translate -f 6 @ C4, records that differ"]           # ... records that differ do
    assert "k_translate_uniform" not in tv["kernels_ms_per_call"] and tv["out_records"] == 6 * tv["records"] and tv["shape_classes_checked"] >= 3
    assert "k_translate_stream" in tv["kernels_ms_per_call"]       # found, placed and written in one pass over the file
    for name in stats_legs:                                        # FASTA `stats` (the pass of stream_fasta2_dev.hpp), exact maps
        assert "k_stats" in ops[name]["kernels_ms_per_call"] and 0 < ops[name]["kernel_frac"] < 1, ops[name]
    # the byte-comparing default (one pass: comparison + placement) and the two-key mode beside it
    assert r["rmdup_keys"].startswith("verify") and "k_rmdup_place" in r["kernels_ms_per_call"]
    assert r["rmdup_keys_two_key"]["exact"] is True and "k_rmdup_sizes" in r["rmdup_keys_two_key"]["kernels_ms_per_call"]
    # a CPU baseline (the oracle: a port, 1 thread and all cores) beside every operator, and the host-bytes-to-result leg
    for name, e in ops.items():
        if name in stats_legs or name == "translate -f 6 @ C4, records that differ":
            continue
        cb = e["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0, (name, cb)
        if name != "rmdup -s @ C5 shard":
            assert cb["all_cores"]["value"] > 0 and cb["all_cores"]["cores"] >= 1, (name, cb)
        assert e["host_ms_per_call"] is not None
    e2e = d["end_to_end"]
    assert "error" not in e2e, e2e
    legs = [v for k, v in e2e.items() if isinstance(v, dict)]
    assert len(legs) == 5 and all(v["exact"] is True for v in legs), e2e
    assert sum("fresh process" in k for k in e2e) == 2, e2e     # FILE -> result through the native driver
    ec = d["end_to_end_config_size"]                               # round 6: FILE -> result on the whole (here: scaled) C2 file in /dev/shm
    assert "error" not in ec, ec
    if "skipped" not in ec:
        assert ec["stats (file -> row on stdout; fresh process)"]["exact"] is True and ec["grep -s -p (file -> one file; fresh process)"]["exact"] is True, ec
    y = d["vendor_yardsticks"]
    assert "error" not in y and y["d2d_copy_GBps_read_plus_write"] > 0 and y["read_sum_int64_GBps"] > 0, y
