"""Randomised: the command line on ONE device with the shard brought whole, against the same command with the shard
streamed from the file piece by piece (what an input that does not fit the device takes: cli/bigseqkit.cpp, round 6 --
record commands through bsk_run_to_store in record-aligned pieces, `stats` in pinned pieces) and against 2 - 3 workers that
share the GPU.  Files come from the generators of test_fuzz_gpu.py (hundreds of few-byte records, records of 5 - 60 kb
among them); piece and chunk sizes are drawn small enough that a piece ends inside a long record now and then.  The bytes
must be the same, or both launches fail alike.  Workers are partitions: each guesses its alphabet from ITS first record, as
an executor of the reference does (helper.go:286-291) -- so only the commands that do not look at the alphabet are held to
the one-device output there (a shard that begins with a record without bases searches one strand only, complements nothing).  (Every output here is held to the oracle elsewhere: this test holds the
paths of the driver to each other, as tests/test_devices_native_gpu.py does on fixed inputs.)"""
import os
import random
import subprocess

import pytest

from test_run_multi_gpu import CLI, ROOT, read_out
from test_fuzz_gpu import rand_fasta, rand_fastq

pytestmark = pytest.mark.gpu

FQ = [["seq"], ["seq", "-n"], ["seq", "-n", "-i"], ["seq", "-m", "20"], ["seq", "-r", "-p"], ["seq", "-s"], ["grep", "-s", "-p", "ACG"],
      ["grep", "-p", "r1"], ["grep", "-n", "-r", "-p", "d$"], ["grep", "-s", "-v", "-p", "AC"], ["subseq", "-r", "1:20"], ["subseq", "-r", "-10:-1"],
      ["locate", "-p", "ACG"], ["locate", "-i", "-p", "acgt"], ["fq2fa"], ["stats", "-a", "-T"], ["stats", "-T"],
      ["rmdup", "-s"], ["rmdup"], ["rmdup", "-n"], ["rmdup", "-s", "-i"]]   # (rmdup: the workers exchange keys and texts; not streamed)
FA = [["seq"], ["seq", "-n"], ["seq", "-s", "-w", "0"], ["seq", "-m", "50", "-w", "70"], ["grep", "-s", "-p", "ACGT"], ["grep", "-p", "s1"],
      ["subseq", "-r", "2:30"], ["locate", "-p", "GAT"], ["translate", "-f", "6", "-x"], ["translate", "-f", "1", "-x", "-w", "0"], ["stats", "-a", "-T"],
      ["stats", "-T"], ["rmdup", "-s"], ["rmdup", "-n", "-i"]]


NEEDS_ALPHABET = (["seq", "-r", "-p"], ["grep", "-s"], ["locate"], ["translate"])


def alphabet_free(args):
    return not any(args[:len(p)] == p for p in NEEDS_ALPHABET)


def launch(cmd, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PATH"] = "/nonexistent"
    env.update(env_extra)
    p = subprocess.run(cmd, capture_output=True, cwd=ROOT, env=env, timeout=600)
    return p.returncode, p.stdout, p.stderr.decode(errors="replace")


@pytest.mark.parametrize("seed", range(max(1, int(os.environ.get("BSK_FUZZ_SEEDS", "24")) // 12)))
def test_streamed_and_shared_shards_write_what_one_whole_shard_writes(seed, tmp_path):
    rng = random.Random(52000 + seed)
    for it in range(6):
        fastq = rng.random() < 0.6
        data = rand_fastq(rng) if fastq else rand_fasta(rng)
        args = list(rng.choice(FQ if fastq else FA))
        src = str(tmp_path / ("in%d.%s" % (it, "fq" if fastq else "fa")))
        open(src, "wb").write(data)
        to_stdout = args[0] in ("stats", "locate")
        outs = []
        envs = [({}, "0"),
                ({"BSK_HOST_PIPELINE_FROM": "0", "BSK_STAGE_BYTES": str(rng.choice([4096, 16384, 65536])),
                  "BSK_STREAM_PIECE_BYTES": str(rng.choice([3000, 20000, 300000])), "BSK_STATS_PIECE_BYTES": str(rng.choice([65536, 70000, 1 << 20]))}, "0"),
                ({"BSK_SHARD_PIECE_BYTES": "4096"}, rng.choice(["0,0", "0,0,0"]))]
        for k, (env, devices) in enumerate(envs):
            out = str(tmp_path / ("o%d_%d" % (it, k)))
            cmd = [CLI] + args + [src, "--devices", devices] + ([] if to_stdout else ["-o", out, "--merge"])
            rc, so, se = launch(cmd, env)
            outs.append((rc, so if to_stdout else (read_out(out) if rc == 0 and os.path.exists(out) else b""), se[-300:]))
        ctx = (seed, it, args, len(data), data[:120], [o[2] for o in outs])
        if args[0] == "rmdup":      # (a global command: a shard that does not fit is refused, not streamed)
            outs[1] = outs[0]
        assert outs[0][0] == outs[1][0], ctx                    # both answer, or both fail
        if alphabet_free(args):
            assert outs[0][0] == outs[2][0], ctx
        if outs[0][0] == 0:
            assert outs[0][1] == outs[1][1], ("streamed", ctx)
            if alphabet_free(args):
                assert outs[0][1] == outs[2][1], ("workers", ctx)


def test_a_pinned_alphabet_guess_is_not_a_given_sequence_type(tmp_path):
    """seed 6 of the test above, round 6: a FASTA file whose first record reads as DNA ("G") and whose others hold U.  Brought
    whole, `seq` prints it (nothing validates: no -t, no -v).  Streamed in pieces, the guess of the first piece is held for
    the others -- one partition, one guess (helper.go:286-291) -- and `seq` took that pinned guess for a -t, switched
    validation on (seq.go:66-72 does so for a GIVEN type) and refused the file."""
    data = b">s0 desc\nG\n" + b"".join(b">s%d desc\nUUaAAgU\nCCuAucC\nGucucCG\n" % k for k in range(1, 400))
    src = str(tmp_path / "in.fa")
    open(src, "wb").write(data)
    outs = []
    for k, env in enumerate(({}, {"BSK_HOST_PIPELINE_FROM": "0", "BSK_STAGE_BYTES": "4096", "BSK_STREAM_PIECE_BYTES": "3000"})):
        out = str(tmp_path / ("o%d" % k))
        rc, so, se = launch([CLI, "seq", "-s", "-w", "0", src, "--devices", "0", "-o", out, "--merge"], env)
        assert rc == 0, se
        outs.append(read_out(out))
    assert outs[0] == outs[1] and outs[0].count(b"\n") == 400
    # ... while a type that IS given validates on both paths alike
    for env in ({}, {"BSK_HOST_PIPELINE_FROM": "0", "BSK_STAGE_BYTES": "4096", "BSK_STREAM_PIECE_BYTES": "3000"}):
        rc, so, se = launch([CLI, "seq", "-t", "dna", "-s", src, "--devices", "0", "-o", str(tmp_path / "x"), "--merge"], env)
        assert rc != 0 and "invalid" in se, se
