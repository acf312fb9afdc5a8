#!/usr/bin/env python3
"""Secondary measurements (1 GPU, HBM-resident): the other hot-path commands at the sizes of
BASELINE.json configs 2-5 (per-GPU shard where the config is an 8-GPU one).  Prints one JSON
object; algorithmic bytes per BASELINE.md section 4.  Not the driver's bench (that is bench.py)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None   # subset: seq,subseq,grep,locate,rmdup,translate


def want(name):
    return only is None or name in only


def synth(kind, flags, nbytes):
    rb = lib.bsk_synth_record_bytes(kind)
    n = int(nbytes) // rb * rb
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    check(lib.bsk_synth_device(kind, 42, flags, 0, C.c_void_p(t.data_ptr()), n, 0, None))
    torch.cuda.synchronize()
    return t, n // rb


last_stages = {}


def run(op_name, fn, opts, t, fmt):
    out = _lib.Out()
    with bsk.Operator(op_name, json.dumps(opts), 0) as op:
        check(fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, fmt, 0, None, C.byref(out)), op.ctx)  # warm-up
        torch.cuda.synchronize()
        prof = os.environ.get("BSK_BENCH_PROFILE") == "1"   # HIP-event time of the stages libbsk brackets, per call
        if prof:
            lib.bsk_profile_reset(op.ctx)
            lib.bsk_profile_enable(op.ctx, 1)
        t0 = time.perf_counter()
        pause = 0.02 if os.environ.get("BSK_TIMELINE") == "1" else 0.0   # scripts/timeline_ops.sh: calls apart in the trace
        for _ in range(reps):
            if pause:
                time.sleep(pause)
            check(fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, fmt, 0, None, C.byref(out)), op.ctx)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps - pause
        last_stages.clear()
        if prof:
            pb = C.create_string_buffer(1 << 16)
            check(lib.bsk_profile_dump(op.ctx, pb, len(pb)), op.ctx)
            for item in pb.value.decode().split(";"):
                if "=" in item:
                    k, v = item.rsplit("=", 1)
                    last_stages[k] = round(float(v.split("/")[0]) / reps, 3)
        return dt, out.len, out.records


res = {}


def report(name, nrec, in_bytes, dt, out_len, note=""):
    alg = in_bytes + out_len
    res[name] = {"records": nrec, "in_GB": round(in_bytes / 1e9, 2), "out_GB": round(out_len / 1e9, 3),
                 "ms": round(dt * 1e3, 2), "M_records_per_s": round(nrec / dt / 1e6, 1),
                 "algorithmic_GBps": round(alg / dt / 1e9, 1), "frac_of_8TBps": round(alg / dt / 8e12, 4),
                 "note": note + (" " + json.dumps(last_stages) if last_stages else "")}


# C2: seq -n on 100 GB FASTQ-150
if want("seq") or want("subseq"):
    t, nrec = synth(0, 0, 100e9 * scale)
    if want("seq"):
        dt, ol, k = run("SeqTransform", lib.bsk_seq_run, {"Name": True}, t, 1)
        assert ol == 12 * nrec or os.environ.get("BSK_DIAG")  # (BSK_DIAG: experiment builds with parts of a kernel removed)
        report("seq -n (C2, 100 GB FASTQ)", nrec, t.numel(), dt, ol)
    if want("subseq"):
        dt, ol, k = run("SubseqTransform", lib.bsk_subseq_run, {"Region": "1:50"}, t[:317 * (nrec // 4)], 1)
        report("subseq -r 1:50 (25 GB FASTQ)", nrec // 4, 317 * (nrec // 4), dt, ol)
    del t
# full re-emit with a transformation: reverse complement of 25 GB FASTQ (algorithmic bytes = 2 x in)
if want("seqrc"):
    t, nrec = synth(0, 0, 25e9 * scale)
    dt, ol, k = run("SeqTransform", lib.bsk_seq_run, {"Reverse": True, "Complement": True}, t, 1)
    report("seq -r -p (25 GB FASTQ)", nrec, t.numel(), dt, ol)
    del t
# whole-record operators on 25 GB FASTQ
if want("records"):
    t, nrec = synth(0, 0, 25e9 * scale)
    dt, ol, k = run("Fq2Fa", lib.bsk_fq2fa_run, {}, t, 1)
    report("fq2fa (25 GB FASTQ)", nrec, t.numel(), dt, ol)
    dt, ol, k = run("Duplicate", lib.bsk_duplicate_run, {"Times": 2}, t, 1)
    report("duplicate -n 2 (25 GB FASTQ)", nrec, t.numel(), dt, ol)
    rng = lambda ctx, p, n, dev, fmt, pid, st, out: lib.bsk_range_run(ctx, p, n, dev, fmt, pid, 0, st, out)
    dt, ol, k = run("Range", rng, {"Range": "1000001:%d" % (nrec - 1000000)}, t, 1)
    report("range -r 1000001:-1000001 (25 GB FASTQ)", nrec, t.numel(), dt, ol)
    dt, ol, k = run("Head", rng, {"N": 1000}, t, 1)
    report("head -n 1000 (25 GB FASTQ)", nrec, t.numel(), dt, ol)
    del t
if want("pair"):
    # two files of 12.5 GB with the same IDs in the same order (the usual paired-end layout)
    t, nrec = synth(0, 0, 12.5e9 * scale)
    both = torch.cat([t, t])
    del t
    outs = (_lib.Out * 4)()
    with bsk.Operator("Pair", "{}", 0) as op:
        for i in range(reps + 1):
            if i == 1:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            check(lib.bsk_pair_run(op.ctx, C.c_void_p(both.data_ptr()), both.numel(), both.numel() // 2, 1, 1, None, outs), op.ctx)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        assert outs[0].records == nrec and outs[1].records == nrec
        report("pair (2 x 12.5 GB FASTQ, all reads paired)", 2 * nrec, both.numel(), dt, outs[0].len + outs[1].len)
    del both
if want("common") or want("concat"):
    t, nrec = synth(0, 0, 12.5e9 * scale)
    both = torch.cat([t, t])
    del t
    out = _lib.Out()
    if want("common"):
        ends = (C.c_uint64 * 2)(both.numel() // 2, both.numel())
        with bsk.Operator("Common", "{}", 0) as op:
            for i in range(reps + 1):
                if i == 1:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                check(lib.bsk_common_run(op.ctx, C.c_void_p(both.data_ptr()), both.numel(), ends, 2, 1, 1, None, C.byref(out)), op.ctx)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            assert out.records == nrec
            report("common (2 x 12.5 GB FASTQ, same IDs)", 2 * nrec, both.numel(), dt, out.len)
    if want("concat"):
        with bsk.Operator("Concat", "{}", 0) as op:
            for i in range(reps + 1):
                if i == 1:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                check(lib.bsk_concat_run(op.ctx, C.c_void_p(both.data_ptr()), both.numel(), both.numel() // 2, 1, 1, None, C.byref(out)), op.ctx)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            assert out.records == nrec
            report("concat (2 x 12.5 GB FASTQ, same IDs)", 2 * nrec, both.numel(), dt, out.len)
    del both
if want("faidx"):
    t, nrec = synth(2, 0, 50e9 * scale)
    fai = lambda ctx, p, n, dev, fmt, pid, st, out: lib.bsk_faidx_run(ctx, p, n, dev, fmt, pid, 0, st, out)
    dt, ol, k = run("Faidx", fai, {}, t, 0)
    report("faidx index rows (C4 input, 50 GB FASTA-5k)", nrec, t.numel(), dt, ol)
    del t
if want("sort"):
    t, nrec = synth(0, 0, 25e9 * scale)
    dt, ol, k = run("Sort", lib.bsk_sort_run, {"ByLength": True, "Reverse": True}, t, 1)
    report("sort -l -r (25 GB FASTQ, all lengths equal)", nrec, t.numel(), dt, ol)
    dt, ol, k = run("Sort", lib.bsk_sort_run, {"Reverse": True}, t, 1)
    report("sort -r by ID (25 GB FASTQ, 11-byte IDs)", nrec, t.numel(), dt, ol)
    half = t[:317 * (nrec // 8)]
    dt, ol, k = run("Sort", lib.bsk_sort_run, {"BySeq": True}, half, 1)
    report("sort -s (3.1 GB FASTQ, 150-base keys)", nrec // 8, half.numel(), dt, ol)
    del t, half
if want("rename"):
    # 25 GB with 20 % duplicated sequences but unique names: nothing to rename; and 12.5 GB given twice: every ID twice
    t, nrec = synth(0, 0, 25e9 * scale)
    dt, ol, k = run("Rename", lib.bsk_rename_run, {}, t, 1)
    report("rename, unique IDs (25 GB FASTQ)", nrec, t.numel(), dt, ol)
    half = t[:317 * (nrec // 2)]
    t2 = torch.cat([half, half])
    del t, half
    dt, ol, k = run("Rename", lib.bsk_rename_run, {}, t2, 1)
    report("rename, every ID twice (25 GB FASTQ)", 2 * (nrec // 2), t2.numel(), dt, ol)
    del t2
# C3: grep -s -p motif, one GPU's 12.5 GB shard
if want("grep") or want("locate") or want("grepid"):
    t, nrec = synth(0, _lib.SYNTH_FLAG_MOTIF, 12.5e9 * scale)
    if want("grep"):
        dt, ol, k = run("Grep", lib.bsk_grep_run, {"BySeq": True, "Pattern": ["ACGTTGCAAGCT"]}, t, 1)
        report("grep -s -p 12-mer (C3 shard, 12.5 GB FASTQ)", nrec, t.numel(), dt, ol, "hits=%d" % k)
    if want("grepid"):
        # seqkit's most common grep: a list of IDs (here 100 000 of them, every 300th record) -> device hash set
        import tempfile
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write("".join("S%010d\n" % i for i in range(0, 30_000_000, 300)))
        dt, ol, k = run("Grep", lib.bsk_grep_run, {"PatternFile": f.name}, t, 1)
        os.unlink(f.name)
        report("grep -f 100k IDs (12.5 GB FASTQ)", nrec, t.numel(), dt, ol, "hits=%d" % k)
    if want("locate"):
        dt, ol, k = run("Locate", lib.bsk_locate_run, {"Pattern": ["ACGTTGCAAGCT"]}, t, 1)
        report("locate -p 12-mer (12.5 GB FASTQ)", nrec, t.numel(), dt, ol, "rows=%d" % k)
    del t
# C5: rmdup -s, one GPU's 25 GB shard
if want("rmdup"):
    t, nrec = synth(0, _lib.SYNTH_FLAG_DUPS, 25e9 * scale)
    dt, ol, k = run("RmDup", lib.bsk_rmdup_run, {"BySeq": True}, t, 1)
    assert k == nrec - nrec // 5
    report("rmdup -s (C5 shard, 25 GB FASTQ, 20% dups)", nrec, t.numel(), dt, ol, "survivors=%d" % k)
    del t
# C4: translate --frame 6 on 50 GB FASTA (5 kb CDS)
if want("translate"):
    t, nrec = synth(2, 0, 50e9 * scale)
    dt, ol, k = run("Translate", lib.bsk_translate_run, {"Frame": ["6"]}, t, 0)
    report("translate -f 6 (C4, 50 GB FASTA-5k)", nrec, t.numel(), dt, ol)
    del t
print(json.dumps(res))
