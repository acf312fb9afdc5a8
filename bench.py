#!/usr/bin/env python3
"""bench.py -- `bigseqkit stats` on the BASELINE C2 workload (100 GB synthetic FASTQ,
150 bp reads, 317 B/record), HBM-resident, through the C ABI of libbsk.so.

One step == one complete Stats over the whole file:
    zero the stats vector -> bsk_stats_run (k_prep + k_stats, one pass over the shard)
    -> StatsReduce across ranks (ONE sum all-reduce of the stats vector over RCCL, N>1 only)
    -> bsk_stats_collect (map[int64]int64) -> Stats()/StatsString() on the host.
The file is cut into N record-aligned shards (strong scaling: total work is fixed).
Prints ONE JSON line (rank 0).  See DESIGN.md section 6 for how each number is defined.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REC = 317
FILE_BYTES = 100_000_000_000
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


# ---------------------------------------------------------------------------------------------------------------------
# "ops": the other BASELINE configs, measured by the same process the driver runs (N = 1).
#   seq -n @ C2 (the 100 GB file of the stats legs), grep -s -p ACGTTGCAAGCT @ one GPU's C3 shard (12.5 GB, planted motif),
#   translate -f 6 @ C4 (50 GB FASTA-5k CDS), rmdup -s @ one GPU's C5 shard (25 GB, 20 % duplicates).
# Every entry: `ms` = mean wall time of `calls` whole operator calls through the C ABI after a warm-up, each one
# HIP-synchronised (record table / selection, sizes, scan, emit -- everything the call does), `algorithmic_bytes` per
# BASELINE.md section 4, `frac` = algorithmic bytes / ms / 8 TB/s, `kernels` = HIP-event time per call of the stages libbsk
# brackets (bsk_profile_dump), and `exact` = the COMPLETE output compared byte for byte with an expectation computed here
# with torch from the fixed layout of the synthetic file (no parser, no oracle).
# ---------------------------------------------------------------------------------------------------------------------
class _Helpers:
    """what the 'ops' legs share: device views of libbsk's output, timed operator calls, the exact row-wise comparison and
    the synthetic inputs"""

    def __init__(self, args, torch, bsk, _lib, lib, check, dev, local):
        self.args, self.torch, self.bsk, self._lib, self.lib, self.check, self.dev, self.local = \
            args, torch, bsk, _lib, lib, check, dev, local
        self.calls = max(1, args.ops_calls)

    def dev_bytes(self, ptr, n):
        """torch uint8 view of n device bytes at ptr (libbsk's output buffer), no copy"""
        if int(n) == 0:
            return self.torch.empty(0, dtype=self.torch.uint8, device=self.dev)

        class _Arr:  # __cuda_array_interface__ works for HIP pointers in torch-rocm
            pass
        a = _Arr()
        a.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        return self.torch.as_tensor(a, device=self.dev)

    def kernels(self, op, calls):
        pb = C.create_string_buffer(1 << 16)
        self.check(self.lib.bsk_profile_dump(op.ctx, pb, len(pb)), op.ctx)
        kern = {}
        for item in pb.value.decode().split(";"):
            if "=" in item:
                k, v = item.rsplit("=", 1)
                ms, n = v.split("/")
                kern[k] = round(float(ms) / calls, 4)
        return kern

    def timed_calls(self, name, fn, opts, buf, nbytes, fmt, sets=()):
        torch, lib, check, calls = self.torch, self.lib, self.check, self.calls
        out = self._lib.Out()
        op = self.bsk.Operator(name, json.dumps(opts), self.local)
        for k, v in sets:
            check(lib.bsk_ctx_set(op.ctx, k, v), op.ctx)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(fn(op.ctx, C.c_void_p(buf.data_ptr()), nbytes, 1, fmt, 0, st, C.byref(out)), op.ctx)  # warm-up (allocations)
        torch.cuda.synchronize()
        times = []
        for _ in range(calls):   # the timed calls: as a caller gets them, no event brackets
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            check(fn(op.ctx, C.c_void_p(buf.data_ptr()), nbytes, 1, fmt, 0, st, C.byref(out)), op.ctx)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        # the same calls once more with HIP events around every stage libbsk brackets (bsk_profile_*): where the time goes.
        # (Round 3 timed WITH the brackets on; two event records per stage cost a 9-stage call ~0.2 ms.)
        lib.bsk_profile_reset(op.ctx)
        lib.bsk_profile_enable(op.ctx, 1)
        for _ in range(calls):
            check(fn(op.ctx, C.c_void_p(buf.data_ptr()), nbytes, 1, fmt, 0, st, C.byref(out)), op.ctx)
            torch.cuda.synchronize()
        lib.bsk_profile_enable(op.ctx, 0)
        return op, out, sum(times) / calls, min(times), self.kernels(op, calls)

    def both_outputs(self, name, fn, opts, buf, nbytes, fmt):
        """The operators that can leave their result as ORDERED SLICES (round 6; include/bsk.h bsk_out.d_seg_*: the reference's
        Call returns []string whose elements share the partition's bytes): the same call timed with the switch "out" =
        "slices" -- the figure of the entry -- and as one block (round 5's contract: "contiguous" inside the entry).  The
        result handed back is the slices call's, made one block AFTER the clock stopped (bsk_out_materialize) so that `exact`
        compares the concatenation of the slices byte for byte."""
        lib, check = self.lib, self.check
        op_c, out_c, mean_c, min_c, kern_c = self.timed_calls(name, fn, opts, buf, nbytes, fmt)
        contiguous = {"ms": round(mean_c * 1e3, 4), "ms_min": round(min_c * 1e3, 4), "kernels_ms_per_call": kern_c,
                      "out_bytes": int(out_c.len), "out_records": int(out_c.records)}
        op_c.close()
        self.torch.cuda.empty_cache()
        op, out, mean_s, min_s, kern = self.timed_calls(name, fn, opts, buf, nbytes, fmt, sets=((b"out", b"slices"),))
        info = {"output": "ordered slices (bsk_ctx_set(ctx, 'out', 'slices')): %d segments, d_data NULL" % int(out.n_segments)
                if out.n_segments else "one block (the call did not qualify for slices)",
                "contiguous": contiguous}
        check(lib.bsk_out_materialize(op.ctx, C.byref(out), None), op.ctx)   # (untimed: for the comparison below)
        self.torch.cuda.synchronize()
        return op, out, mean_s, min_s, kern, info

    def entry(self, cmd, workload, nrec, in_bytes, alg_bytes, out, mean_s, min_s, kern, exact, how, extra=None):
        e = {"command": cmd, "workload": workload, "records": int(nrec), "in_bytes": int(in_bytes),
             "out_bytes": int(out.len), "out_records": int(out.records), "calls": self.calls,
             "ms": round(mean_s * 1e3, 4), "ms_min": round(min_s * 1e3, 4),
             "M_records_per_s": round(nrec / mean_s / 1e6, 2),
             "algorithmic_bytes": int(alg_bytes), "achieved_GBps": round(alg_bytes / mean_s / 1e9, 1),
             "frac": round(alg_bytes / mean_s / 1e9 / HBM_PEAK_GBS, 4),
             "kernels_ms_per_call": kern,
             "host_ms_per_call": round(mean_s * 1e3 - sum(kern.values()), 4) if kern else None,
             "exact": bool(exact), "exact_how": how}
        if extra:
            e.update(extra)
        return e

    def rows_equal(self, out_t, view, mask_fn, width, rows_per_chunk=4_000_000):
        """out_t == concat(view[i] for i with mask_fn(i0, i1)[i - i0]) ; mask None = every row; width = bytes per row"""
        torch = self.torch
        pos, ok, n = 0, True, view.shape[0]
        for i0 in range(0, n, rows_per_chunk):
            i1 = min(n, i0 + rows_per_chunk)
            blk = view[i0:i1]
            if mask_fn is not None:
                blk = blk[mask_fn(i0, i1)]
            k = blk.shape[0] * width
            if pos + k > out_t.numel():
                return False, pos
            ok = ok and bool(torch.equal(out_t[pos:pos + k].view(-1, width), blk))
            pos += k
            del blk
        return ok and pos == out_t.numel(), pos

    def synth(self, kind, flags, want_bytes, first_record=0):
        torch = self.torch
        rb = self.lib.bsk_synth_record_bytes(kind)
        n = int(want_bytes) // rb * rb
        t = torch.empty(n, dtype=torch.uint8, device=self.dev)
        self.check(self.lib.bsk_synth_device(kind, 42, flags, first_record, C.c_void_p(t.data_ptr()), n, self.local, None))
        torch.cuda.synchronize()
        return t, n // rb

    def motif_hit_mask(self, view, planted, first_record=0):
        """rows of `view` (records of 317 bytes) whose bases hold ACGTTGCAAGCT or its reverse complement: a sliding compare"""
        torch, dev = self.torch, self.dev
        pats = [torch.tensor(list(p), dtype=torch.uint8, device=dev) for p in (b"ACGTTGCAAGCT", b"AGCTTGCAACGT")]  # + / -

        def hit_mask(i0, i1):
            seqs = view[i0:i1, 13:163]
            m = torch.zeros(i1 - i0, dtype=torch.bool, device=dev)
            for p in pats:
                w = seqs[:, 0:139] == p[0]
                for j in range(1, 12):
                    w &= seqs[:, j:j + 139] == p[j]
                m |= w.any(dim=1)
            idx = torch.arange(first_record + i0, first_record + i1, device=dev) % 100
            planted[0] += int(((idx == 0) | (idx == 50)).sum().item())
            return m
        return hit_mask


_MALLOC_SET = False


def _malloc_for_many_threads():
    """glibc's allocator for the all-cores legs: the restatement allocates per record like the Go code it restates; with 256
    threads the default thresholds give freed memory back to the kernel and map it again all the time (33 minutes of system
    time in a 5-minute run, one address space's mmap lock).  Keep what was freed: no trimming, no mmap for large blocks."""
    global _MALLOC_SET
    if _MALLOC_SET:
        return
    _MALLOC_SET = True
    try:
        libc = C.CDLL("libc.so.6")
        libc.mallopt(-1, 1 << 30)    # M_TRIM_THRESHOLD
        libc.mallopt(-3, 1 << 30)    # M_MMAP_THRESHOLD
        libc.mallopt(-2, 64 << 20)   # M_TOP_PAD
    except OSError:
        pass


def cpu_baseline_op(H, fn_name, opts, tensor, rec_bytes, fastq, all_cores=True, seconds=None, min_bytes=0):
    """The oracle (oracle/: C++ restatement of the reference's operator, `port`) on a bounded prefix of the SAME input, 1 thread
    and -- where the records are independent -- all host cores (one record-aligned slice per thread; ctypes releases the
    GIL).  A reported baseline, not a target; returns the `cpu_baseline` object of an `ops` entry.
    Round 5: through oracle.run_ptr -- the slice is read in place and the output goes into an uninitialised buffer sized
    for the operator.  Round 4 went through the test entries, which copy the input, zero-fill 4 x its size and copy the
    result: with 256 threads the page faults of those buffers were what got timed (`seq -n` 1.30 M records/s on one thread,
    1.69 on 256)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    _malloc_for_many_threads()
    if seconds is None:
        seconds = max(0.3, H.args.cpu_seconds / 6.0)   # (five operators x two legs beside the stats baseline: ~30 s in all)
    oj = json.dumps(opts)
    out_factor = {"seq": 1.0 / 16 if opts.get("Name") else 1.05, "grep": 1.0 / 8, "subseq": 0.5, "translate": 2.2, "rmdup": 1.01}[fn_name]
    nrec_all = tensor.numel() // rec_bytes
    nthr = max(1, min(os.cpu_count() or 1, 256)) if all_cores else 1
    pilot_rec = max(1, min(nrec_all, (4 << 20) // rec_bytes))
    host = tensor[:min(nrec_all, max(pilot_rec, (1 << 30) // rec_bytes)) * rec_bytes].cpu().numpy()   # (<= 1 GB; grown below if the legs want more)
    base = host.ctypes.data

    def call(first_rec, nrec):
        nb = nrec * rec_bytes
        return oracle.run_ptr(fn_name, base + first_rec * rec_bytes, nb, fastq, oj, int(nb * out_factor) + 4096)
    t0 = time.perf_counter()
    call(0, pilot_rec)
    rate = pilot_rec / max(1e-6, time.perf_counter() - t0)           # records / s, one thread
    srec = int(max(pilot_rec, min(nrec_all, max(rate * seconds, min_bytes // rec_bytes), (512 << 20) // rec_bytes)))
    per = max(1, int(min(nrec_all // nthr, rate * seconds, (64 << 20) // rec_bytes))) if all_cores else 0
    need = max(srec, per * nthr)
    if need * rec_bytes > host.nbytes:
        host = tensor[:need * rec_bytes].cpu().numpy()
        base = host.ctypes.data
    t0 = time.perf_counter()
    call(0, srec)
    ct = time.perf_counter() - t0
    out = {"value": round(srec / ct / 1e6, 4), "unit": "M records/s", "gb_per_s": round(srec * rec_bytes / ct / 1e9, 4), "cores": 1,
           "kind": "port", "sample": "oracle.%s (C++ restatement of the reference operator, NOT IgnisHPC/Go) on the first %d records "
                                     "(%.3f GB) of the same input, %.2f s, 1 thread of %d host cores" % (fn_name, srec, srec * rec_bytes / 1e9, ct, os.cpu_count())}
    if all_cores:
        # the restatement allocates per record like the Go code it restates: with every hardware thread at once the heaps of
        # 256 arenas grow through one address space's mmap lock and `subseq` ran at 5.6 M records/s on 256 threads against
        # 3.2 on one.  So the leg climbs -- 1/16, 1/4, all of the hardware threads -- and reports the best of them with the
        # thread count it used; it stops climbing when more threads gave less.
        try:
            from concurrent.futures import ThreadPoolExecutor
            best = None
            tried = []
            for T in sorted({max(1, nthr // 16), max(1, nthr // 4), nthr}):
                with ThreadPoolExecutor(T) as ex:
                    list(ex.map(lambda k: call(k * per, min(per, 64)), range(T)))   # threads started
                    t0 = time.perf_counter()
                    list(ex.map(lambda k: call(k * per, per), range(T)))
                    ct = time.perf_counter() - t0
                val = per * T / ct / 1e6
                tried.append("%d threads: %.2f" % (T, val))
                if best is None or val > best[0]:
                    best = (val, T, ct)
                elif val < 0.8 * best[0]:
                    break
            val, T, ct = best
            out["all_cores"] = {"value": round(val, 2), "unit": "M records/s", "gb_per_s": round(val * 1e6 * rec_bytes / 1e9, 2),
                                "cores": T, "kind": "port", "sample": "%d threads x %d records (%.2f GB), one pass, %.2f s; M records/s by thread count: %s"
                                                                      % (T, per, per * T * rec_bytes / 1e9, ct, ", ".join(tried))}
        except Exception as e:
            out["all_cores"] = {"error": str(e)[:200]}
    return out


def vendor_yardsticks(torch, dev, gb=20.0):
    """What VENDOR kernels move on this box, beside the 8 TB/s of the data sheet that `frac` is quoted against: a plain
    device-to-device copy (torch `copy_` = the runtime's copy kernel; read + write bytes) and a plain read (`torch.sum`
    over an int64 view).  Five calls each after one warm-up, HBM-resident."""
    n = int(gb * 1e9) // 8 * 8
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.random_(0, 255)

    def rate(fn, nbytes):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        return round(nbytes / ((time.perf_counter() - t0) / 5) / 1e9, 1)
    a64 = a.view(torch.int64)
    out = {"bytes": n, "d2d_copy_GBps_read_plus_write": rate(lambda: b.copy_(a), 2 * n), "read_sum_int64_GBps": rate(lambda: a64.sum(), n),
           "note": "torch / runtime kernels on the same GPU in the same run: what a copy and a reduction reach of the 8 000 GB/s peak"}
    del a, b, a64
    torch.cuda.empty_cache()
    return out


def run_end_to_end(H, gb=8.0):
    """File-to-result rates with the input in HOST memory (pinned): what a caller that hands libbsk a file's bytes gets --
    PCIe Gen5 x16 (~55 GB/s) in, and for the record operators the output back out and into files.  `stats` -> the map;
    `seq -n` -> one merged file (StoreFASTX); `grep -s -p` -> a directory of four part files written by four contexts at once
    (StoreFASTXN).  Wall clock of the calls, second run.  PCIe- / page-cache-bound by two orders of magnitude against the HBM-
    resident rates: reported beside them, never `value`."""
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    torch, lib, check, bsk, _lib = H.torch, H.lib, H.check, H.bsk, H._lib
    t, nrec = H.synth(_lib.SYNTH_FASTQ150, _lib.SYNTH_FLAG_MOTIF, gb * 1e9)
    n = t.numel()
    # what the device path answers on the same bytes (exact-checked in the ops leg): the numbers the files must hold
    out = _lib.Out()
    with bsk.Operator("Grep", json.dumps({"BySeq": True, "Pattern": ["ACGTTGCAAGCT"]}), H.local) as op:
        check(lib.bsk_grep_run(op.ctx, C.c_void_p(t.data_ptr()), n, 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out)), op.ctx)
        torch.cuda.synchronize()
        want_hits = int(out.records)
    h = lib.bsk_host_alloc(n)
    if not h:
        return {"error": "pinned allocation of %.1f GB failed" % (n / 1e9)}
    check(lib.bsk_device_copy(C.c_void_p(h), C.c_void_p(t.data_ptr()), n, 2))
    del t
    torch.cuda.empty_cache()
    res = {"input": "%.2f GB FASTQ-150 (%d records, C3 motif planted) in pinned host memory" % (n / 1e9, nrec),
           "note": "wall clock, host bytes -> result; PCIe- and page-cache-bound; never `value`"}
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    try:
        with bsk.Operator("Stats", "{}", H.local) as op:
            keys, vals, cnt = (C.c_int64 * 4096)(), (C.c_int64 * 4096)(), C.c_size_t()
            best = None
            for _ in range(2):
                check(lib.bsk_stats_reset(op.ctx, None), op.ctx)
                t0 = time.perf_counter()
                check(lib.bsk_stats_run(op.ctx, C.c_void_p(h), n, 0, bsk.FORMAT_FASTQ, 0, None, None), op.ctx)
                check(lib.bsk_stats_collect(op.ctx, None, keys, vals, 4096, C.byref(cnt)), op.ctx)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            m = dict(zip(keys[:cnt.value], vals[:cnt.value]))
            res["stats (host bytes -> map)"] = {"s": round(best, 4), "GB_per_s": round(n / best / 1e9, 2),
                                                "M_records_per_s": round(nrec / best / 1e6, 1), "exact": m.get(150) == nrec}

        def to_store(op_name, opts, parts, merge):
            path = os.path.join(base, "bsk_bench_e2e_%d" % os.getpid())
            best, ob, orec = None, 0, 0
            for _ in range(2):
                shutil.rmtree(path, ignore_errors=True)
                if os.path.exists(path):
                    os.unlink(path)
                st = C.c_void_p()
                assert lib.bsk_store_open(path.encode(), 1 if merge else 0, C.byref(st)) == 0
                per = nrec // parts
                bounds = [(k * per * REC, ((k + 1) * per if k + 1 < parts else nrec) * REC) for k in range(parts)]
                ops = [bsk.Operator(op_name, json.dumps(opts), H.local) for _ in range(parts)]

                def work(k):
                    lo, hi = bounds[k]
                    nb, nr = C.c_uint64(), C.c_uint64()
                    check(lib.bsk_run_to_store(ops[k].ctx, C.c_void_p(h + lo), hi - lo, bsk.FORMAT_FASTQ, k, st, k, C.byref(nb), C.byref(nr)), ops[k].ctx)
                    return nb.value, nr.value
                t0 = time.perf_counter()
                with ThreadPoolExecutor(parts) as ex:
                    got = list(ex.map(work, range(parts)))
                tot = C.c_uint64()
                assert lib.bsk_store_close(st, C.byref(tot)) == 0
                dt = time.perf_counter() - t0
                for o in ops:
                    o.close()
                ob, orec = sum(g[0] for g in got), sum(g[1] for g in got)
                assert tot.value == ob
                best = dt if best is None else min(best, dt)
            on_disk = sum(os.path.getsize(os.path.join(path, f)) for f in os.listdir(path)) if os.path.isdir(path) else os.path.getsize(path)
            shutil.rmtree(path, ignore_errors=True)
            if os.path.exists(path):
                os.unlink(path)
            return best, ob, orec, on_disk

        dt, ob, orec, disk = to_store("SeqTransform", {"Name": True}, 1, True)
        res["seq -n (host bytes -> one merged file in %s)" % base] = {
            "s": round(dt, 4), "in_GB_per_s": round(n / dt / 1e9, 2), "out_bytes": ob, "M_records_per_s": round(nrec / dt / 1e6, 1),
            "exact": ob == 12 * nrec and orec == nrec and disk == ob}
        dt, ob, orec, disk = to_store("Grep", {"BySeq": True, "Pattern": ["ACGTTGCAAGCT"]}, 4, False)
        res["grep -s -p (host bytes -> 4 part files in %s, 4 contexts at once)" % base] = {
            "s": round(dt, 4), "in_GB_per_s": round(n / dt / 1e9, 2), "out_bytes": ob, "M_records_per_s": round(nrec / dt / 1e6, 1),
            "exact": orec == want_hits and ob == REC * want_hits and disk == ob}
        res.update(file_to_result(base, h, n, nrec, want_hits))
    finally:
        lib.bsk_host_free(C.c_void_p(h))
    return res


def file_to_result_config_size(H, gb):
    """FILE -> result at the CONFIG's size (BASELINE.md section 4 asks for it beside the kernel-
    resident number): the whole synthetic C2 / C3 file -- `gb` GB of FASTQ-150, motif planted -- written to /dev/shm (not
    timed), then `bigseqkit stats -T <file> --devices 0` and `bigseqkit grep -s -p <motif> <file> -o <out> --merge --devices 0`,
    a fresh process each: the steady-state rate of bsk_shard_load (8 readers, pread || H2D) on a file that is ~200 times the
    start-up cost.  Skipped, with the reason, when the box cannot hold the file in memory."""
    import shutil
    import subprocess
    torch, lib, check, bsk, _lib = H.torch, H.lib, H.check, H.bsk, H._lib
    cli = os.path.join(ROOT, "bigseqkit_amd", "bin", "bigseqkit")
    base = "/dev/shm"
    n = int(gb * 1e9) // REC * REC
    nrec = n // REC
    if not os.path.exists(cli) or not os.path.isdir(base):
        return {"skipped": "no command line binary / no /dev/shm"}
    free = shutil.disk_usage(base).free
    avail = 0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    if free < n * 1.1 + (8 << 30) or avail < n * 1.25 + (48 << 30):
        return {"skipped": "the %.0f GB file does not fit this box's memory next to its page-cache copy's reader (free in /dev/shm %.0f GB, "
                           "MemAvailable %.0f GB)" % (n / 1e9, free / 1e9, avail / 1e9)}
    src = os.path.join(base, "bsk_bench_c2_%d.fastq" % os.getpid())
    dst = os.path.join(base, "bsk_bench_c2_out_%d" % os.getpid())
    res = {"input": "%.1f GB FASTQ-150 (%d records, the C2 / C3 file, motif planted) in %s" % (n / 1e9, nrec, base)}
    try:
        # the file: produced on the device piece by piece, through one pinned buffer, written with 8 threads on disjoint ranges
        piece = (2 << 30) // REC * REC
        hbuf = lib.bsk_host_alloc(piece)
        if not hbuf:
            return {"skipped": "no pinned buffer for writing the file"}
        t0 = time.perf_counter()
        want_hits = 0
        try:
            dpiece = torch.empty(piece, dtype=torch.uint8, device=H.dev)
            view = memoryview((C.c_char * piece).from_address(hbuf)).cast("B")
            fd = os.open(src, os.O_CREAT | os.O_TRUNC | os.O_WRONLY, 0o600)
            out = _lib.Out()
            with bsk.Operator("Grep", json.dumps({"BySeq": True, "Pattern": ["ACGTTGCAAGCT"]}), H.local) as op:
                at = 0
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(8) as ex:
                    while at < n:
                        ln = min(piece, n - at)
                        assert lib.bsk_synth_device(_lib.SYNTH_FASTQ150, 42, _lib.SYNTH_FLAG_MOTIF, at // REC, C.c_void_p(dpiece.data_ptr()), ln, H.local, None) == 0
                        torch.cuda.synchronize()
                        check(lib.bsk_grep_run(op.ctx, C.c_void_p(dpiece.data_ptr()), ln, 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out)), op.ctx)
                        torch.cuda.synchronize()
                        want_hits += int(out.records)   # (what the device path answers on the same bytes; exact-checked in the ops leg)
                        check(lib.bsk_device_copy(C.c_void_p(hbuf), C.c_void_p(dpiece.data_ptr()), ln, 2))
                        step = (ln + 7) // 8
                        list(ex.map(lambda k: os.pwrite(fd, view[k * step:min(ln, (k + 1) * step)], at + k * step) if k * step < ln else 0, range(8)))
                        at += ln
            os.close(fd)
            del dpiece
        finally:
            lib.bsk_host_free(C.c_void_p(hbuf))
        torch.cuda.empty_cache()
        res["file_written_s"] = round(time.perf_counter() - t0, 2)
        assert os.path.getsize(src) == n

        def run(args):
            t0 = time.perf_counter()
            p = subprocess.run([cli] + args, capture_output=True, timeout=900, env=dict(os.environ, TMPDIR=base))
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                raise RuntimeError("%s: rc %d: %s" % (" ".join(args), p.returncode, p.stderr.decode(errors="replace")[-300:]))
            return dt, p.stdout
        dt, outp = run(["stats", "-T", src, "--devices", "0"])
        row = outp.decode().strip().split("\n")[-1].split("\t")
        res["stats (file -> row on stdout; fresh process)"] = {
            "s": round(dt, 3), "GB_per_s": round(n / dt / 1e9, 2), "M_records_per_s": round(nrec / dt / 1e6, 1),
            "exact": len(row) > 4 and row[3].replace(",", "") == str(nrec) and row[4].replace(",", "") == str(150 * nrec)}
        dt, outp = run(["grep", "-s", "-p", "ACGTTGCAAGCT", src, "-o", dst, "--merge", "--devices", "0"])
        ob = os.path.getsize(dst) if os.path.isfile(dst) else -1
        res["grep -s -p (file -> one file; fresh process)"] = {
            "s": round(dt, 3), "in_GB_per_s": round(n / dt / 1e9, 2), "out_bytes": ob, "M_records_per_s": round(nrec / dt / 1e6, 1),
            "exact": ob == REC * want_hits}
    except Exception as e:
        res["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    finally:
        for q in (src, dst):
            if os.path.isdir(q):
                shutil.rmtree(q, ignore_errors=True)
            elif os.path.exists(q):
                os.unlink(q)
    return res


def file_to_result(base, h, n, nrec, want_hits):
    """FILE -> result through the native driver `bigseqkit_amd/bin/bigseqkit <cmd> <file> --devices 0`: a fresh
    process per call (exec, HIP + RCCL start-up, the file's bytes from the page cache into pinned memory, copy to the GPU,
    kernels, the answer on stdout / in a file).  The input is the same sample, written to `base` first (not timed)."""
    import subprocess
    cli = os.path.join(ROOT, "bigseqkit_amd", "bin", "bigseqkit")
    if not os.path.exists(cli):
        return {"file -> result": {"error": "bigseqkit_amd/bin/bigseqkit is not built"}}
    src = os.path.join(base, "bsk_bench_in_%d.fastq" % os.getpid())
    dst = os.path.join(base, "bsk_bench_out_%d" % os.getpid())
    res = {}
    try:
        view = memoryview((C.c_char * n).from_address(h)).cast("B")
        with open(src, "wb", buffering=0) as f:
            at = 0
            while at < n:
                at += f.write(view[at:at + (256 << 20)])

        def run(args):
            best, outp = None, b""
            for _ in range(2):
                t0 = time.perf_counter()
                p = subprocess.run([cli] + args, capture_output=True, timeout=600, env=dict(os.environ, TMPDIR=base))
                dt = time.perf_counter() - t0
                if p.returncode != 0:
                    raise RuntimeError("%s: rc %d: %s" % (" ".join(args), p.returncode, p.stderr.decode(errors="replace")[-300:]))
                best, outp = (dt if best is None else min(best, dt)), p.stdout
            return best, outp
        dt, outp = run(["stats", "-T", src, "--devices", "0"])
        row = outp.decode().strip().split("\n")[-1].split("\t")
        res["stats (file in %s -> row on stdout; fresh process)" % base] = {
            "s": round(dt, 4), "GB_per_s": round(n / dt / 1e9, 2), "M_records_per_s": round(nrec / dt / 1e6, 1),
            "exact": len(row) > 4 and row[3].replace(",", "") == str(nrec) and row[4].replace(",", "") == str(150 * nrec)}
        dt, outp = run(["grep", "-s", "-p", "ACGTTGCAAGCT", src, "-o", dst, "--merge", "--devices", "0"])
        ob = os.path.getsize(dst) if os.path.isfile(dst) else -1
        res["grep -s -p (file in %s -> one file; fresh process)" % base] = {
            "s": round(dt, 4), "in_GB_per_s": round(n / dt / 1e9, 2), "out_bytes": ob, "M_records_per_s": round(nrec / dt / 1e6, 1),
            "exact": ob == REC * want_hits}
    except Exception as e:
        res["file -> result"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    finally:
        import shutil
        for q in (src, dst):
            if os.path.isdir(q):
                shutil.rmtree(q, ignore_errors=True)
            elif os.path.exists(q):
                os.unlink(q)
    return res


def stats_leg(H, cmd, t, nrec, fmt, opts, want, workload, how, pre_ok=True):
    """One `stats` leg of the 'ops' object: `calls` whole steps (reset -> bsk_stats_run -> bsk_stats_collect, each one
    synchronised by the collect), the map compared with `want` (every key of `want` must be there with that value, and the
    length bins must hold nothing else), HIP-event time of the stages in a second set of calls."""
    torch, lib, check, bsk = H.torch, H.lib, H.check, H.bsk
    keys, vals, cnt = (C.c_int64 * 65536)(), (C.c_int64 * 65536)(), C.c_size_t()
    n = t.numel()
    with bsk.Operator("Stats", json.dumps(opts), H.local) as op:
        def step():
            check(lib.bsk_stats_reset(op.ctx, None), op.ctx)
            check(lib.bsk_stats_run(op.ctx, C.c_void_p(t.data_ptr()), n, 1, fmt, 0, None, None), op.ctx)
            check(lib.bsk_stats_collect(op.ctx, None, keys, vals, 65536, C.byref(cnt)), op.ctx)
        step()
        times = []
        for _ in range(H.calls):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
        m = dict(zip(keys[:cnt.value], vals[:cnt.value]))
        lib.bsk_profile_reset(op.ctx)
        lib.bsk_profile_enable(op.ctx, 1)
        for _ in range(H.calls):
            step()
        lib.bsk_profile_enable(op.ctx, 0)
        kern = H.kernels(op, H.calls)
    lens = {k: v for k, v in m.items() if k >= 0}
    ok = bool(pre_ok) and lens == {k: v for k, v in want.items() if k >= 0} and all(m.get(k, 0) == v for k, v in want.items() if k < 0)
    mean_s = sum(times) / len(times)
    e = {"command": cmd, "workload": workload, "records": int(nrec), "in_bytes": int(n), "calls": H.calls,
         "ms": round(mean_s * 1e3, 4), "ms_min": round(min(times) * 1e3, 4), "M_records_per_s": round(nrec / mean_s / 1e6, 2),
         "algorithmic_bytes": int(n), "achieved_GBps": round(n / mean_s / 1e9, 1), "frac": round(n / mean_s / 1e9 / HBM_PEAK_GBS, 4),
         "kernels_ms_per_call": kern, "host_ms_per_call": round(mean_s * 1e3 - sum(kern.values()), 4) if kern else None,
         "exact": ok, "exact_how": how}
    if "k_stats" in kern and kern["k_stats"] > 0:
        e["kernel_frac"] = round(n / (kern["k_stats"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return e


def run_ops(args, torch, bsk, _lib, lib, check, dev, local, shard, total_rec):
    ops = {}
    H = _Helpers(args, torch, bsk, _lib, lib, check, dev, local)
    calls, dev_bytes, timed_calls, entry, rows_equal, synth = H.calls, H.dev_bytes, H.timed_calls, H.entry, H.rows_equal, H.synth

    # ---- seq -n @ C2: the names of the 100 GB file -----------------------------------------------------------------
    nbytes = total_rec * REC
    op, out, mean_s, min_s, kern, both = H.both_outputs("SeqTransform", lib.bsk_seq_run, {"Name": True}, shard, nbytes, bsk.FORMAT_FASTQ)
    names = dev_bytes(out.d_data, out.len)
    view = shard.view(total_rec, REC)
    ok = out.len == 12 * total_rec and out.records == total_rec
    if ok:
        ok, _ = rows_equal(names, view[:, 1:13], None, 12, 16_000_000)
        ok = ok and bytes(names[:12].cpu().tolist()) == b"S0000000000\n" \
            and bytes(names[-12:].cpu().tolist()) == b"S%010d\n" % (total_rec - 1)
    ops["seq -n @ C2"] = entry("seq -n", "%.1f GB FASTQ-150 (the file of the stats legs)" % (nbytes / 1e9), total_rec, nbytes,
                               nbytes + 12 * total_rec, out, mean_s, min_s, kern, ok,
                               "output == columns [1, 13) of every 317-byte record (torch.equal over all %d names), "
                               "12 x N bytes, first / last name" % total_rec,
                               dict(both, **({} if args.no_cpu_baseline else {"cpu_baseline": cpu_baseline_op(H, "seq", {"Name": True}, shard, REC, True)})))
    del names
    op.close()

    # ---- subseq -r 1:50 on the first quarter of that file (25 GB): header, 50 bases, '+', 50 qualities per record ------
    # (not a BASELINE config of its own: north_star lists subseq among the hot-path commands, VERDICT r02 named it the
    # kernel furthest from its roof after rmdup)
    nsub = total_rec // 4
    op, out, mean_s, min_s, kern, both = H.both_outputs("SubseqTransform", lib.bsk_subseq_run, {"Region": "1:50"}, shard, nsub * REC,
                                                      bsk.FORMAT_FASTQ)
    got = dev_bytes(out.d_data, out.len)
    ok = out.len == 117 * nsub and out.records == nsub
    if ok:
        sep = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=dev)
        nl = sep[:1]
        for i0 in range(0, nsub, 4_000_000):
            i1 = min(nsub, i0 + 4_000_000)
            k = i1 - i0
            want = torch.cat([view[i0:i1, 0:63], sep.expand(k, 3), view[i0:i1, 166:216], nl.expand(k, 1)], dim=1)
            ok = ok and bool(torch.equal(got[117 * i0:117 * i1].view(k, 117), want))
            del want
    ops["subseq -r 1:50 (25 GB)"] = entry(
        "subseq -r 1:50", "%.1f GB FASTQ-150 (the first quarter of the file of the stats legs)" % (nsub * REC / 1e9), nsub,
        nsub * REC, nsub * REC + 117 * nsub, out, mean_s, min_s, kern, ok,
        "output == columns [0, 63) ++ '\\n+\\n' ++ columns [166, 216) ++ '\\n' of every 317-byte record (torch.equal over all "
        "%d records)" % nsub,
        dict(both, **({} if args.no_cpu_baseline else {"cpu_baseline": cpu_baseline_op(H, "subseq", {"Region": "1:50"}, shard, REC, True)})))
    del got, view
    op.close()
    shard.data = torch.empty(0, dtype=torch.uint8, device=dev)  # the 100 GB file is not needed any more
    torch.cuda.empty_cache()  # libbsk allocates with hipMalloc, outside torch's pool

    # ---- stats @ FASTA-1k: BASELINE C1's layout (1 kb records in 60-column lines) at 20 GB -- the FASTA pass ------
    t, nrec = synth(_lib.SYNTH_FASTA1K, 0, 20e9 * args.ops_scale)
    wl = "%.1f GB FASTA, %d records of 1 000 bases in 60-column lines (the layout of BASELINE C1)" % (t.numel() / 1e9, nrec)
    v1k = t.view(nrec, 1027)
    cols = torch.tensor([10 + (b // 60) * 61 + (b % 60) for b in range(1000)], device=dev)
    layout_ok, gaps = True, 0
    for i0 in range(0, nrec, 2_000_000):     # the expectation from the bytes: '>' + 8 name bytes + '\n', 1 000 letters, 17 line breaks
        blk = v1k[i0:i0 + 2_000_000]
        layout_ok = layout_ok and bool((blk[:, 0] == ord(">")).all()) and bool((blk[:, 9] == 10).all()) \
            and int((blk == 10).sum().item()) == 18 * blk.shape[0]
        sq = blk[:, cols]
        gaps += int(((sq == ord("-")) | (sq == ord(" ")) | (sq == ord("."))).sum().item())
        layout_ok = layout_ok and bool((sq >= ord("A")).all())
        del blk, sq
    ops["stats @ FASTA-1k (C1 layout, 20 GB)"] = stats_leg(
        H, "stats", t, nrec, bsk.FORMAT_FASTA, {}, {1000: nrec}, wl,
        "map == {1000: N}: every record is '>' + 8 name bytes + a line break, then 1 000 letters with 17 line breaks "
        "(checked on the bytes with torch: markers, line-break count per record, letters at the sequence columns)", layout_ok)
    ops["stats -a @ FASTA-1k (C1 layout, 20 GB)"] = stats_leg(
        H, "stats -a", t, nrec, bsk.FORMAT_FASTA, {"All": True}, {1000: nrec, -3: gaps}, wl,
        "map == {1000: N, gap: the count of '-', ' ', '.' at the sequence columns (torch)}", layout_ok)
    del t, v1k, cols
    torch.cuda.empty_cache()

    # ---- grep -s -p ACGTTGCAAGCT @ C3: one GPU's 12.5 GB shard of the 100 GB file, motif planted in 2 % of the reads --
    t, nrec = synth(_lib.SYNTH_FASTQ150, _lib.SYNTH_FLAG_MOTIF, 12.5e9 * args.ops_scale)
    # (round 6: the hits are whole records of the shard -- with out=slices they stay where they are, 16 bytes of slice list
    # per hit; the one-block call of the same run is nested as `contiguous`)
    op, out, mean_s, min_s, kern, both = H.both_outputs("Grep", lib.bsk_grep_run, {"BySeq": True, "Pattern": ["ACGTTGCAAGCT"]}, t,
                                                        t.numel(), bsk.FORMAT_FASTQ)
    view = t.view(nrec, REC)
    planted = [0]
    hit_mask = H.motif_hit_mask(view, planted)

    got = dev_bytes(out.d_data, out.len)
    ok, pos = rows_equal(got, view, hit_mask, REC, 2_000_000)
    hits = pos // REC
    ok = ok and out.records == hits and hits >= planted[0]
    both["contiguous"]["algorithmic_bytes"] = int(t.numel() + out.len)
    both["contiguous"]["frac"] = round(both["contiguous"]["algorithmic_bytes"] / (both["contiguous"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    both["algorithmic_bytes_note"] = "slices: the input once + 16 bytes of slice list per hit (the hits stay in the shard); contiguous: + the hits written once"
    ops["grep -s -p @ C3 shard"] = entry(
        "grep -s -p ACGTTGCAAGCT", "%.2f GB FASTQ-150, one GPU's shard of C3, motif planted on + / - strand in 2 %% of the reads"
        % (t.numel() / 1e9), nrec, t.numel(), t.numel() + (16 * int(out.records) if "slices" in both["output"][:16] else out.len),
        out, mean_s, min_s, kern, ok,
        "output == the records whose bases hold the 12-mer or its reverse complement (sliding compare in torch over all "
        "records), in file order", dict(dict(both, hits=int(hits), planted=int(planted[0]), background=int(hits - planted[0])),
                                        **({} if args.no_cpu_baseline else {"cpu_baseline": cpu_baseline_op(
                                            H, "grep", {"BySeq": True, "Pattern": ["ACGTTGCAAGCT"]}, t, REC, True)})))
    del got, view, t
    op.close()
    torch.cuda.empty_cache()

    # ---- translate -f 6 @ C4: 50 GB FASTA, 5 kb CDS records wrapped at 60 -------------------------------------------
    # expectation: per input record six elements (frames 1, 2, 3, -1, -2, -3), each ">" + name + "\n" + protein wrapped at 60
    # + "\n"; table 1 (the standard code) in TCAG order -- computed here with torch from records of a KNOWN shape (hdr =
    # bytes of the header line with its newline, L bases in 60-column lines)
    aa_tab = torch.tensor(list(b"FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"), dtype=torch.uint8, device=dev)
    code = torch.zeros(256, dtype=torch.int64, device=dev)
    comp = torch.zeros(256, dtype=torch.int64, device=dev)
    for ch, v, cc in ((b"T", 0, b"A"), (b"C", 1, b"G"), (b"A", 2, b"T"), (b"G", 3, b"C")):
        code[ch[0]] = v
        comp[ch[0]] = cc[0]

    def shape_of(hdr, L):
        naa = [(L - k) // 3 for k in (0, 1, 2, 0, 1, 2)]
        el_len = [hdr + a + (a + 59) // 60 for a in naa]
        return naa, el_len, sum(el_len)

    def expected_translation(blk, hdr, L):
        """blk: (m, hdr + L + ceil(L / 60)) input records of one shape -> (m, OUT_RB) expected output, frame 1 of row 0"""
        m = blk.shape[0]
        naa, el_len, out_rb = shape_of(hdr, L)
        bidx = torch.arange(L, device=dev)
        seqs = blk[:, hdr + (bidx // 60) * 61 + (bidx % 60)].long()   # (m, L) bases
        rc = comp[seqs.flip(1)]                                        # reverse complement
        exp = torch.full((m, out_rb), 10, dtype=torch.uint8, device=dev)   # '\n' everywhere, then fill
        base, first = 0, None
        for e, (src, k) in enumerate(((seqs, 0), (seqs, 1), (seqs, 2), (rc, 0), (rc, 1), (rc, 2))):
            a = naa[e]
            c = code[src[:, k:k + 3 * a]].view(m, a, 3)
            prot = aa_tab[c[:, :, 0] * 16 + c[:, :, 1] * 4 + c[:, :, 2]]
            exp[:, base] = ord(">")
            exp[:, base + 1:base + hdr] = blk[:, 1:hdr]               # name + '\n'
            j = torch.arange(a, device=dev)
            exp[:, base + hdr + (j // 60) * 61 + (j % 60)] = prot
            if e == 0:
                first = prot[0]
            base += el_len[e]
        return exp, first

    t, nrec = synth(_lib.SYNTH_FASTA5K_CDS, 0, 50e9 * args.ops_scale)
    RB = 5107
    op, out, mean_s, min_s, kern = timed_calls("Translate", lib.bsk_translate_run, {"Frame": ["6"]}, t, t.numel(), bsk.FORMAT_FASTA)
    _, _, OUT_RB = shape_of(22, 5001)
    view = t.view(nrec, RB)
    got = dev_bytes(out.d_data, out.len)
    ok = out.len == OUT_RB * nrec and out.records == 6 * nrec
    first_protein = None
    if ok:
        gv = got.view(nrec, OUT_RB)
        chunk = 100_000
        for i0 in range(0, nrec, chunk):
            i1 = min(nrec, i0 + chunk)
            exp, first = expected_translation(view[i0:i1], 22, 5001)
            if first_protein is None:
                first_protein = bytes(first.cpu().tolist())
            ok = ok and bool(torch.equal(gv[i0:i1], exp))
            del exp
        ok = ok and first_protein is not None and first_protein[:1] == b"M" and first_protein[-1:] == b"*" \
            and len(first_protein) == 1667 and b"*" not in first_protein[:-1]
    ops["translate -f 6 @ C4"] = entry(
        "translate --frame 6", "%.1f GB FASTA, %d CDS records of 5 001 bases wrapped at 60 (ATG + 1 665 sense codons + TAA)"
        % (t.numel() / 1e9, nrec), nrec, t.numel(), t.numel() + out.len, out, mean_s, min_s, kern, ok,
        "output == six frames per record translated here with torch (standard code as a 64-entry gather, reverse strand = "
        "flipped complement), headers and 60-column wrapping included, all records; frame 1 of record 0 is M...* of 1 667 aa",
        dict({"layout_path": "uniform: every record has the shape of the first (UniformLayout, verified in-kernel: stage "
                             "'k_translate_uniform'%s); the leg below is the same command on records that differ"
                             % ("" if "k_translate_uniform" in kern else " -- NOT taken"),},
             **({} if args.no_cpu_baseline else {"cpu_baseline": cpu_baseline_op(H, "translate", {"Frame": ["6"]}, t, RB, False, min_bytes=int(0.5e9 * min(1.0, args.ops_scale * 50)))})))
    del got, view
    op.close()
    # the same file as the input of `stats` (FASTA, 5 kb records): the default row, exact from the fixed layout
    ops["stats @ C4 input (50 GB FASTA-5k)"] = stats_leg(H, "stats", t, nrec, bsk.FORMAT_FASTA, {}, {5001: nrec},
                                                        "%.1f GB FASTA, %d records of 5 001 bases in 60-column lines" % (t.numel() / 1e9, nrec),
                                                        "map == {5001: N} (every record of the synthetic layout has 5 001 bases: 22 header "
                                                        "bytes, 84 line breaks, 5 107 bytes per record)")
    del t
    torch.cuda.empty_cache()

    # ---- the same command on records that do NOT all look alike: unpadded record numbers in the header
    # (15 .. 22 header bytes), one record in a hundred 4 998 or 5 004 bases long -- UniformLayout is refused, the call takes
    # the table path (k_fasta_starts / k_fasta_heads + k_translate_wide<.., table>)
    VAR = _lib.SYNTH_FASTA5K_VAR
    nrec = 1
    while lib.bsk_synth_offset(VAR, nrec * 2) <= 50e9 * args.ops_scale:
        nrec *= 2
    lo_n, hi_n = nrec, nrec * 2
    while lo_n + 1 < hi_n:          # the largest record count whose file fits the size of C4
        mid = (lo_n + hi_n) // 2
        if lib.bsk_synth_offset(VAR, mid) <= 50e9 * args.ops_scale:
            lo_n = mid
        else:
            hi_n = mid
    nrec = lo_n
    nbytes_v = int(lib.bsk_synth_offset(VAR, nrec))
    t = torch.empty(nbytes_v, dtype=torch.uint8, device=dev)
    check(lib.bsk_synth_device(VAR, 42, 0, 0, C.c_void_p(t.data_ptr()), nbytes_v, local, None))
    torch.cuda.synchronize()
    op, out, mean_s, min_s, kern = timed_calls("Translate", lib.bsk_translate_run, {"Frame": ["6"]}, t, nbytes_v, bsk.FORMAT_FASTA)
    got = dev_bytes(out.d_data, out.len)
    # shapes by record number: digits d(i), bases L(i) (synth.hpp KIND_FASTA5K_VAR); offsets are prefix sums of the sizes
    idx = torch.arange(nrec, device=dev, dtype=torch.int64)
    dig = torch.ones(nrec, device=dev, dtype=torch.int64)
    p10 = 10
    while p10 <= nrec:
        dig += (idx >= p10).long()
        p10 *= 10
    Ls = torch.full((nrec,), 5001, device=dev, dtype=torch.int64)
    Ls[idx % 100 == 37] = 4998
    Ls[idx % 100 == 73] = 5004
    in_sz = 14 + dig + Ls + 84
    in_off = torch.cumsum(in_sz, 0) - in_sz
    out_sz = torch.zeros(nrec, device=dev, dtype=torch.int64)
    for k in (0, 1, 2, 0, 1, 2):
        a = (Ls - k) // 3
        out_sz += 14 + dig + a + (a + 59) // 60
    out_off = torch.cumsum(out_sz, 0) - out_sz
    ok = int(in_sz.sum().item()) == nbytes_v and out.len == int(out_sz.sum().item()) and out.records == 6 * nrec
    nclass = 0
    if ok:
        chunk = 50_000
        for i0 in range(0, nrec, chunk):
            i1 = min(nrec, i0 + chunk)
            dd, ll = dig[i0:i1], Ls[i0:i1]
            for d in torch.unique(dd).tolist():
                for L in (5001, 4998, 5004):
                    sel = torch.nonzero((dd == d) & (ll == L)).flatten() + i0
                    if sel.numel() == 0:
                        continue
                    nclass += 1
                    hdr = 14 + int(d)
                    rbc = hdr + L + 84
                    _, _, orb = shape_of(hdr, L)
                    blk = t[in_off[sel][:, None] + torch.arange(rbc, device=dev)[None, :]]
                    exp, _ = expected_translation(blk, hdr, L)
                    g = got[out_off[sel][:, None] + torch.arange(orb, device=dev)[None, :]]
                    ok = ok and bool(torch.equal(g, exp))
                    del blk, exp, g
    ops["translate -f 6 @ C4, records that differ"] = entry(
        "translate --frame 6", "%.1f GB FASTA, %d CDS records: '>cds%%d len=%%d' with unpadded numbers (15 .. 22 header bytes), "
        "99 %% of 5 001 bases, 1 %% of 4 998 / 5 004, wrapped at 60" % (nbytes_v / 1e9, nrec), nrec, nbytes_v, nbytes_v + out.len, out,
        mean_s, min_s, kern, ok,
        "output == the same torch translation, records gathered by shape class (header digits x length) from offsets that are "
        "prefix sums of the record sizes, every record compared",
        {"layout_path": "%s: the head sample's records differ, UniformLayout is refused ('k_translate_uniform' is %s the stages of "
                        "this call)" % ("one pass over the file that finds the records, places them through a look-back and writes "
                                        "their frames ('k_translate_stream'), no record table"
                                        if "k_translate_stream" in kern else "record table, then the translate kernel",
                                        "NOT among" if "k_translate_uniform" not in kern else "AMONG"),
         "shape_classes_checked": nclass})
    del got, t, idx, dig, Ls, in_sz, in_off, out_sz, out_off
    op.close()
    torch.cuda.empty_cache()

    # ---- rmdup -s @ C5: one GPU's 25 GB shard, every fifth record repeats the bases of an earlier one -------------------
    t, nrec = synth(_lib.SYNTH_FASTQ150, _lib.SYNTH_FLAG_DUPS, 25e9 * args.ops_scale)
    op, out, mean_s, min_s, kern, both = H.both_outputs("RmDup", lib.bsk_rmdup_run, {"BySeq": True}, t, t.numel(), bsk.FORMAT_FASTQ)
    view = t.view(nrec, REC)
    got = dev_bytes(out.d_data, out.len)
    keep = nrec - nrec // 5
    ok = out.records == keep and out.len == keep * REC
    if ok:
        ok, _ = rows_equal(got, view, lambda i0, i1: (torch.arange(i0, i1, device=dev) % 5) != 4, REC)
    # a second pass over the output removes nothing
    out2 = _lib.Out()
    with bsk.Operator("RmDup", json.dumps({"BySeq": True}), local) as op2:
        check(lib.bsk_rmdup_run(op2.ctx, C.c_void_p(out.d_data), out.len, 1, bsk.FORMAT_FASTQ, 0, None, C.byref(out2)), op2.ctx)
        torch.cuda.synchronize()
        ok = ok and out2.records == keep and out2.len == out.len
    # the same call with rmdup_keys = two-key: equal (k1, k2) decide without the byte comparison of the default (PARITY.md KEYS)
    op_k, out_k, mean_k, min_k, kern_k = timed_calls("RmDup", lib.bsk_rmdup_run, {"BySeq": True}, t, t.numel(), bsk.FORMAT_FASTQ,
                                                     sets=((b"rmdup_keys", b"two-key"),))
    ok_k = out_k.records == keep and out_k.len == keep * REC
    if ok_k:
        ok_k, _ = rows_equal(dev_bytes(out_k.d_data, out_k.len), view, lambda i0, i1: (torch.arange(i0, i1, device=dev) % 5) != 4, REC)
    op_k.close()
    extra = dict(both)
    extra["algorithmic_bytes_note"] = ("slices: the input once + 16 table bytes per record; the survivors are slices of the shard and are not "
                                       "moved (RmDupCheck's result elements are the strings it was handed, rmdup.go:200-222).  contiguous: + "
                                       "the output written once")
    extra["contiguous"]["algorithmic_bytes"] = int(t.numel() + 16 * nrec + out.len)
    extra["contiguous"]["frac"] = round(extra["contiguous"]["algorithmic_bytes"] / (extra["contiguous"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    extra.update({"survivors": int(out.records),
             "rmdup_keys": "verify (default): ONE 64-bit key groups the records -- the chain-free grouping key of csrc/hash_dev.hpp, whose "
                           "value nothing but the grouping sees -- and the sequence bytes of every duplicate are compared with its "
                           "survivor's (RmDupCheck, rmdup.go:193-199); a shard whose comparison meets two sequences under one key "
                           "runs again with XXH64 + the second key",
             "rmdup_keys_two_key": {"ms": round(mean_k * 1e3, 4), "ms_min": round(min_k * 1e3, 4), "kernels_ms_per_call": kern_k,
                                    "exact": bool(ok_k), "note": "bsk_ctx_set(ctx, 'rmdup_keys', 'two-key'): no byte comparison; one block"}})
    if not args.no_cpu_baseline:
        extra["cpu_baseline"] = cpu_baseline_op(H, "rmdup", {"BySeq": True}, t, REC, True, all_cores=False)
        extra["cpu_baseline"]["sample"] += " (duplicates are global: one thread, one group table)"
        # all cores: duplicates are global, so the threads share ONE call (oracle rmdup_call_mt: the records parsed and keyed by
        # `threads` threads, every key group settled by the thread that owns key % threads -- IgnisHPC's executors + GroupByKey
        # restated with threads).  A labelled baseline, never a target.
        try:
            import oracle
            srec = min(nrec, max(1, int(extra["cpu_baseline"]["value"] * 1e6 * max(0.5, args.cpu_seconds / 6.0) * 8)))
            host = t[:srec * REC].cpu().numpy()
            best, tried = None, []
            for T in sorted({max(1, (os.cpu_count() or 1) // 16), max(1, (os.cpu_count() or 1) // 4), os.cpu_count() or 1}):
                t0 = time.perf_counter()
                oracle.run_ptr("rmdup_mt", host.ctypes.data, srec * REC, True, json.dumps({"BySeq": True}), int(srec * REC * 1.01) + 4096, threads=T)
                ct = time.perf_counter() - t0
                val = srec / ct / 1e6
                tried.append("%d threads: %.2f" % (T, val))
                if best is None or val > best[0]:
                    best = (val, T, ct)
                elif val < 0.8 * best[0]:
                    break
            extra["cpu_baseline"]["all_cores"] = {"value": round(best[0], 2), "unit": "M records/s", "gb_per_s": round(best[0] * REC / 1e3, 2), "cores": best[1],
                                                  "kind": "port", "sample": "oracle.rmdup_mt on the first %d records (%.2f GB), ONE call on %d threads, %.2f s; "
                                                                            "M records/s by thread count: %s" % (srec, srec * REC / 1e9, best[1], best[2], ", ".join(tried))}
            del host
        except Exception as e:
            extra["cpu_baseline"]["all_cores"] = {"error": str(e)[:200]}
    ops["rmdup -s @ C5 shard"] = entry(
        "rmdup -s", "%.1f GB FASTQ-150, one GPU's shard of C5, record i with i %% 5 == 4 repeats the bases of an earlier record"
        % (t.numel() / 1e9), nrec, t.numel(), t.numel() + 16 * nrec + (0 if "slices" in both["output"] else out.len), out, mean_s, min_s, kern, ok,
        "output == the records with i % 5 != 4, byte for byte in file order (N - N // 5 survivors); rmdup of the output keeps "
        "every record", extra)
    del got, view, t
    op.close()
    torch.cuda.empty_cache()
    attach_traffic(ops)
    return ops


def cut_by_anchor(lib, flags, total_rec, k, world):
    """Record number at which shard k of `world` begins: the first record start at or behind the NOMINAL BYTE offset
    total_bytes * k / world, found by bsk_find_record_start (the ReadFixer rule every caller of the library cuts a file with)
    in a window of the synthetic file produced on the host around that offset -- not by record arithmetic."""
    if k <= 0:
        return 0
    if k >= world:
        return total_rec
    nominal = total_rec * REC * k // world
    first = max(0, nominal // REC - 2)
    nwin = min(total_rec - first, 4096) * REC
    buf = C.create_string_buffer(nwin)
    assert lib.bsk_synth_host(0, 42, flags, first, buf, nwin) == 0
    out = C.c_size_t()
    assert lib.bsk_find_record_start(C.cast(buf, C.c_void_p), nwin, nominal - first * REC, 1, C.byref(out)) == 0
    pos = first * REC + out.value
    assert pos % REC == 0 and nominal <= pos < nominal + REC, (pos, nominal)
    return pos // REC


def attach_traffic(ops):
    """`traffic` (HBM bytes per call from the PMC passes, a FETCH_SIZE factor per access shape) and `traffic_over_algorithmic`
    for the ops entries, from the committed evidence of the same commands at the same sizes (scripts/r05_evidence.sh ->
    profiles/*_ops_traffic.json; rocprofv3 cannot wrap the process that is being timed).  The counters belong to the kernel
    sources they were collected on: a changed source marks the figure STALE instead of passing it on silently."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_ops_traffic.json")))
    if not files:
        return
    try:
        tj = json.load(open(files[-1]))
    except ValueError:
        return
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "bigseqkit_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp", ".inc")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    src = "profiles/" + os.path.basename(files[-1]) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/bench_ops.py; " \
          "FETCH_SIZE x 2: a request moves the 128-byte line whatever part of it was asked for, profiles/r06_fetch_calibration.json)"
    if tj.get("kernel_sources_sha256") != h.hexdigest():
        src += " -- STALE: the kernel sources have changed since that pass"
    prefix = {"seq -n @ C2": "seq -n", "subseq -r 1:50 (25 GB)": "subseq", "grep -s -p @ C3 shard": "grep -s",
              "translate -f 6 @ C4": "translate", "rmdup -s @ C5 shard": "rmdup"}
    for name, pre in prefix.items():
        e = ops.get(name)
        t = next((v for k, v in tj.get("ops", {}).items() if k.startswith(pre)), None)
        if not isinstance(e, dict) or t is None:
            continue
        # (only at the sizes the counters were collected on)
        if abs(e["algorithmic_bytes"] / 1e9 - t["algorithmic_GB"]) > 0.05 * t["algorithmic_GB"]:   # (rmdup: bench.py adds the 16 B/record key exchange of SURVEY 8d to the algorithmic bytes)
            continue
        e["traffic"] = int(t["traffic_GB"] * 1e9)
        e["traffic_over_algorithmic"] = t["traffic_over_algorithmic"]
        e["traffic_upper_bound_over_algorithmic"] = t["upper_bound_over_algorithmic"]
        e["traffic_source"] = src
    # the kernels the ops evidence does not reach (scripts/r06_evidence.sh step 4: the 50 GB inputs); the newest file, held to
    # the kernel sources like the others (round 5's carried no hash: VERDICT r05 weak 7)
    xfs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_extra_traffic.json")))
    xf = xfs[-1] if xfs else ""
    if xf:
        try:
            xj = json.load(open(xf))
            xk = xj.get("kernels", {})
        except ValueError:
            xj, xk = {}, {}
        xstale = "" if xj.get("kernel_sources_sha256") == h.hexdigest() else " -- STALE: the kernel sources have changed since that pass"
        for name, pre in (("stats @ C4 input (50 GB FASTA-5k)", "k_stats<false, false"), ("translate -f 6 @ C4, records that differ", "k_translate_stream")):
            e = ops.get(name)
            t = next((v for k, v in xk.items() if k.startswith(pre)), None)
            if not isinstance(e, dict) or t is None or "traffic_over_algorithmic" not in t or "traffic" in e:
                continue
            if abs(e["algorithmic_bytes"] / 1e9 - t["algorithmic_GB"]) > 0.05 * t["algorithmic_GB"]:
                continue
            e["traffic"] = int((t["read_GB"] + t["write_GB"]) * 1e9)
            e["traffic_over_algorithmic"] = t["traffic_over_algorithmic"]
            if "read_over_input" in t:
                e["input_read_over_input"] = t["read_over_input"]
            e["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE of the dominant kernel of the same command " \
                                  "at the same size)%s" % (os.path.basename(xf), xstale)


def run_ops_multi(args, torch, bsk, _lib, lib, check, dev, local, rank, world, bdist, backend_name):
    """The two BASELINE configs that are DEFINED on several GPUs, at N > 1 (every rank calls this):
      grep -s -p ACGTTGCAAGCT @ C3 -- the 100 GB file (motif planted) in N record-aligned shards, bsk_grep_run per rank,
            the global number of selected records from ONE sum all-reduce (GrepReduceCount, bigseqkit/grep.go:175);
            the selected records stay in every rank's HBM (rank order == file order).  Strong scaling.
      rmdup -s @ C5 -- 25 GB of the duplicate-planted file per rank (C5 is 8 x 25 GB; weak scaling below 8 GPUs) through
            dist.rmdup_distributed: keys -> all_gather(counts) -> pack -> all_to_all(24-byte tuples to owner = key % N)
            -> resolve -> all_to_all(keep bytes) -> emit (GroupByKey + RmDupCheck, bigseqkit/rmdup.go:97); a duplicate's
            first occurrence may live on the rank before (the copy source is up to 1 001 records back); survivors stay in HBM.
    Timing: barrier + synchronize on both sides of `calls` whole calls, max over ranks.  `exact` = every rank's COMPLETE
    output compared with the expectation computed with torch from the fixed layout (no parser, no oracle), AND-ed over the
    ranks; the global counts are sums over the ranks of what torch counted."""
    H = _Helpers(args, torch, bsk, _lib, lib, check, dev, local)
    calls = H.calls
    ops = {}

    def job_time(fn):
        fn()  # warm-up (allocations, RCCL channel set-up)
        bdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            r = fn()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
        bdist.barrier()
        dt = bdist.all_reduce_max_float(time.perf_counter() - t0, dev)
        return r, dt / calls, own / calls

    def all_ok(ok):
        return bdist.all_reduce_count(0 if ok else 1, dev) == 0

    # ---- grep -s -p @ C3 over N GPUs -----------------------------------------------------------------------------------
    total = int(100e9 * args.ops_scale) // REC
    lo, hi = cut_by_anchor(lib, _lib.SYNTH_FLAG_MOTIF, total, rank, world), cut_by_anchor(lib, _lib.SYNTH_FLAG_MOTIF, total, rank + 1, world)
    t, nrec = H.synth(_lib.SYNTH_FASTQ150, _lib.SYNTH_FLAG_MOTIF, (hi - lo) * REC, lo)
    out = _lib.Out()
    op = bsk.Operator("Grep", json.dumps({"BySeq": True, "Pattern": ["ACGTTGCAAGCT"]}), local)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def grep_call():
        check(lib.bsk_grep_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, bsk.FORMAT_FASTQ, rank, st, C.byref(out)), op.ctx)
        return bdist.all_reduce_count(out.records, dev)

    lib.bsk_profile_reset(op.ctx)
    global_hits, per_call, own = job_time(grep_call)
    view = t.view(nrec, REC)
    planted = [0]
    got = H.dev_bytes(out.d_data, out.len)
    ok, pos = H.rows_equal(got, view, H.motif_hit_mask(view, planted, lo), REC, 2_000_000)
    ok = ok and out.records == pos // REC
    want_hits = bdist.all_reduce_count(pos // REC, dev)
    planted_all = bdist.all_reduce_count(planted[0], dev)
    out_bytes_all = bdist.all_reduce_count(out.len, dev)
    rows = bdist.all_gather_floats([own * 1e3, nrec], dev)
    ok = all_ok(ok) and global_hits == want_hits and want_hits >= planted_all
    alg = total * REC + out_bytes_all
    ops["grep -s -p @ C3"] = {
        "command": "grep -s -p ACGTTGCAAGCT", "n_gpus": world, "backend": backend_name, "scaling": "strong",
        "workload": "%.1f GB FASTQ-150 (C3), motif planted on + / - strand in 2 %% of the reads, %d record-aligned shards of %.2f GB"
                    % (total * REC / 1e9, world, (hi - lo) * REC / 1e9),
        "records": total, "in_bytes": total * REC, "out_bytes": out_bytes_all, "hits": int(global_hits),
        "planted": int(planted_all), "background": int(global_hits - planted_all), "calls": calls,
        "collective": "one sum all-reduce of the per-rank record counts (GrepReduceCount)",
        "ms": round(per_call * 1e3, 4), "M_records_per_s": round(total / per_call / 1e6, 2),
        "algorithmic_bytes": int(alg), "achieved_GBps": round(alg / per_call / 1e9, 1),
        "frac": round(alg / per_call / 1e9 / (HBM_PEAK_GBS * world), 4),
        "per_rank_own_ms": [round(r[0], 4) for r in rows], "per_rank_records": [int(r[1]) for r in rows],
        "exact": bool(ok),
        "exact_how": "every rank: output == the rows of its shard whose bases hold the 12-mer or its reverse complement "
                     "(torch sliding compare, all records); the all-reduced count == the sum of those row counts"}
    del got, view, t
    op.close()
    torch.cuda.empty_cache()

    # ---- rmdup -s @ C5 over N GPUs: 25 GB per rank ----------------------------------------------------------------------
    per = int(25e9 * args.ops_scale) // REC
    lo = cut_by_anchor(lib, _lib.SYNTH_FLAG_DUPS, per * world, rank, world)   # (== rank * per: the cut of a file of `world` such shards)
    assert lo == rank * per
    t, nrec = H.synth(_lib.SYNTH_FASTQ150, _lib.SYNTH_FLAG_DUPS, per * REC, lo)
    be = bdist.HipRmDupBackend(json.dumps({"BySeq": True}), local)
    text, per_call, own = job_time(lambda: bdist.rmdup_distributed(t, bsk.FORMAT_FASTQ, be, to_host=False))
    phases = {}
    text = bdist.rmdup_distributed(t, bsk.FORMAT_FASTQ, be, to_host=False, phases=phases)   # one more call, synchronised per phase
    view = t.view(nrec, REC)
    got = text.tensor()
    ok, _ = H.rows_equal(got, view, lambda i0, i1: (torch.arange(lo + i0, lo + i1, device=dev) % 5) != 4, REC)
    N = per * world
    surv_all = bdist.all_reduce_count(text.records, dev)
    out_bytes_all = bdist.all_reduce_count(text.len, dev)
    ok = all_ok(ok and text.len == REC * text.records) and surv_all == N - N // 5
    names = ["keys", "pack", "all_to_all", "resolve", "reply", "xpack", "xchange", "xcompare", "xreply", "xapply", "emit"]
    rows = bdist.all_gather_floats([own * 1e3] + [phases.get(k, 0.0) for k in names]
                                   + [phases.get("tuple_bytes_sent", 0), phases.get("tuple_bytes_sent_off_rank", 0)], dev)
    alg = N * REC + 16 * N + out_bytes_all
    # round 6: every duplicate is byte-compared with its survivor, the ones whose survivor lives on another rank through the
    # text exchange of dist._xcheck: (pairs inside the shard, pairs across ranks, flagged records, subject bytes sent) per rank
    lp, xp, fl = be.pair_stats()
    lp_rows = bdist.all_gather_floats([float(lp), float(xp), float(fl), float(phases.get("xcheck_text_bytes_sent", 0)), float(nrec - text.records)], dev)
    compared_all, dups_all = sum(int(r[0] + r[1]) for r in lp_rows), sum(int(r[4]) for r in lp_rows)
    ok = ok and compared_all == dups_all == N // 5 and sum(int(r[2]) for r in lp_rows) == 0
    # the same job with the cross-rank comparison switched off (round 5's decision rule: pairs that cross ranks rest on their
    # two keys): what the text exchange costs
    os.environ["BSK_RMDUP_XCHECK"] = "off"
    try:
        text_k, per_call_k, _ = job_time(lambda: bdist.rmdup_distributed(t, bsk.FORMAT_FASTQ, be, to_host=False))
        ok_k = all_ok(text_k.len == text.len and text_k.records == text.records)
    finally:
        del os.environ["BSK_RMDUP_XCHECK"]
    ops["rmdup -s @ C5"] = {
        "command": "rmdup -s", "n_gpus": world, "backend": backend_name, "scaling": "weak",
        "workload": "%.1f GB FASTQ-150 per rank x %d ranks (C5 is 8 x 25 GB), record i with i %% 5 == 4 repeats the bases of a "
                    "record up to 1 001 places earlier (possibly on the rank before)" % (per * REC / 1e9, world),
        "records": N, "in_bytes": N * REC, "out_bytes": out_bytes_all, "survivors": int(surv_all), "calls": calls,
        "collective": "all_gather(record counts) + all_to_all_single(24-byte tuples to owner = key %% N) + all_to_all_single(keep bytes, "
                      "survivor indices) + the text of every duplicate whose survivor lives on another rank to that rank "
                      "(24-byte requests + the bases) and one verdict byte back",
        "ms": round(per_call * 1e3, 4), "M_records_per_s": round(N / per_call / 1e6, 2),
        "algorithmic_bytes": int(alg), "achieved_GBps": round(alg / per_call / 1e9, 1),
        "frac": round(alg / per_call / 1e9 / (HBM_PEAK_GBS * world), 4),
        "per_rank_own_ms": [round(r[0], 4) for r in rows],
        "phases_ms_per_rank": {k: [round(r[1 + i], 4) for r in rows] for i, k in enumerate(names)},
        "tuple_bytes_sent_per_rank": [int(r[1 + len(names)]) for r in rows], "tuple_bytes_sent_off_rank_per_rank": [int(r[2 + len(names)]) for r in rows],
        "survivors_resident": "HBM (DeviceText: the context's output buffer; no host copy)",
        "rmdup_keys": "across ranks a record travels as (XXH64, second 64-bit key, global index) to owner = key % N, where equal (k1, k2) "
                      "GROUP -- the owner holds no text.  The owner's reply names the survivor, and EVERY duplicate is then byte-compared "
                      "with it (RmDupCheck's test, bigseqkit-lib/rmdup.go:193-211): inside its shard when the survivor lives there, else "
                      "its bases travel to the survivor's rank, which answers one byte.  Records whose text differs from their "
                      "survivor's are regrouped by text over all ranks (none here: real keys)",
        "pairs_byte_compared_per_rank": [int(r[0]) for r in lp_rows],
        "pairs_byte_compared_across_ranks_per_rank": [int(r[1]) for r in lp_rows],
        "compared_pairs": int(compared_all), "duplicates": int(dups_all), "flagged_records": int(sum(int(r[2]) for r in lp_rows)),
        "subject_bytes_sent_per_rank": [int(r[3]) for r in lp_rows],
        "without_cross_rank_comparison": {"ms": round(per_call_k * 1e3, 4), "exact": bool(ok_k),
                                          "note": "BSK_RMDUP_XCHECK=off: pairs that cross ranks rest on their two keys (round 5); "
                                                  "the difference to `ms` is what the text exchange costs"},
        "exact": bool(ok),
        "exact_how": "every rank: output == the records of its shard with GLOBAL index %% 5 != 4, byte for byte in file order; "
                     "survivors over all ranks == N - N // 5"}
    del got, view, t, text
    be.close()
    torch.cuda.empty_cache()
    return ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gb", type=float, default=FILE_BYTES / 1e9, help="size of the synthetic file (GB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-ops", action="store_true", help="skip the 'ops' object (seq -n / grep / translate / rmdup at the "
                                                           "BASELINE config sizes, N = 1 only)")
    ap.add_argument("--no-scaling-model", action="store_true", help="skip the single-GPU shard steps of 'scaling_model' (the "
                    "profiling passes: every k_stats dispatch of the run is then a whole-file one)")
    ap.add_argument("--no-e2e-full", action="store_true", help="skip the FILE -> result legs on the whole C2 file in /dev/shm")
    ap.add_argument("--ops-scale", type=float, default=1.0, help="scale the sizes of the 'ops' workloads (tests)")
    ap.add_argument("--ops-calls", type=int, default=5, help="timed calls per operator of the 'ops' object")
    ap.add_argument("--launch-check", action="store_true",
                    help="only prove that --gpus N ranks start and reduce together (backend from BSK_BENCH_BACKEND, "
                         "default nccl; the CPU test suite runs it with gloo), print one JSON line and exit")
    args = ap.parse_args()
    want_world = max(1, args.gpus)

    # ---- self-launch: `python bench.py --gpus N` starts N ranks itself (one process per GPU); under torchrun
    # (WORLD_SIZE set) this process IS one of the ranks.  Either way the world size must equal --gpus.
    if "WORLD_SIZE" not in os.environ and want_world > 1:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(want_world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != want_world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node == --gpus "
                         "(or without torchrun: bench.py starts the ranks itself)" % (want_world, world))
    backend = os.environ.get("BSK_BENCH_BACKEND", "nccl")
    # BSK_BENCH_DIST_SINGLE=1 (with --gpus 1): the N-rank code path -- process group, the all-reduce of every step, the
    # multi-GPU legs with their all-to-all -- with ONE rank.  On a one-GPU box this is the only way the RCCL calls of that
    # path run at all; the line says "single_rank_dist_check" and is no scaling measurement.
    dist_on = world > 1 or os.environ.get("BSK_BENCH_DIST_SINGLE") == "1"
    if dist_on and world == 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("BSK_DIST_SINGLE_RANK_COLLECTIVES", "1")
    if dist_on or args.launch_check:
        # RCCL (and gloo) write their greetings to file descriptor 1 -- RCCL's version banner appears when the communicator is
        # first used or torn down, i.e. AFTER the result line (seen with one rank over RCCL) -- and the contract is ONE JSON
        # line on stdout.  Descriptor 1 becomes the ranks' stderr; sys.stdout is re-opened on the real one.
        sys.stdout.flush()
        _real_out = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(_real_out, "w")
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
            one = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            one = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(one)
        assert dist.get_world_size() == want_world and int(one.item()) == want_world
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": dist.get_world_size(), "backend": backend,
                              "ranks_reduced": int(one.item())}), flush=True)
        dist.destroy_process_group()
        return

    import bigseqkit_amd as bsk
    from bigseqkit_amd import _lib, dist as bdist
    from bigseqkit_amd._lib import lib, check

    if lib.bsk_device_count() <= 0 or not torch.cuda.is_available():
        raise SystemExit("bench.py: no HIP device visible: the hot path has no CPU fallback (BSK_ERR_NO_DEVICE)")
    # BSK_BENCH_SHARE_GPU=1: a FUNCTIONAL check of the N-rank path on a box with fewer GPUs than ranks -- the ranks share
    # the devices (rank r on GPU r % count) and reduce over gloo (RCCL refuses two ranks on one device).  The line it prints
    # says so ("shared_gpu_functional_check"); it is not a scaling measurement.
    share = os.environ.get("BSK_BENCH_SHARE_GPU") == "1" and torch.cuda.device_count() < world
    if share:
        backend = "gloo"
        local = local % torch.cuda.device_count()
    if torch.cuda.device_count() < (local + 1):
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        pg_timeout = datetime.timedelta(minutes=5)   # a rank that died must not hold the others for the default 10 - 30 min
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=pg_timeout)
        assert dist.get_world_size() == want_world

    # ---- the synthetic file, cut into record-aligned shards --------------------------
    total_rec = int(args.gb * 1e9) // REC
    while True:
        lo, hi = cut_by_anchor(lib, 0, total_rec, rank, world), cut_by_anchor(lib, 0, total_rec, rank + 1, world)
        nrec = hi - lo
        # last whole record; the very last shard drops the final '\n' (a file need not end with one)
        nbytes = nrec * REC
        try:
            shard = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            break
        except RuntimeError:
            if world > 1:
                raise
            total_rec //= 2  # smaller HBM than an MI355X: say so in config.workload
    check(lib.bsk_synth_device(_lib.SYNTH_FASTQ150, 42, 0, lo, C.c_void_p(shard.data_ptr()), nbytes, local, None))
    torch.cuda.synchronize()

    def make_op(all_):
        o = bsk.SeqKitStatsOptions().Tabular(True).All(all_)
        op = bsk.Operator("Stats", o.to_json(), local)
        vlen = lib.bsk_stats_vector_len(op.ctx)
        vec = torch.zeros(vlen, dtype=torch.int64, device=dev)
        return op, vec

    reduce_ev = [] if dist_on else None  # (start, stop) HIP events around the all-reduce of every timed step

    def one_step(op, vec, nb=None):
        nb = nbytes if nb is None else nb
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        vec.zero_()
        check(lib.bsk_stats_reset(op.ctx, st), op.ctx)  # error flags + overflow list of the context (the vector is ours)
        check(lib.bsk_stats_run(op.ctx, C.c_void_p(shard.data_ptr()), nb, 1, bsk.FORMAT_FASTQ, rank,
                                C.c_void_p(vec.data_ptr()), st), op.ctx)
        if dist_on:
            # StatsReduce (bigseqkit/stats.go:91): ONE sum all-reduce of the dense map over RCCL -- the only collective --
            # and the collect behind it (the step's one synchronising copy; the overflow lists of chromosome-sized
            # records are exchanged only when the reduced vector counts any: never for reads).  HIP events on the stream
            # bracket the all-reduce (device time of the collective incl. the wait for the slowest rank), read after the
            # timed region.
            if reduce_ev is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                bdist.all_reduce_stats_vector(vec)
                ev[1].record()
                reduce_ev.append(ev)
                m = bdist.collect_reduced(op, vec, reduce=False)
            else:
                m = bdist.collect_reduced(op, vec)
        else:
            m = bsk.api._collect_map(op, C.c_void_p(vec.data_ptr()))
        info = bsk.api._finalize(op, m)
        buf = C.create_string_buffer(4096)
        check(lib.bsk_stats_string(op.ctx, b"input0", b"N/A", C.byref(info), buf, len(buf)), op.ctx)
        return m, buf.value.decode()

    def timed(op, vec, steps, warmup, nb=None):
        for _ in range(warmup):
            one_step(op, vec, nb)
        lib.bsk_profile_reset(op.ctx)
        lib.bsk_profile_enable(op.ctx, 1)
        if dist_on:
            bdist.barrier()
        torch.cuda.synchronize()
        if reduce_ev is not None:
            del reduce_ev[:]
        t0 = time.perf_counter()
        for _ in range(steps):
            m, text = one_step(op, vec, nb)
        torch.cuda.synchronize()
        t_own = time.perf_counter() - t0          # this rank's K steps (before it waits for the others)
        if dist_on:
            bdist.barrier()
        dt = time.perf_counter() - t0
        lib.bsk_profile_enable(op.ctx, 0)
        ms, n = C.c_double(), C.c_uint64()
        lib.bsk_profile_read(op.ctx, b"k_stats", C.byref(ms), C.byref(n))
        k_ms = ms.value / max(1, n.value)
        lib.bsk_profile_read(op.ctx, b"k_prep", C.byref(ms), C.byref(n))
        p_ms = ms.value / max(1, n.value)
        r_ms = sum(a.elapsed_time(b) for a, b in reduce_ev) / max(1, len(reduce_ev)) if reduce_ev else 0.0
        ranks = None
        if dist_on:
            dt = bdist.all_reduce_max_float(dt, dev)
            # what every rank saw: explains a step time that is not 1 / N of the single-GPU one
            rows = bdist.all_gather_floats([k_ms, p_ms, r_ms, t_own / steps * 1e3], dev)
            ranks = {"k_stats_ms": [round(r[0], 4) for r in rows], "k_prep_ms": [round(r[1], 4) for r in rows],
                     "allreduce_ms": [round(r[2], 4) for r in rows], "own_ms_per_step": [round(r[3], 4) for r in rows]}
            own = ranks["own_ms_per_step"]
            ranks["barrier_skew_ms_per_step"] = round(max(own) - min(own), 4)
        return dt, m, text, k_ms, p_ms, r_ms, ranks

    op, vec = make_op(False)
    dt, m, text, k_ms, p_ms, reduce_ms, per_rank = timed(op, vec, args.steps, args.warmup)
    want_row = "input0\tN/A\tDNA\t%d\t%d\t150\t150.0\t150" % (total_rec, total_rec * 150)
    verified = (m.get(150) == total_rec) and text.splitlines()[1] == want_row
    op.close()

    # secondary line: stats -a (Q20/Q30/gap counters on top), same timing recipe
    op, vec = make_op(True)
    dta, ma, texta, ka_ms, _, _, _ = timed(op, vec, max(3, args.steps // 2), 1)
    stepsa = max(3, args.steps // 2)
    op.close()
    # exact expectation for -a, computed WITHOUT the parser: FASTQ-150 records are 317 bytes with the quality string at
    # columns [166, 316), so Q20 / Q30 are plain column counts over the fixed layout (torch, chunked; untimed)
    q20 = q30 = 0
    view = shard.view(nrec, REC)
    step_rows = 4_000_000
    for r0 in range(0, nrec, step_rows):
        q = view[r0:r0 + step_rows, 166:316]
        q20 += int((q >= 33 + 20).sum().item())
        q30 += int((q >= 33 + 30).sum().item())
        del q
    if dist_on:
        q20, q30 = bdist.all_reduce_count(q20, dev), bdist.all_reduce_count(q30, dev)
    verified_a = (ma.get(150) == total_rec and ma.get(-3) == 0 and ma.get(-1) == q20 and ma.get(-2) == q30
                  and sum(v for k, v in ma.items() if k >= 0) == total_rec)

    # ---- what one GPU does with the shard of an N-GPU job (N = 1 line only): the same step on the first 1/2, 1/4, 1/8 of the
    # file -- the term of the 1 / 2 / 4 / 8 curve that this box CAN measure.  An ESTIMATE of the curve, not the curve: the
    # 512 KB all-reduce per step and the other GPUs are not in it.
    scaling_model = None
    if not dist_on and not args.no_scaling_model:
        op, vec = make_op(False)
        rows = {}
        for g in (1, 2, 4, 8):
            nb_g = (nrec // g) * REC
            if nb_g <= 0:
                continue
            dtg, mg, _, kg, pg, _, _ = timed(op, vec, max(5, args.steps // 2), 2, nb_g)
            step = dtg / max(5, args.steps // 2) * 1e3
            rows[str(g)] = {"shard_GB": round(nb_g / 1e9, 3), "ms_per_step": round(step, 4), "k_stats_ms": round(kg, 4), "k_prep_ms": round(pg, 4),
                            "outside_k_stats_ms": round(step - kg, 4), "k_stats_frac": round(nb_g / (kg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kg > 0 else None,
                            "exact": mg.get(150) == nrec // g}
        op.close()
        if "1" in rows:
            scaling_model = {"what": "ESTIMATE: one GPU running the stats step on the shard an N-GPU job would give it (the first 1/N of the same "
                                     "file); no collective, no second GPU -- not a measured scaling curve",
                             "per_n_gpus": rows,
                             "speedup_if_ranks_do_not_disturb_each_other": {g: round(rows["1"]["ms_per_step"] / r["ms_per_step"], 3) for g, r in rows.items()}}

    # ---- the BASELINE configs that are defined on several GPUs (C3, C5) -- every rank takes part
    ops_multi = None
    if dist_on and not args.no_ops:
        del view
        shard = None
        torch.cuda.empty_cache()
        try:
            ops_multi = run_ops_multi(args, torch, bsk, _lib, lib, check, dev, local, rank, world, bdist, dist.get_backend())
        except Exception as e:  # reported, not hidden; the collectives of the other ranks may then time out -- say which rank
            ops_multi = {"error": "rank %d: %s: %s" % (rank, type(e).__name__, str(e)[:400])}
            if rank != 0:
                print("bench.py rank %d: ops leg failed: %s" % (rank, ops_multi["error"]), file=sys.stderr, flush=True)

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    total_bytes = total_rec * REC
    ms_per_step = dt / args.steps * 1e3
    # HBM bytes per k_stats launch from the PMC passes (rocprofv3 cannot wrap itself: the counters come from the
    # committed profile of this same command, scripts/gpu_round.sh + scripts/pmc_traffic.py); only valid for the
    # full single-GPU workload it was collected on.
    traffic, traffic_src = None, None
    pmc = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if pmc and not dist_on and total_bytes > 99e9:
        try:
            pj = json.load(open(pmc[-1]))
            traffic = pj["per_launch"]["stats"]["traffic_bytes"]
            traffic_src = "profiles/" + os.path.basename(pmc[-1]) + " (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes, gfx950-corrected)"
            # the counters belong to the kernel sources they were collected on: say so when those have changed since
            import hashlib
            h = hashlib.sha256()
            for f in ("stream_stats.hip", "stream_core_dev.hpp", "anchor_wave_dev.hpp"):
                h.update(open(os.path.join(ROOT, "bigseqkit_amd", "csrc", f), "rb").read())
            if pj.get("kernel_sources_sha256") and pj["kernel_sources_sha256"] != h.hexdigest():
                traffic_src += " -- STALE: the kernel sources have changed since that pass"
        except (KeyError, ValueError):
            pass
    out = {
        "metric": "M records/s + GB/s (vs HBM roofline) on 100GB FASTQ stats",
        "value": round(total_rec * args.steps / dt / 1e6, 3),
        "unit": "M records/s",
        "gb_per_s": round(total_bytes * args.steps / dt / 1e9, 2),
        "frac_of_hbm_peak": round(total_bytes * args.steps / dt / 1e9 / (HBM_PEAK_GBS * world), 4),
        "n_gpus": dist.get_world_size() if dist_on else 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "bigseqkit stats on %.1f GB synthetic FASTQ-150 (BASELINE C2: %d records x 317 B), "
                               "HBM-resident, %d record-aligned shard(s)" % (total_bytes / 1e9, total_rec, world),
                   "command": "stats", "records": total_rec, "bytes": total_bytes, "seed": 42},
        "bit_exact_vs_expected_row": bool(verified),
        "shard_bytes_per_rank": nbytes,
        "allreduce_ms_per_step": round(reduce_ms, 4) if dist_on else None,
        "backend": (dist.get_backend() if dist_on else None),
        "per_rank": per_rank,
        "shared_gpu_functional_check": bool(share),
        "single_rank_dist_check": bool(dist_on and world == 1),
        "roofline": {
            "bound": "hbm",
            "kernel": "k_stats<FASTQ,default>",
            "achieved": round(nbytes / (k_ms * 1e-3) / 1e9, 2) if k_ms > 0 else None,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(nbytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else None,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": nbytes,
            "avg_launch_ms": round(k_ms, 4),
            "k_prep_avg_launch_ms": round(p_ms, 4),
        },
        "stats_all": {
            "value": round(total_rec * stepsa / dta / 1e6, 3), "unit": "M records/s",
            "gb_per_s": round(total_bytes * stepsa / dta / 1e9, 2), "ms_per_step": round(dta / stepsa * 1e3, 4),
            "k_stats_avg_launch_ms": round(ka_ms, 4),
            "roofline_frac": round(nbytes / (ka_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ka_ms > 0 else None,
            "verified": bool(verified_a),
            "expected": {"q20": q20, "q30": q30, "gap": 0, "how": "column counts over the fixed 317-byte layout (torch), exact equality"},
        },
    }

    # ---- CPU baseline: the oracle (port of the reference algorithm), 1 thread, bounded sample
    if not dist_on and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle
        pilot = shard[:REC * 200_000].cpu()
        t0 = time.perf_counter()
        oracle.stats_map_ptr(pilot.data_ptr(), pilot.numel(), True, "{}")
        rate = pilot.numel() / (time.perf_counter() - t0)  # bytes/s
        srec = int(min(8e9, nbytes)) // REC       # host sample: first <= 8 GB of the file
        sample = shard[:REC * srec].cpu()
        passes = max(1, int(round(args.cpu_seconds * rate / (srec * REC))))
        t0 = time.perf_counter()
        for _ in range(passes):
            cm = oracle.stats_map_ptr(sample.data_ptr(), sample.numel(), True, "{}")
        ct = time.perf_counter() - t0
        assert cm.get(150) == srec
        out["cpu_baseline"] = {
            "value": round(srec * passes / ct / 1e6, 4), "unit": "M records/s",
            "gb_per_s": round(srec * passes * REC / ct / 1e9, 4), "cores": 1, "kind": "port",
            "sample": "oracle/ (C++ restatement of ReadFixer+SeqParser+Stats, NOT IgnisHPC/Go): %d pass(es) over the "
                      "first %d records (%.2f GB) of the same file, %.1f s, 1 thread of %d host cores"
                      % (passes, srec, srec * REC / 1e9, ct, os.cpu_count()),
        }
        # the same port on every host core (one thread per record-aligned slice of the sample; ctypes releases the
        # GIL): the fairer "what would the CPU path do on this box" number, still a reported baseline only
        try:
            from concurrent.futures import ThreadPoolExecutor
            nthr = max(1, min(os.cpu_count() or 1, 256))
            per = srec // nthr
            if per > 0:
                base = sample.data_ptr()
                def work(k):
                    lo = k * per
                    cnt = per if k + 1 < nthr else srec - lo
                    m = oracle.stats_map_ptr(base + lo * REC, cnt * REC, True, "{}")
                    return m.get(150, 0)
                reps = 8
                with ThreadPoolExecutor(nthr) as ex:
                    list(ex.map(work, range(nthr)))  # warm-up: threads started, pages touched
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        got = sum(ex.map(work, range(nthr)))
                    ct = (time.perf_counter() - t0) / reps
                assert got == srec
                out["cpu_baseline_all_cores"] = {
                    "value": round(srec / ct / 1e6, 2), "unit": "M records/s", "gb_per_s": round(srec * REC / ct / 1e9, 2),
                    "cores": nthr, "kind": "port",
                    "sample": "the same oracle, %d threads, mean of %d passes over the same %.2f GB sample, %.2f s per pass" % (nthr, reps, srec * REC / 1e9, ct),
                }
        except Exception as e:  # never let the extra baseline break the bench line
            out["cpu_baseline_all_cores"] = {"error": str(e)[:200]}
    # ---- the other BASELINE configs (N = 1): seq -n @ C2, grep @ C3 shard, translate @ C4, rmdup @ C5 shard
    if not dist_on and not args.no_ops:
        if total_bytes > 99e9 or args.ops_scale != 1.0:
            try:
                out["ops"] = run_ops(args, torch, bsk, _lib, lib, check, dev, local, shard, total_rec)
            except Exception as e:  # the headline line must survive a failure here; the failure is reported, not hidden
                out["ops"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
            try:
                shard = None
                torch.cuda.empty_cache()
                out["end_to_end"] = run_end_to_end(_Helpers(args, torch, bsk, _lib, lib, check, dev, local), 8.0 * min(1.0, args.ops_scale * 4))
            except Exception as e:
                out["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
            if not args.no_e2e_full:
                try:
                    out["end_to_end_config_size"] = file_to_result_config_size(_Helpers(args, torch, bsk, _lib, lib, check, dev, local),
                                                                               (total_bytes / 1e9) if args.ops_scale == 1.0 else 100.0 * args.ops_scale)
                except Exception as e:
                    out["end_to_end_config_size"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
            try:
                out["vendor_yardsticks"] = vendor_yardsticks(torch, dev, 20.0 * min(1.0, args.ops_scale * 4))
            except Exception as e:
                out["vendor_yardsticks"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        else:
            out["ops"] = {"skipped": "the 'ops' workloads are defined at the full BASELINE sizes (--gb 100) or with --ops-scale"}
    if scaling_model is not None:
        out["scaling_model"] = scaling_model
    if ops_multi is not None:
        out["ops"] = ops_multi
    print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
