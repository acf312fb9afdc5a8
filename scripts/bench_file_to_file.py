#!/usr/bin/env python3
"""End-to-end (host buffer -> output file) rate of bsk_run_to_store: a partition in PINNED host memory goes through the
chunked pipeline (H2D of chunk i+1 || kernels of chunk i || D2H + write of chunk i-1) into a FileStore.  The PCIe-inclusive
numbers next to the HBM-resident ones of bench_ops.py; they are never `value` of bench.py.  Output goes to /dev/shm (page
cache speed: shows the pipeline) and, with --disk, to a file under /tmp as well.
Usage: python scripts/bench_file_to_file.py [GB of FASTQ, default 8] [--disk]"""
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

gb = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 8.0
disk = "--disk" in sys.argv
lib.bsk_host_alloc.restype = C.c_void_p


def pinned_copy_of(kind, flags, nbytes):
    rb = lib.bsk_synth_record_bytes(kind)
    n = int(nbytes) // rb * rb
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    check(lib.bsk_synth_device(kind, 42, flags, 0, C.c_void_p(t.data_ptr()), n, 0, None))
    torch.cuda.synchronize()
    h = lib.bsk_host_alloc(n)
    assert h, "pinned allocation failed"
    check(lib.bsk_device_copy(C.c_void_p(h), C.c_void_p(t.data_ptr()), n, 2))
    del t
    torch.cuda.empty_cache()
    return h, n, n // rb


def run(op_name, opts, h, n, fmt, path):
    s = C.c_void_p()
    assert lib.bsk_store_open(path.encode(), 1, C.byref(s)) == 0
    nb, nr = C.c_uint64(), C.c_uint64()
    with bsk.Operator(op_name, json.dumps(opts), 0) as op:
        t0 = time.perf_counter()
        check(lib.bsk_run_to_store(op.ctx, C.c_void_p(h), n, fmt, 0, s, 0, C.byref(nb), C.byref(nr)), op.ctx)
        dt = time.perf_counter() - t0
    tot = C.c_uint64()
    assert lib.bsk_store_close(s, C.byref(tot)) == 0
    os.unlink(path)
    return dt, nb.value, nr.value


res = {"note": "bsk_run_to_store from pinned host memory into one file; wall clock of the call (H2D + kernels + D2H + write); "
               "GB/s = input bytes / time; second call of each command (buffers allocated)", "rows": {}}
targets = [("/dev/shm/bsk_f2f.out", "page cache (/dev/shm)")] + ([("/tmp/bsk_f2f.out", "disk (/tmp)")] if disk else [])
h, n, nrec = pinned_copy_of(_lib.SYNTH_FASTQ150, _lib.SYNTH_FLAG_MOTIF, gb * 1e9)
for name, op, opts in (("seq -n", "SeqTransform", {"Name": True}),
                       ("grep -s -p 12-mer", "Grep", {"BySeq": True, "Pattern": ["ACGTTGCAAGCT"]}),
                       ("seq -r -p (full re-emit)", "SeqTransform", {"Reverse": True, "Complement": True})):
    for path, where in targets:
        run(op, opts, h, n, 1, path)
        dt, ob, orec = run(op, opts, h, n, 1, path)
        res["rows"]["%s -> %s" % (name, where)] = {"in_GB": round(n / 1e9, 2), "out_GB": round(ob / 1e9, 3), "s": round(dt, 3),
                                                    "in_GB_per_s": round(n / dt / 1e9, 2), "M_records_per_s": round(nrec / dt / 1e6, 1)}
lib.bsk_host_free(C.c_void_p(h))
h, n, nrec = pinned_copy_of(_lib.SYNTH_FASTA5K_CDS, 0, gb * 0.5e9)
for path, where in targets:
    run("Translate", {"Frame": ["6"]}, h, n, 0, path)
    dt, ob, orec = run("Translate", {"Frame": ["6"]}, h, n, 0, path)
    res["rows"]["translate -f 6 -> %s" % where] = {"in_GB": round(n / 1e9, 2), "out_GB": round(ob / 1e9, 3), "s": round(dt, 3),
                                                   "in_GB_per_s": round(n / dt / 1e9, 2), "in_plus_out_GB_per_s": round((n + ob) / dt / 1e9, 2)}
lib.bsk_host_free(C.c_void_p(h))

# ---- a directory of part files (StoreFASTXN, the reference's default layout) written by several contexts at once: ONE file
# takes 6 - 10 GB/s whatever is done (scripts/experiments/write_rates.cpp), a file per writer scales
if "--parts" in sys.argv:
    from concurrent.futures import ThreadPoolExecutor
    from bigseqkit_amd import dist as bdist
    import shutil
    h, n, nrec = pinned_copy_of(_lib.SYNTH_FASTQ150, 0, gb * 1e9)
    rb = 317
    for P in (1, 4, 8, 16):
        per = nrec // P
        bounds = [(k * per * rb, ((k + 1) * per if k + 1 < P else nrec) * rb) for k in range(P)]
        for where, base in ([("page cache (/dev/shm)", "/dev/shm/bsk_f2f_parts")] + ([("disk (/tmp)", "/tmp/bsk_f2f_parts")] if disk else [])):
            best = None
            for rep in range(2):
                shutil.rmtree(base, ignore_errors=True)
                s = C.c_void_p()
                assert lib.bsk_store_open(base.encode(), 0, C.byref(s)) == 0
                ops = [bsk.Operator("SeqTransform", json.dumps({"Reverse": True, "Complement": True, "Config": {"SeqType": "dna", "Quiet": True}}), 0) for _ in range(P)]

                def work(k):
                    lo, hi = bounds[k]
                    nb, nr = C.c_uint64(), C.c_uint64()
                    check(lib.bsk_run_to_store(ops[k].ctx, C.c_void_p(h + lo), hi - lo, 1, k, s, k, C.byref(nb), C.byref(nr)), ops[k].ctx)
                    return nb.value
                t0 = time.perf_counter()
                with ThreadPoolExecutor(P) as ex:
                    outb = sum(ex.map(work, range(P)))
                dt = time.perf_counter() - t0
                tot = C.c_uint64()
                assert lib.bsk_store_close(s, C.byref(tot)) == 0 and tot.value == outb
                for o in ops:
                    o.close()
                best = dt if best is None else min(best, dt)
            shutil.rmtree(base, ignore_errors=True)
            res["rows"]["seq -r -p, %d partitions -> %d part files at once -> %s" % (P, P, where)] = {
                "in_GB": round(n / 1e9, 2), "out_GB": round(outb / 1e9, 2), "s": round(best, 3),
                "in_GB_per_s": round(n / best / 1e9, 2), "in_plus_out_GB_per_s": round((n + outb) / best / 1e9, 2)}
    lib.bsk_host_free(C.c_void_p(h))
print(json.dumps(res, indent=1))
