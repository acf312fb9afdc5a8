#!/usr/bin/env python3
"""FILE -> device at the size of C2: `bigseqkit stats -T <100 GB file in /dev/shm> --devices 0`, a fresh process per run, by
number of readers and piece size of bsk_shard_load (BSK_SHARD_READERS, BSK_SHARD_PIECE_BYTES).  Usage: r06_load_sweep.py [GB]"""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
from concurrent.futures import ThreadPoolExecutor

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
REC = 317
n = int(gb * 1e9) // REC * REC
src = "/dev/shm/bsk_sweep_%d.fastq" % os.getpid()
cli = os.path.join(ROOT, "bigseqkit_amd", "bin", "bigseqkit")
piece = (2 << 30) // REC * REC
hbuf = lib.bsk_host_alloc(piece)
view = memoryview((C.c_char * piece).from_address(hbuf)).cast("B")
dpiece = torch.empty(piece, dtype=torch.uint8, device="cuda")
fd = os.open(src, os.O_CREAT | os.O_TRUNC | os.O_WRONLY, 0o600)
res = {}
try:
    at = 0
    with ThreadPoolExecutor(8) as ex:
        while at < n:
            ln = min(piece, n - at)
            assert lib.bsk_synth_device(0, 42, 0, at // REC, C.c_void_p(dpiece.data_ptr()), ln, 0, None) == 0
            torch.cuda.synchronize()
            check(lib.bsk_device_copy(C.c_void_p(hbuf), C.c_void_p(dpiece.data_ptr()), ln, 2))
            step = (ln + 7) // 8
            list(ex.map(lambda k: os.pwrite(fd, view[k * step:min(ln, (k + 1) * step)], at + k * step) if k * step < ln else 0, range(8)))
            at += ln
    os.close(fd)
    del dpiece
    torch.cuda.empty_cache()
    settings = ((24, 16 << 20), (8, 16 << 20), (16, 16 << 20), (32, 16 << 20), (48, 16 << 20), (64, 16 << 20), (24, 64 << 20), (48, 64 << 20),
                (24, 4 << 20), (24, 16 << 20))
    reps = 2
    if os.environ.get("SWEEP_TIMING"):   # where a slow run spends its time: the same two settings four times, BSK_CLI_TIMING lines
        settings, reps = tuple((int(r), int(m) << 20) for r, m in (x.split("x") for x in os.environ["SWEEP_TIMING"].split(","))), 3
        os.environ["BSK_SHARD_TIMING"] = "1"
    for readers, pbytes in settings:
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            p = subprocess.run([cli, "stats", "-T", src, "--devices", "0"], capture_output=True, timeout=900,
                               env=dict(os.environ, BSK_SHARD_READERS=str(readers), BSK_SHARD_PIECE_BYTES=str(pbytes)))
            ts.append(round(time.perf_counter() - t0, 3))
            assert p.returncode == 0, p.stderr[-300:]
            if os.environ.get("SWEEP_TIMING"):
                print("   ", ts[-1], p.stderr.decode(errors="replace").replace("\n", " | ")[-700:], flush=True)
        res["%d readers x %d MiB" % (readers, pbytes >> 20)] = {"s": ts, "GB_per_s": round(n / 1e9 / min(ts), 1)}
        print(readers, pbytes >> 20, ts, flush=True)
finally:
    lib.bsk_host_free(C.c_void_p(hbuf))
    if os.path.exists(src):
        os.unlink(src)
print(json.dumps(res))
