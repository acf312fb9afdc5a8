"""`python -m bigseqkit_amd.run --devices 0,1,... -- <command> [flags] file` (or `bigseqkit <command> ... --devices 0-7`):
the seven hot-path commands on SEVERAL GPUs of one node -- one worker process per device, as the reference runs one
executor per partition set (/root/reference/bigseqkit/helper.go:148-195 ReadFASTA/Q[N] + StoreFASTX[N];
bigseqkit-cli/helper.go:87-141 ignisDriver: read -> command -> union -> store).

  * the FILE is cut, not copied: every worker maps it, looks for the first record start in a 1 MiB window behind its nominal
    cut (bsk_find_record_start: the ReadFixer rule) and reads only its own byte range into pinned host memory, on CPUs of its
    GPU's NUMA node;
  * seq / grep / locate / subseq / translate / fq2fa: bsk_run_to_store on the shard (H2D || kernels || D2H + write) into
    `<out>/part%05d` files, one per worker (StoreFASTXN), or -- with --merge -- into ONE file: the workers learn their
    offsets from an all_gather of their sizes (the reference passes an MPI token, bigseqkit-lib/helper.go:399-429);
  * stats: one sum all-reduce of the device stats vector (StatsReduce, bigseqkit/stats.go:91), rank 0 prints the table;
  * grep -C: one sum all-reduce of the counts (GrepReduceCount, bigseqkit/grep.go:175);
  * rmdup: the 24-byte tuple exchange of dist.rmdup_distributed (GroupByKey, bigseqkit/rmdup.go:97); survivors stay in HBM
    until the store drains them.
The flags are the CLI's own: the launcher asks `bigseqkit ... --plan` for the operator name, the option JSON and the
output place, so there is no second flag parser.  Collectives run over RCCL ("nccl") with a GPU per worker; `--share-gpu`
(tests, or more workers than GPUs) lets workers share devices and reduce over gloo.
"""
import ctypes as C
import json
import mmap
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CLI = os.path.join(HERE, "bin", "bigseqkit")
STREAMED = {"seq", "grep", "locate", "subseq", "translate", "fq2fa"}
SUPPORTED = STREAMED | {"stats", "rmdup"}


def parse_devices(text):
    """"0,1,2" / "0-3" / "0-1,4" -> [0, 1, ...]"""
    out = []
    for piece in text.split(","):
        piece = piece.strip()
        if not piece:
            continue
        if "-" in piece:
            a, b = piece.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(piece))
    if not out:
        raise SystemExit("bigseqkit_amd.run: --devices names no device")
    return out


def plan_of(cli_args):
    p = subprocess.run([CLI] + list(cli_args) + ["--plan"], capture_output=True, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stderr)
        raise SystemExit(p.returncode or 1)
    return json.loads(p.stdout.strip().splitlines()[-1])


def sniff_format(path, first_byte):
    """bigseqkit-cli/helper.go:63-78: the extension, then the first byte"""
    low = path.lower()
    if low.endswith((".fa", ".fna", ".ffn", ".faa", ".frn")):
        return 0
    if low.endswith((".fq", ".fastq")):
        return 1
    if first_byte == b">":
        return 0
    if first_byte == b"@":
        return 1
    raise SystemExit(" <file> must be fasta or fastq")


def cut_points(mm, size, world, fmt, lib, check):
    """world + 1 offsets; cut k = the first record start at or behind size * k / world, searched in a window of the mapped
    file (1 MiB either side, grown until the next larger window names the same start): nothing but those windows is touched"""
    import numpy as np
    view = np.frombuffer(mm, dtype=np.uint8)
    base = view.ctypes.data
    cuts = [0]
    for k in range(1, world):
        nominal = size * k // world
        lo = max(nominal, cuts[-1])
        if lo >= size:
            cuts.append(size)
            continue
        win = 1 << 20

        def look(w):
            # up to `w` bytes before `lo` (a FASTQ start is also judged by the record that ends there) and `w` behind it
            a = lo - w if lo > w else 0
            b = min(size, lo + w)
            out = C.c_size_t()
            check(lib.bsk_find_record_start(C.c_void_p(base + a), b - a, lo - a, fmt, C.byref(out)))
            return a + out.value, b
        while True:
            found, b = look(win)
            # a start close to the window's end was judged on a cut-off record, and a candidate passed over because the
            # window cut its record off shows up as a different answer of the next larger window: look again with more text
            if b < size and (found + (64 << 10) > b or look(win * 4)[0] != found):
                win *= 4
                continue
            cuts.append(min(found, size))
            break
    cuts.append(size)
    return cuts


def numa_cpus_of_gpu(torch, device):
    """CPUs of the NUMA node the GPU hangs on (None when the platform does not say)"""
    try:
        p = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return None
        cpus = set()
        for piece in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            if "-" in piece:
                a, b = piece.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif piece:
                cpus.add(int(piece))
        return cpus or None
    except Exception:
        return None


def worker(devices, share, cli_args):
    import torch
    import torch.distributed as dist
    import bigseqkit_amd as bsk
    from bigseqkit_amd import _lib, dist as bdist
    from bigseqkit_amd._lib import lib, check

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    plan = plan_of(cli_args)
    use = plan["use"]
    if use not in SUPPORTED:
        raise SystemExit("bigseqkit_amd.run: '%s' runs on one device (bigseqkit %s ... --device N); several GPUs: %s"
                         % (use, use, ", ".join(sorted(SUPPORTED))))
    if len(plan["files"]) != 1:
        raise SystemExit("bigseqkit_amd.run: exactly one input file (it is cut into one shard per GPU)")
    path = plan["files"][0]
    opts = plan["opts"]
    if lib.bsk_device_count() <= 0 or not torch.cuda.is_available():
        raise SystemExit("bigseqkit_amd.run: no HIP device visible (the hot path has no CPU fallback)")
    ngpu = torch.cuda.device_count()
    device = devices[rank % len(devices)]
    backend = "nccl"
    if share or len(set(devices)) < world or device >= ngpu:
        backend = "gloo"                  # RCCL refuses two ranks on one device
        device = device % ngpu
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    cpus = numa_cpus_of_gpu(torch, device)
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)  # the reader / writer threads of this worker stay next to its GPU
        except OSError:
            pass
    # BSK_DIST_SINGLE_RANK_COLLECTIVES=1: ONE worker still forms a process group and runs every collective of the N-worker
    # path (dist._active) -- how a one-GPU box executes the RCCL calls of this entry point at all (tests/test_run_multi_gpu.py)
    grouped = world > 1 or os.environ.get("BSK_DIST_SINGLE_RANK_COLLECTIVES") == "1"
    if grouped:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (one worker started by hand: any free port)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        # The collective libraries greet on file descriptor 1 -- "[Gloo] Rank 0 is connected to ..." while the group forms,
        # RCCL's version banner when its communicator is first used (after a `stats` table had been printed: found by
        # running one worker over RCCL, tests/test_run_multi_gpu.py) -- and stdout is where `stats`, `grep -C` and `-o -`
        # put their RESULT.  For the life of a grouped worker descriptor 1 IS its stderr; results go to the real stdout
        # through sys.stdout, which is re-opened on a duplicate of it.
        sys.stdout.flush()
        real_out = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(real_out, "w")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=30))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
        bdist.barrier()
    try:
        return _work(plan, use, opts, path, world, rank, device, dev, torch, bsk, _lib, bdist, lib, check)
    finally:
        if grouped:
            sys.stdout.flush()
            dist.destroy_process_group()


def _work(plan, use, opts, path, world, rank, device, dev, torch, bsk, _lib, bdist, lib, check):
    size = os.path.getsize(path)
    fd = os.open(path, os.O_RDONLY)
    try:
        mm = mmap.mmap(fd, 0, access=mmap.ACCESS_READ) if size else None
        fmt = sniff_format(path, mm[:1] if size else b"")
        cuts = cut_points(mm, size, world, fmt, lib, check) if size else [0] * (world + 1)
        lo, hi = cuts[rank], cuts[rank + 1]
        n = hi - lo
        # this worker's bytes, and only they, into pinned host memory
        h = lib.bsk_host_alloc(max(1, n))
        if not h:
            raise SystemExit("bigseqkit_amd.run: pinned allocation of %d bytes failed" % n)
        buf = memoryview((C.c_ubyte * max(1, n)).from_address(h)).cast("B")
        done = 0
        while done < n:
            got = os.preadv(fd, [buf[done:min(n, done + (256 << 20))]], lo + done)
            if got <= 0:
                raise SystemExit("bigseqkit_amd.run: short read of %s" % path)
            done += got
        if mm is not None:
            mm.close()
    finally:
        os.close(fd)
    ojs = json.dumps(opts)
    out_file = plan["out_file"] or (path + "-out")
    merge = bool(plan["merge"])
    try:
        if use == "stats":
            return _stats(h, n, fmt, ojs, rank, world, device, dev, torch, bsk, bdist, lib, check)
        if use == "grep" and opts.get("Count"):
            with bsk.Operator("Grep", ojs, device) as op:
                out = _lib.Out()
                check(lib.bsk_grep_run(op.ctx, C.c_void_p(h), n, 0, fmt, rank, None, C.byref(out)), op.ctx)
                cnt = C.c_uint64()
                check(lib.bsk_grep_last_count(op.ctx, C.byref(cnt)), op.ctx)
            total = bdist.all_reduce_count(cnt.value, dev)
            if rank == 0:
                sys.stdout.write(str(total))   # fmt.Print: no newline (bigseqkit-cli/grep.go:14)
                sys.stdout.flush()
            return 0
        return _records(use, plan["op"], h, n, fmt, ojs, out_file, merge, rank, world, device, dev, torch, bsk, _lib, bdist, lib, check)
    finally:
        lib.bsk_host_free(C.c_void_p(h))


def _stats(h, n, fmt, ojs, rank, world, device, dev, torch, bsk, bdist, lib, check):
    with bsk.Operator("Stats", ojs, device) as op:
        vec = torch.zeros(lib.bsk_stats_vector_len(op.ctx), dtype=torch.int64, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.bsk_stats_reset(op.ctx, st), op.ctx)
        if n:
            check(lib.bsk_stats_run(op.ctx, C.c_void_p(h), n, 0, fmt, rank, C.c_void_p(vec.data_ptr()), st), op.ctx)
        m = bdist.collect_reduced(op, vec)          # StatsReduce + the driver's collect (+ overflow lists when needed)
        if rank == 0:
            info = bsk.api._finalize(op, m)
            buf = C.create_string_buffer(1 << 16)
            check(lib.bsk_stats_string(op.ctx, b"input0", b"N/A", C.byref(info), buf, len(buf)), op.ctx)
            table = buf.value.decode()
            nl = table.find("\n")
            sys.stdout.write(table[:nl + 1] + table[nl + 1:] + "\n")   # (head + Join(lines[1:]) + "\n", as the CLI)
            sys.stdout.flush()
    return 0


def _records(use, op_name, h, n, fmt, ojs, out_file, merge, rank, world, device, dev, torch, bsk, _lib, bdist, lib, check):
    """the record commands: every worker's output is part `rank`; --merge: one file, the parts at the offsets of a scan"""
    to_stdout = out_file == "-"
    if to_stdout or merge:
        # this worker's part goes to a spool file of its own first.  Its name holds a token that only this JOB knows (the
        # rendezvous port) and it is created with O_EXCL: two jobs started from one shell no longer share a spool, and a
        # planted link is not followed (the name used to be /tmp/bsk-stdout-<ppid>...)
        import tempfile
        token = "%s-%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
        base = os.path.join(tempfile.gettempdir(), "bsk-stdout") if to_stdout else out_file
        target = "%s.bsk-%s-part%05d.tmp" % (base, token, rank)
        try:
            os.close(os.open(target, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o600))
        except OSError:
            raise SystemExit("bigseqkit_amd.run: cannot create " + target)
        st = C.c_void_p()
        if lib.bsk_store_open(target.encode(), 1, C.byref(st)) != 0:
            raise SystemExit("bigseqkit_amd.run: cannot create " + target)
        part = 0
    else:
        if rank == 0:
            os.makedirs(out_file, exist_ok=True)
            # (part files of an earlier run with MORE workers would be read as part of this result)
            k = world
            while os.path.exists(os.path.join(out_file, "part%05d" % k)):
                os.unlink(os.path.join(out_file, "part%05d" % k))
                k += 1
        bdist.barrier()
        st = C.c_void_p()
        if lib.bsk_store_open(out_file.encode(), 0, C.byref(st)) != 0:
            raise SystemExit("bigseqkit_amd.run: cannot open the directory " + out_file)
        part = rank
    nb, nr = C.c_uint64(), C.c_uint64()
    try:
        if use == "rmdup":
            be = bdist.HipRmDupBackend(ojs, device)
            try:
                shard = torch.empty(max(1, n), dtype=torch.uint8, device=dev)[:n]
                if n:
                    check(lib.bsk_device_copy(C.c_void_p(shard.data_ptr()), C.c_void_p(h), n, 1))
                text = bdist.rmdup_distributed(shard, fmt, be, to_host=False)
                o = _lib.Out(text.ptr, text.len, text.records)
                check(lib.bsk_store_put(st, be.op.ctx, part, C.byref(o)), be.op.ctx)
                nb.value, nr.value = text.len, text.records
            finally:
                be.close()
        else:
            with bsk.Operator(op_name, ojs, device) as op:
                check(lib.bsk_run_to_store(op.ctx, C.c_void_p(h), n, fmt, rank, st, part, C.byref(nb), C.byref(nr)), op.ctx)
    finally:
        tot = C.c_uint64()
        if lib.bsk_store_close(st, C.byref(tot)) != 0:
            raise SystemExit("bigseqkit_amd.run: closing the output failed")
    if not (to_stdout or merge):
        if nb.value == 0 and world > 1:        # (an empty part file still marks the partition, as SaveAsTextFile does)
            open(os.path.join(out_file, "part%05d" % rank), "ab").close()
        return 0
    # ---- one file: offsets from an all_gather of the sizes, every worker places its own part (FileStore's order)
    sizes, _ = bdist._all_gather_int(nb.value, dev)   # (one element without a group)
    off, total = sum(sizes[:rank]), sum(sizes)
    if to_stdout:
        for r in range(world):               # in turn
            if r == rank:
                with open(target, "rb") as f:
                    while True:
                        b = f.read(64 << 20)
                        if not b:
                            break
                        sys.stdout.buffer.write(b)
                sys.stdout.buffer.flush()
            bdist.barrier()
        os.unlink(target)
        return 0
    if rank == 0:
        with open(out_file, "wb") as f:
            f.truncate(total)
    bdist.barrier()
    src = os.open(target, os.O_RDONLY)
    dst = os.open(out_file, os.O_WRONLY)
    try:
        done = 0
        while done < nb.value:
            try:
                k = os.copy_file_range(src, dst, min(nb.value - done, 1 << 30), done, off + done)
            except (OSError, AttributeError):
                b = os.pread(src, min(nb.value - done, 64 << 20), done)
                k = os.pwrite(dst, b, off + done)
            if k <= 0:
                raise SystemExit("bigseqkit_amd.run: short copy into " + out_file)
            done += k
    finally:
        os.close(src)
        os.close(dst)
    os.unlink(target)
    bdist.barrier()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    devices, share, cli_args = None, False, []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "--":
            cli_args += argv[i + 1:]
            break
        if a == "--devices" and i + 1 < len(argv):
            devices = parse_devices(argv[i + 1])
            i += 2
            continue
        if a.startswith("--devices="):
            devices = parse_devices(a.split("=", 1)[1])
            i += 1
            continue
        if a == "--share-gpu":
            share = True
            i += 1
            continue
        cli_args.append(a)
        i += 1
    if devices is None:
        devices = [0]
    if os.environ.get("BSK_RUN_SHARE_GPU") == "1":
        share = True
    world = len(devices)
    if "WORLD_SIZE" not in os.environ and world > 1:
        plan_of(cli_args)   # flag errors once, before any worker starts
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(port), "-m", "bigseqkit_amd.run", "--devices", ",".join(map(str, devices))]
        if share:
            cmd.append("--share-gpu")
        cmd += ["--"] + cli_args
        os.execv(sys.executable, cmd)
    return worker(devices, share, cli_args)


if __name__ == "__main__":
    sys.exit(main())
