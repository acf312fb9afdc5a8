"""Host-side mirror of the reference's driver library for the hot path.

Same entry points, argument meaning and error behaviour as
/root/reference/bigseqkit/{helper,stats}.go, but the dataflow
``MapPartitions(libSource("Stats")) -> Reduce(libSource("StatsReduce"))`` is a
sequence of C-ABI calls into libbsk.so (include/bsk.h) instead of IgnisHPC tasks.
"""
import ctypes as C
import os

from . import _lib
from ._lib import lib, check, BskError, FORMAT_FASTA, FORMAT_FASTQ
from .options import SeqKitStatsOptions, SeqKitSeqOptions, SeqKitGrepOptions, SeqKitSubseqOptions, SeqKitTranslateOptions, SeqKitRmDupOptions, SeqKitLocateOptions, SeqKitFq2FaOptions, SeqKitHeadOptions, SeqKitDuplicateOptions, SeqKitRenameOptions, SeqKitSortOptions, SeqKitFaidxOptions, SeqKitPairOptions, SeqKitCommonOptions, SeqKitConcatOptions


class SeqFrame:
    """What ``*api.IDataFrame[string]`` is to the reference: the records of one input,
    held as shards (partitions) of raw FASTA/FASTQ text that begin on a record.
    A shard is ``(buffer, nbytes, on_device)``; device buffers are anything with
    ``data_ptr()`` (a torch uint8 tensor), host buffers are bytes-like."""

    def __init__(self, fmt, shards):
        self.format = fmt
        self.shards = list(shards)

    @staticmethod
    def _ptr(buf):
        if hasattr(buf, "data_ptr"):
            return C.c_void_p(buf.data_ptr()), int(buf.numel()), bool(buf.is_cuda)
        mv = memoryview(buf)
        if mv.readonly:
            arr = (C.c_char * len(mv)).from_buffer_copy(mv)  # keeps a private copy alive
        else:
            arr = (C.c_char * len(mv)).from_buffer(mv)
        return C.cast(arr, C.c_void_p), len(mv), False, arr

    def partitions(self):
        for pid, buf in enumerate(self.shards):
            r = self._ptr(buf)
            yield pid, r[0], r[1], r[2], r  # keep r alive while the pointer is in use


def _read(path_or_bytes, fmt, min_partitions=1):
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) or hasattr(path_or_bytes, "data_ptr"):
        data = path_or_bytes
    else:
        with open(os.fspath(path_or_bytes), "rb") as f:
            data = f.read()
    if hasattr(data, "data_ptr") or min_partitions <= 1:
        return SeqFrame(fmt, [data])
    # PlainFileN(path, minPartitions, delim) + ReadFixer: cut on record starts
    n = len(data)
    arr = (C.c_char * n).from_buffer_copy(data)
    cuts = [0]
    for k in range(1, min_partitions):
        out = C.c_size_t()
        check(lib.bsk_find_record_start(C.cast(arr, C.c_void_p), n, n * k // min_partitions, fmt, C.byref(out)))
        if out.value > cuts[-1]:
            cuts.append(out.value)
    cuts.append(n)
    return SeqFrame(fmt, [bytes(data[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a])


def ReadFASTA(path, worker=None):
    """bigseqkit/helper.go:148-154"""
    return _read(path, FORMAT_FASTA)


def ReadFASTAN(path, minPartitions, worker=None):
    """bigseqkit/helper.go:156-162"""
    return _read(path, FORMAT_FASTA, minPartitions)


def ReadFASTQ(path, worker=None):
    """bigseqkit/helper.go:164-170"""
    return _read(path, FORMAT_FASTQ)


def ReadFASTQN(path, minPartitions, worker=None):
    """bigseqkit/helper.go:172-178"""
    return _read(path, FORMAT_FASTQ, minPartitions)


class Operator:
    """One plugin operator between Before() and After() (bsk_create .. bsk_destroy)."""

    def __init__(self, name, opts_json, device=0):
        self.ctx = C.c_void_p()
        if isinstance(opts_json, str):
            opts_json = opts_json.encode()
        check(lib.bsk_create(name.encode(), opts_json, device, C.byref(self.ctx)))

    def close(self):
        if self.ctx:
            lib.bsk_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def opts_json(self):
        return lib.bsk_opts_json(self.ctx).decode()


def _collect_map(op, d_vec=None):
    cap = 1024
    while True:
        n = C.c_size_t()
        keys = (C.c_int64 * cap)()
        vals = (C.c_int64 * cap)()
        rc = lib.bsk_stats_collect(op.ctx, d_vec, keys, vals, cap, C.byref(n))
        if rc == _lib.BSK_ERR_CAPACITY and n.value > cap:
            cap = n.value
            continue
        check(rc, op.ctx)
        return dict(zip(keys[:n.value], vals[:n.value]))


def stats_map(input, o=None, device=0, stream=None):
    """MapPartitions(Stats) + Reduce(StatsReduce) (bigseqkit/stats.go:81-94): the merged
    map[int64]int64.  Returns (map, operator) -- the operator carries the first record
    the driver needs for the type column."""
    o = o or SeqKitStatsOptions()
    op = Operator("Stats", o.to_json(), device)
    try:
        for pid, ptr, n, on_dev, keep in input.partitions():
            check(lib.bsk_stats_run(op.ctx, ptr, n, 1 if on_dev else 0, input.format, pid, None, stream), op.ctx)
        return _collect_map(op), op
    except Exception:
        op.close()
        raise


def _finalize(op, m):
    ks = sorted(m)
    keys = (C.c_int64 * len(ks))(*ks)
    vals = (C.c_int64 * len(ks))(*[m[k] for k in ks])
    info = _lib.StatInfo()
    check(lib.bsk_stats_finalize(op.ctx, keys, vals, len(ks), C.byref(info)), op.ctx)
    return info


def Stats(name, format, input, o=None, device=0):
    """bigseqkit/stats.go:75-166 -> StatInfo"""
    m, op = stats_map(input, o, device)
    with op:
        return _finalize(op, m)


def StatsString(name, format, input, o=None, device=0):
    """bigseqkit/stats.go:168-288"""
    m, op = stats_map(input, o, device)
    with op:
        info = _finalize(op, m)
        buf = C.create_string_buffer(1 << 16)
        check(lib.bsk_stats_string(op.ctx, name.encode(), format.encode(), C.byref(info), buf, len(buf)), op.ctx)
        return buf.value.decode()


def _run_records(op_name, run_fn, input, o, device=0, stream=None, finish=None):
    """MapPartitions(libSource(op_name)) over the shards of `input`: the concatenated
    FileStore bytes (element + newline per output record) and the number of elements."""
    chunks, nrec = [], 0
    with Operator(op_name, o.to_json(), device) as op:
        for pid, ptr, n, on_dev, keep in input.partitions():
            out = _lib.Out()
            check(run_fn(op.ctx, ptr, n, 1 if on_dev else 0, input.format, pid, stream, C.byref(out)), op.ctx)
            buf = C.create_string_buffer(max(1, out.len))
            check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
            chunks.append(buf.raw[:out.len])
            nrec += out.records
        if finish is not None:  # After() with an error return
            check(finish(op.ctx), op.ctx)
    return b"".join(chunks), nrec


def _one_shard(input):
    """The operators whose result depends on ALL records (rmdup, rename, sort, grep --delete-matched: GroupByKey /
    SortByKey / Reduce over the whole dataframe in the reference) see the input as ONE shard: the shards of `input`
    back to back, a newline added where a shard lacks the final one (what cli/bigseqkit.cpp does with several input
    files).  Across ranks the exchange of dist.py takes this place."""
    if len(input.shards) <= 1:
        return input
    if all(hasattr(b, "data_ptr") for b in input.shards):
        both, _ = _join_files(list(input.shards))
        return SeqFrame(input.format, [both])
    if any(hasattr(b, "data_ptr") for b in input.shards):
        raise ValueError("a SeqFrame must hold either host or device shards, not both")
    parts = []
    for b in input.shards:
        b = bytes(b)
        parts.append(b if (not b or b.endswith(b"\n")) else b + b"\n")
    return SeqFrame(input.format, [b"".join(parts)])


def Seq(input, o=None, device=0):
    """bigseqkit/seq.go:157-170 -- returns the bytes StoreFASTX would write"""
    return _run_records("SeqTransform", lib.bsk_seq_run, input, o or SeqKitSeqOptions(), device)[0]


def Grep(input, o, device=0):
    """bigseqkit/grep.go:130-159 (Count forced off, :136-137)"""
    o._v["Count"] = False
    if o._v.get("DeleteMatched"):  # the driver keeps the lowest partition per pattern (bigseqkit/grep.go:144-156): global
        input = _one_shard(input)
    return _run_records("Grep", lib.bsk_grep_run, input, o, device)[0]


def GrepCount(input, o, device=0):
    """bigseqkit/grep.go:161-180: per-partition counts summed (GrepReduceCount)"""
    o._v["Count"] = True
    total = 0
    with Operator("Grep", o.to_json(), device) as op:
        for pid, ptr, n, on_dev, keep in input.partitions():
            out = _lib.Out()
            check(lib.bsk_grep_run(op.ctx, ptr, n, 1 if on_dev else 0, input.format, pid, None, C.byref(out)), op.ctx)
            cnt = C.c_uint64()
            check(lib.bsk_grep_last_count(op.ctx, C.byref(cnt)), op.ctx)
            total += cnt.value
    return total


def Subseq(input, o, device=0):
    """bigseqkit/subseq.go:86-100"""
    return _run_records("SubseqTransform", lib.bsk_subseq_run, input, o, device)[0]


def Locate(input, o, device=0):
    """bigseqkit/locate.go:122-134 (MapPartitionsWithIndex: partition 0 carries the header row)"""
    return _run_records("Locate", lib.bsk_locate_run, input, o, device)[0]


def Translate(input, o=None, device=0):
    """bigseqkit/translate.go:87-100"""
    return _run_records("Translate", lib.bsk_translate_run, input, o or SeqKitTranslateOptions(), device)[0]


def RmDup(input, o=None, device=0):
    """bigseqkit/rmdup.go:70-108 (GroupByKey, :97: duplicates are global -- several shards are joined into one)"""
    return _run_records("RmDup", lib.bsk_rmdup_run, _one_shard(input), o or SeqKitRmDupOptions(), device,
                        finish=lib.bsk_rmdup_finish)[0]


def Fq2Fa(input, o=None, device=0):
    """bigseqkit/fq2fa.go:25-37"""
    return _run_records("Fq2Fa", lib.bsk_fq2fa_run, input, o or SeqKitFq2FaOptions(), device)[0]


def Duplicate(input, o=None, device=0):
    """bigseqkit/duplicate.go:31-43 (Flatmap: the copies of a record are adjacent)"""
    return _run_records("Duplicate", lib.bsk_duplicate_run, input, o or SeqKitDuplicateOptions(), device)[0]


def Rename(input, o=None, device=0):
    """bigseqkit/rename.go:34-60 (ordinals count over the whole dataframe: several shards are joined, like RmDup)"""
    return _run_records("Rename", lib.bsk_rename_run, _one_shard(input), o or SeqKitRenameOptions(), device)[0]


def Sort(input, o=None, device=0):
    """bigseqkit/sort.go:91-147 (SortByKey over the whole dataframe: several shards are joined)"""
    return _run_records("Sort", lib.bsk_sort_run, _one_shard(input), o or SeqKitSortOptions(), device)[0]


def Faidx(input, o=None, device=0):
    """bigseqkit/faidx.go:61-95 (index rows only): the FaidxOffset pass is the running sum of the shard sizes"""
    chunks, base = [], 0
    with Operator("Faidx", (o or SeqKitFaidxOptions()).to_json(), device) as op:
        for pid, ptr, n, on_dev, keep in input.partitions():
            out = _lib.Out()
            check(lib.bsk_faidx_run(op.ctx, ptr, n, 1 if on_dev else 0, input.format, pid, base, None, C.byref(out)), op.ctx)
            buf = C.create_string_buffer(max(1, out.len))
            check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
            chunks.append(buf.raw[:out.len])
            base += n
    return b"".join(chunks)


def Pair(inputA, inputB, o=None, device=0):
    """bigseqkit/pair.go:68-100 -> (paired.1, paired.2, unpaired.1, unpaired.2) as the bytes SaveAsTextFile would write.
    Both inputs are one device-resident shard each; they are joined (file 1, then file 2) for the one call."""
    import torch
    a, b = inputA.shards[0], inputB.shards[0]
    if not (hasattr(a, "data_ptr") and hasattr(b, "data_ptr")):
        raise ValueError("Pair: both inputs must be device tensors")
    parts = [a]
    if a.numel() and int(a[-1]) != 10:
        parts.append(torch.tensor([10], dtype=torch.uint8, device=a.device))
    n_first = sum(p.numel() for p in parts)
    both = torch.cat(parts + [b]) if b.numel() or len(parts) > 1 else a
    outs = (_lib.Out * 4)()
    res = []
    with Operator("Pair", (o or SeqKitPairOptions()).to_json(), device) as op:
        check(lib.bsk_pair_run(op.ctx, C.c_void_p(both.data_ptr()) if both.numel() else None, both.numel(), n_first, 1,
                               inputA.format, None, outs), op.ctx)
        for k in range(4):
            buf = C.create_string_buffer(max(1, outs[k].len))
            check(lib.bsk_out_to_host(op.ctx, C.byref(outs[k]), buf, outs[k].len), op.ctx)
            res.append(buf.raw[:outs[k].len])
    return tuple(res)


def _join_files(tensors):
    """device tensors back to back, a newline added to a file that lacks the final one -> (tensor, [end offsets])"""
    import torch
    parts, ends, at = [], [], 0
    for t in tensors:
        parts.append(t)
        at += t.numel()
        if t.numel() and int(t[-1]) != 10:
            parts.append(torch.tensor([10], dtype=torch.uint8, device=t.device))
            at += 1
        ends.append(at)
    return (torch.cat(parts) if len(parts) > 1 else parts[0]), ends


def Common(inputA, inputB, o=None, *inputN, device=0):
    """bigseqkit/common.go:68-109 -> the records of inputA common to all inputs (one device-resident shard each)"""
    frames = [inputA, inputB, *inputN]
    both, ends = _join_files([f.shards[0] for f in frames])
    arr = (C.c_uint64 * len(ends))(*ends)
    out = _lib.Out()
    with Operator("Common", (o or SeqKitCommonOptions()).to_json(), device) as op:
        check(lib.bsk_common_run(op.ctx, C.c_void_p(both.data_ptr()) if both.numel() else None, both.numel(), arr, len(ends), 1,
                                 inputA.format, None, C.byref(out)), op.ctx)
        buf = C.create_string_buffer(max(1, out.len))
        check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
        return buf.raw[:out.len]


def Concat(inputA, inputB, o=None, device=0):
    """bigseqkit/concat.go:53-90 (one device-resident shard per input)"""
    both, ends = _join_files([inputA.shards[0], inputB.shards[0]])
    out = _lib.Out()
    with Operator("Concat", (o or SeqKitConcatOptions()).to_json(), device) as op:
        check(lib.bsk_concat_run(op.ctx, C.c_void_p(both.data_ptr()) if both.numel() else None, both.numel(), ends[0], 1,
                                 inputA.format, None, C.byref(out)), op.ctx)
        buf = C.create_string_buffer(max(1, out.len))
        check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
        return buf.raw[:out.len]


def FaidxQuery(input, o, device=0):
    """the `queries` dataframe of bigseqkit/faidx.go:96-106 (Regions / RegionFile of the options)"""
    return _run_records("Faidx", lib.bsk_faidx_query_run, input, o, device)[0]


def Count(input, device=0):
    """input.Count(): records per shard (the record table of every shard is built once)"""
    counts = []
    with Operator("SeqTransform", "{}", device) as op:
        for pid, ptr, n, on_dev, keep in input.partitions():
            nrec = C.c_uint64()
            check(lib.bsk_index_build(op.ctx, ptr, n, 1 if on_dev else 0, input.format, None, C.byref(nrec)), op.ctx)
            counts.append(nrec.value)
    return counts


def _range(op_name, input, o, device):
    """bigseqkit/range.go:36-103: MapWithIndex(RangePrepare) + Filter(RangeFilter); the index of a record is its
    position in the whole input, so every shard is told where it starts.  The counts cost one index pass per shard,
    which is what IgnisHPC's MapWithIndex (and input.Count() for negative positions) also pay."""
    chunks = []
    with Operator(op_name, o.to_json(), device) as op:
        counts = Count(input, device) if len(input.shards) > 1 else [0]
        needs = C.c_int()
        check(lib.bsk_range_needs_count(op.ctx, C.byref(needs)), op.ctx)
        if needs.value:
            if len(input.shards) == 1:
                counts = Count(input, device)
            check(lib.bsk_range_set_count(op.ctx, sum(counts)), op.ctx)
        first = 0
        for (pid, ptr, n, on_dev, keep), cnt in zip(input.partitions(), counts):
            out = _lib.Out()
            check(lib.bsk_range_run(op.ctx, ptr, n, 1 if on_dev else 0, input.format, pid, first, None, C.byref(out)), op.ctx)
            buf = C.create_string_buffer(max(1, out.len))
            check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
            chunks.append(buf.raw[:out.len])
            first += cnt
    return b"".join(chunks)


def Range(input, o, device=0):
    """bigseqkit/range.go:36-103"""
    return _range("Range", input, o, device)


def Head(input, o=None, device=0):
    """bigseqkit/head.go:34-44: Range("1:N")"""
    return _range("Head", input, o or SeqKitHeadOptions(), device)


def build_index(input, device=0):
    """Record table of the first shard (tests): list of (start, head_len, seq_len, aux)."""
    with Operator("SeqTransform", "{}", device) as op:
        for pid, ptr, n, on_dev, keep in input.partitions():
            nrec = C.c_uint64()
            check(lib.bsk_index_build(op.ctx, ptr, n, 1 if on_dev else 0, input.format, None, C.byref(nrec)), op.ctx)
            k = nrec.value
            st, hl, sl, ax = (C.c_uint64 * k)(), (C.c_uint32 * k)(), (C.c_uint32 * k)(), (C.c_uint32 * k)()
            check(lib.bsk_index_copy(op.ctx, st, hl, sl, ax, k), op.ctx)
            return list(zip(st, hl, sl, ax))
    return []
