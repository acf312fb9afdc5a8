"""ctypes binding of libbsk.so -- the C ABI declared in include/bsk.h.

The HIP library is the product; there is no Python/CPU fallback.  Importing this
module fails loudly when the shared library has not been built
(``python -c 'import __graft_entry__ as g; g.build()'`` or ``./build.sh``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BSK_LIB", os.path.join(_HERE, "lib", "libbsk.so"))

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"libbsk.so not found at {LIB_PATH}: build it with ./build.sh "
        "(hipcc --offload-arch=gfx950); bigseqkit_amd has no CPU fallback")



def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch ships its own libamdhip64.so (same SONAME,
    libamdhip64.so.7, as /opt/rocm's).  If libbsk.so were loaded first it would bind to
    /opt/rocm's copy and a later `import torch` would bring a second runtime into the
    process (observed: "No HIP GPUs are available", and streams / device pointers could not
    be shared).  Loading torch's copy first -- without importing torch -- makes both bind
    to it.  Without torch installed, libbsk.so uses /opt/rocm's runtime via its RUNPATH."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def _import_torch_first():
    """PyTorch, when it is installed, is imported BEFORE libbsk.so touches the HIP runtime.  Measured on the GPU box
    (scripts/dbg_torch_init.py): `import torch` after a HIP call of libbsk (hipGetDeviceCount is enough) takes 2.7 - 10.5 s
    instead of 0.8 s -- with the runtime already initialised, libtorch_hip.so registers ALL its code objects eagerly while it
    is loaded, which reads the whole library; on a box whose image is not in the page cache that was a wait of 9 - 13
    minutes in front of the first GPU test of a session (round 3: two of five full test runs).  Imported first, torch
    registers lazily.
    Round 4: importing bigseqkit_amd no longer imports torch by itself -- 0.8 s and torch's memory are a heavy
    hidden side effect for a caller that never uses it.  The rule is the caller's: `import torch` BEFORE `import bigseqkit_amd`
    (tests/conftest.py, bench.py, bigseqkit_amd/run.py do), or set BSK_TORCH_FIRST=1 to have it done here."""
    import sys
    if "torch" in sys.modules or os.environ.get("BSK_TORCH_FIRST") != "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


_import_torch_first()
_preload_hip_runtime()
lib = C.CDLL(LIB_PATH)

BSK_OK, BSK_ERR_INVALID_ARG, BSK_ERR_OPTS, BSK_ERR_FORMAT, BSK_ERR_UNSUPPORTED, BSK_ERR_HIP, \
    BSK_ERR_NO_DEVICE, BSK_ERR_CAPACITY, BSK_ERR_OVERFLOW_EXCHANGE = range(9)
FORMAT_FASTA, FORMAT_FASTQ = 0, 1
STATS_HDR = 8
SYNTH_FASTQ150, SYNTH_FASTA1K, SYNTH_FASTA5K_CDS, SYNTH_FASTA5K_VAR = 0, 1, 2, 3
SYNTH_FLAG_MOTIF, SYNTH_FLAG_DUPS = 1, 2


class StatInfo(C.Structure):
    _fields_ = [("type", C.c_char * 16),
                ("num", C.c_uint64), ("len_sum", C.c_uint64), ("gap_sum", C.c_uint64),
                ("len_min", C.c_uint64), ("len_max", C.c_uint64), ("n50", C.c_uint64),
                ("l50", C.c_int64),
                ("len_avg", C.c_double), ("q1", C.c_double), ("q2", C.c_double), ("q3", C.c_double),
                ("q20", C.c_double), ("q30", C.c_double)]


class Out(C.Structure):
    # (d_seg_*: the result as ordered slices -- switch "out" = "slices", include/bsk.h; n_segments == 0: d_data holds the text)
    _fields_ = [("d_data", C.c_void_p), ("len", C.c_size_t), ("records", C.c_uint64),
                ("d_seg_src", C.c_void_p), ("d_seg_off", C.c_void_p), ("n_segments", C.c_uint64)]


_vp, _sz, _i, _i64, _u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_uint64
_p = C.POINTER

# every symbol include/bsk.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "bsk_version": (_i, []),
    "bsk_device_count": (_i, []),
    "bsk_global_error": (C.c_char_p, []),
    "bsk_last_error": (C.c_char_p, [_vp]),
    "bsk_create": (_i, [C.c_char_p, C.c_char_p, _i, _p(_vp)]),
    "bsk_destroy": (None, [_vp]),
    "bsk_opts_json": (C.c_char_p, [_vp]),
    "bsk_log_text": (C.c_char_p, [_vp]),
    "bsk_ctx_set": (_i, [_vp, C.c_char_p, C.c_char_p]),
    "bsk_find_record_start": (_i, [_vp, _sz, _sz, _i, _p(_sz)]),
    "bsk_stats_vector_len": (_sz, [_vp]),
    "bsk_stats_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _vp]),
    "bsk_stats_reset": (_i, [_vp, _vp]),
    "bsk_stats_collect": (_i, [_vp, _vp, _p(_i64), _p(_i64), _sz, _p(_sz)]),
    "bsk_stats_overflow_total": (_i, [_vp, _p(_u64)]),
    "bsk_stats_overflow_get": (_i, [_vp, _p(_u64), _sz, _p(_sz)]),
    "bsk_stats_overflow_add": (_i, [_vp, _p(_u64), _sz]),
    "bsk_stats_collect_host": (_i, [_vp, _p(_u64), _sz, _vp, _sz, _i, _p(_i64), _p(_i64), _sz, _p(_sz)]),
    "bsk_stats_merge": (_i, [_p(_i64), _p(_i64), _sz, _p(_i64), _p(_i64), _sz, _p(_i64), _p(_i64), _sz, _p(_sz)]),
    "bsk_stats_finalize": (_i, [_vp, _p(_i64), _p(_i64), _sz, _p(StatInfo)]),
    "bsk_stats_string": (_i, [_vp, C.c_char_p, C.c_char_p, _p(StatInfo), C.c_char_p, _sz]),
    "bsk_out_to_host": (_i, [_vp, _p(Out), _vp, _sz]),
    "bsk_out_materialize": (_i, [_vp, _p(Out), _vp]),
    "bsk_index_build": (_i, [_vp, _vp, _sz, _i, _i, _vp, _p(_u64)]),
    "bsk_index_copy": (_i, [_vp, _p(_u64), _p(C.c_uint32), _p(C.c_uint32), _p(C.c_uint32), _sz]),
    "bsk_seq_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_grep_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_grep_last_count": (_i, [_vp, _p(_u64)]),
    "bsk_subseq_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_locate_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_translate_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_rmdup_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_fq2fa_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_rename_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_sort_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_pair_run": (_i, [_vp, _vp, _sz, _sz, _i, _i, _vp, _p(Out)]),
    "bsk_concat_run": (_i, [_vp, _vp, _sz, _sz, _i, _i, _vp, _p(Out)]),
    "bsk_common_run": (_i, [_vp, _vp, _sz, _p(C.c_uint64), C.c_uint32, _i, _i, _vp, _p(Out)]),
    "bsk_faidx_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, C.c_uint64, _vp, _p(Out)]),
    "bsk_faidx_query_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_duplicate_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, _vp, _p(Out)]),
    "bsk_range_run": (_i, [_vp, _vp, _sz, _i, _i, _i64, C.c_uint64, _vp, _p(Out)]),
    "bsk_range_needs_count": (_i, [_vp, _p(C.c_int)]),
    "bsk_range_set_count": (_i, [_vp, C.c_uint64]),
    "bsk_range_bounds": (_i, [_vp, _p(_i64), _p(_i64)]),
    "bsk_device_select": (_i, [_i]),
    "bsk_device_alloc": (_vp, [_sz]),
    "bsk_device_free": (None, [_vp]),
    "bsk_device_copy": (_i, [_vp, _vp, _sz, _i]),
    "bsk_host_alloc": (_vp, [_sz]),
    "bsk_host_free": (None, [_vp]),
    "bsk_regex_match": (_i, [C.c_char_p, _vp, _sz, _p(C.c_int)]),
    "bsk_rmdup_finish": (_i, [_vp]),
    "bsk_rmdup_dist_keys": (_i, [_vp, _vp, _sz, _i, _vp, _p(C.c_uint64)]),
    "bsk_rmdup_dist_pack": (_i, [_vp, C.c_uint64, _i, _vp, _p(C.c_uint64), _vp]),
    "bsk_rmdup_dist_resolve": (_i, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "bsk_rmdup_dist_emit": (_i, [_vp, _vp, _vp, C.c_uint64, _vp, _p(Out)]),
    "bsk_rmdup_dist_resolve_ex": (_i, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "bsk_rmdup_dist_emit_ex": (_i, [_vp, _vp, _vp, _vp, C.c_uint64, _vp, _p(Out), _p(C.c_uint64)]),
    "bsk_comm_unique_id": (_i, [_vp]),
    "bsk_comm_init_rank": (_i, [_i, _i, _vp, _i, _p(_vp)]),
    "bsk_comm_init_all": (_i, [_i, _p(_i), _p(_vp)]),
    "bsk_comm_destroy": (_i, [_vp]),
    "bsk_comm_info": (_i, [_vp, _p(_i), _p(_i), _p(_i), _p(_i)]),
    "bsk_comm_error": (C.c_char_p, [_vp]),
    "bsk_comm_barrier": (_i, [_vp, _vp]),
    "bsk_comm_allreduce_u64": (_i, [_vp, _vp, _sz, _i, _vp]),
    "bsk_comm_allgather_u64": (_i, [_vp, _u64, _p(_u64), _vp]),
    "bsk_count_allreduce": (_i, [_vp, _p(_u64), _vp]),
    "bsk_stats_collect_reduced": (_i, [_vp, _vp, _vp, _vp, _p(_i64), _p(_i64), _sz, _p(_sz)]),
    "bsk_rmdup_dist_run": (_i, [_vp, _vp, _vp, _sz, _i, _vp, _p(Out)]),
    "bsk_rmdup_dist_xpack": (_i, [_vp, _vp, _vp, _vp, C.c_uint64, _p(C.c_uint64), _i, _p(C.c_uint64), _p(C.c_uint64), _p(_vp), _p(_vp), _vp]),
    "bsk_rmdup_dist_xcompare": (_i, [_vp, _vp, _p(C.c_uint64), _vp, _p(C.c_uint64), _i, _vp, _vp]),
    "bsk_rmdup_dist_xapply": (_i, [_vp, _vp, _p(C.c_uint64), _p(C.c_uint64), _vp]),
    "bsk_rmdup_dist_flagged_get": (_i, [_vp, _vp, _sz, _p(_sz)]),
    "bsk_rmdup_dist_flagged_settle": (_i, [_vp, _vp, _sz]),
    "bsk_rmdup_dist_stats": (_i, [_vp, _p(C.c_uint64), _p(C.c_uint64), _p(C.c_uint64)]),
    "bsk_shard_load": (_i, [_i, _u64, _sz, _i, _i, C.POINTER(_vp)]),
    "bsk_synth_record_bytes": (_sz, [_i]),
    "bsk_synth_offset": (_u64, [_i, _u64]),
    "bsk_synth_host": (_i, [_i, _u64, C.c_uint, _u64, _vp, _sz]),
    "bsk_synth_device": (_i, [_i, _u64, C.c_uint, _u64, _vp, _sz, _i, _vp]),
    "bsk_event_create": (_i, [_p(_vp)]),
    "bsk_event_record": (_i, [_vp, _vp]),
    "bsk_event_elapsed_ms": (_i, [_vp, _vp, _p(C.c_float)]),
    "bsk_event_destroy": (_i, [_vp]),
    "bsk_selftest_regex_find": (_i, [C.c_char_p, C.c_char_p, _sz, _sz, _p(C.c_uint32), _p(C.c_uint32)]),
    "bsk_store_open": (_i, [C.c_char_p, _i, _p(_vp)]),
    "bsk_store_error": (C.c_char_p, [_vp]),
    "bsk_store_put": (_i, [_vp, _vp, _u64, _vp]),
    "bsk_store_put_host": (_i, [_vp, _u64, _vp, _sz]),
    "bsk_store_close": (_i, [_vp, _p(_u64)]),
    "bsk_run_to_store": (_i, [_vp, _vp, _sz, _i, _i64, _vp, _u64, _p(_u64), _p(_u64)]),
    "bsk_profile_enable": (_i, [_vp, _i]),
    "bsk_profile_read": (_i, [_vp, C.c_char_p, _p(C.c_double), _p(_u64)]),
    "bsk_profile_reset": (_i, [_vp]),
    "bsk_profile_dump": (_i, [_vp, C.c_char_p, _sz]),
    "bsk_selftest_scan": (_i, [_i, _p(C.c_uint32), _p(C.c_uint32)]),
    "bsk_selftest_rmdup_keys": (_i, [_vp, _p(_u64), _p(_u64), _sz, _p(_sz)]),
    "bsk_selftest_stream_read": (_i, [_vp, _sz, _i, _i, _p(C.c_float)]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)  # AttributeError here == symbol missing from the build
    _f.restype = _res
    _f.argtypes = _args


class BskError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def check(rc, ctx=None):
    if rc != BSK_OK:
        # (every failure also leaves its text in the calling thread's bsk_global_error(); a call refused as "context busy"
        # leaves it ONLY there -- the context's text belongs to the call that is running)
        # (the thread's text is sticky: only a refusal -- always BSK_ERR_INVALID_ARG -- is looked up there first)
        g = lib.bsk_global_error() or b""
        msg = g if (not ctx or (rc == BSK_ERR_INVALID_ARG and b"context busy" in g)) else (lib.bsk_last_error(ctx) or g)
        raise BskError(rc, (msg or b"").decode("utf-8", "replace"))
