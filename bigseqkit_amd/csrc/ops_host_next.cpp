// Host side of the operators SURVEY 8(f) lists as "next": fq2fa, range / head, duplicate (rank 2); rename, pair,
// common, concat (rank 3); sort, faidx index rows (rank 4).  Option validation, driver-side arithmetic and the launch
// sequences; the kernels are in ops_records / ops_group / ops_sort / ops_faidx / ops_concat and the seq emit.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/bsk.h"
#include "ctx.hpp"
#include "ops_host.hpp"
#include "ops_host_internal.hpp"
#include "ops_concat.hpp"
#include "ops_faidx.hpp"
#include "ops_grep.hpp"
#include "ops_group.hpp"
#include "ops_records.hpp"
#include "ops_segcopy.hpp"
#include "ops_rmdup.hpp"
#include "ops_seq.hpp"
#include "ops_sort.hpp"
#include "stream_stats.hpp"

namespace bsk {

// The grouping / sorting operators pack (group << 32) | record index and use 32-bit permutations: a shard of 2^32 or more
// records (> 4 x 10^9: more than 1 TB of 317-byte reads) is refused instead of being grouped wrongly.
static int check_u32_records(bsk_ctx* c, const char* op) {
    if (c->table.n < (1ull << 32)) return BSK_OK;
    c->set_error(std::string("libbsk: ") + op + ": 2^32 or more records in one shard are not supported (cut the input into more shards)");
    return BSK_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------
// fq2fa, range / head, duplicate
// ---------------------------------------------------------------------------
// Go strconv.ParseInt(s, 10, 64) with its error text
static int64_t go_parse_int(const std::string& s) {
    const std::string err = "strconv.ParseInt: parsing \"" + s + "\": invalid syntax";
    size_t i = 0;
    bool neg = false;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) neg = s[i++] == '-';
    if (i >= s.size()) throw OptError(err);
    unsigned long long v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') throw OptError(err);
        if (v > (0x7FFFFFFFFFFFFFFFull - (unsigned)(s[i] - '0')) / 10ull)
            throw OptError("strconv.ParseInt: parsing \"" + s + "\": value out of range");
        v = v * 10ull + (unsigned)(s[i] - '0');
    }
    return neg ? -(int64_t)v : (int64_t)v;
}

// Before() of Fq2Fa (bigseqkit-lib/fq2fa.go:26-33) and the driver side of Range / Head / Duplicate
// (bigseqkit/range.go:36-66, head.go:34-44, duplicate.go:31-43)
void validate_records_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    if (c->op == Op::Fq2Fa || c->op == Op::Rename || c->op == Op::Pair || c->op == Op::Concat) { check_id_regexp(c); return; }
    if (c->op == Op::Duplicate) {
        // make([]string, times) panics for a negative count; zero copies is an empty result
        if (o.i("Times") < 0) throw OptError("value of -n (--times) should not be negative");
        if (o.i("Times") > 0xFFFFFFFFll) throw OptError("value of -n (--times) too large");
        return;
    }
    std::string range = c->op == Op::Head ? "1:" + std::to_string(o.i("N")) : o.s("Range");
    if (range.empty()) throw OptError("flag -r (--range) needed");
    std::vector<std::string> r;  // strings.Split(range, ":")
    for (size_t a = 0;;) {
        const size_t b = range.find(':', a);
        r.push_back(range.substr(a, b == std::string::npos ? std::string::npos : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
    }
    int64_t start = go_parse_int(r[0]);
    int64_t end = -1;
    if (r.size() > 1) end = go_parse_int(r[1]);
    if (start == 0 || end == 0) throw OptError("either start and end should not be 0");
    if (start > 0) --start;
    if (end == -1) end = INT64_MAX;
    c->range_start = start;
    c->range_end = end;
    c->range_needs_count = start < -1 || end < -1;  // range.go:69
    c->range_resolved = false;
    if (!c->range_needs_count) {
        const int rc = range_resolve(c, 0);
        if (rc != BSK_OK) throw OptError(c->last_error);
    }
}

// bigseqkit/range.go:69-86: negative positions count from the end.  PARITY.md RNG: the reference's final check reads
// `if start <= end { error }`, which rejects every non-empty range; the evident intent (an empty or inverted range is
// the error) is what runs here, the arithmetic above it is kept as written.
int range_resolve(bsk_ctx* c, int64_t n_records) {
    if (c->range_resolved) return BSK_OK;
    if (c->range_needs_count) {
        if (c->range_start < 0) c->range_start += n_records;
        if (c->range_end < 0) c->range_end += n_records;
    }
    if (c->range_start >= c->range_end) {
        c->set_error("start must be > than end");
        return BSK_ERR_OPTS;
    }
    c->range_resolved = true;
    return BSK_OK;
}

// Fq2Fa.Call (bigseqkit-lib/fq2fa.go:35-59): record.Seq.Qual = []; record.Format(0) -- '>' + name, the sequence on one line
int fq2fa_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const bool fastq = format == BSK_FORMAT_FASTQ;
    int rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    SeqParams P = format_params(c, fastq);
    P.print_qual = 0;
    P.line_width = 0;
    P.fasta_out = 1;
    P.buf_end = d_buf + n;
    TextTableH tt{nullptr, nullptr, nullptr};
    rc = prepare_text(c, d_buf, format, st, &tt);
    if (rc != BSK_OK) return rc;
    P.text_w = tt.text_w; P.lin_off = tt.lin_off; P.lin = tt.lin;
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, launch_seq_size(d_buf, c->table, P, c->d_out_len, c->d_status, st));
    uint64_t total = 0, kept = 0;
    rc = finish_sizes(c, st, &total, &kept);
    if (rc != BSK_OK) return rc;
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    apply_long(c, &P);
    { const int rce = emit_records(c, d_buf, n, P, total, kept, st); if (rce != BSK_OK) return rce; }
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    return BSK_OK;
}

// RangePrepare + RangeFilter (bigseqkit-lib/range.go:26-43), Duplicate.Call (duplicate.go:24-30)
int records_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    RecordsParams P;
    memset(&P, 0, sizeof P);
    P.fastq = format == BSK_FORMAT_FASTQ;
    P.first_record = c->cur_first_record;
    if (c->op == Op::Duplicate) {
        P.lo = INT64_MIN;
        P.hi = INT64_MAX;
        P.times = (uint32_t)c->opts.i("Times");
        if (P.times == 0) return empty_result(c, out);
    } else {
        if (!c->range_resolved) {
            c->set_error("libbsk: a range with negative positions needs the record count first (bsk_range_set_count)");
            return BSK_ERR_INVALID_ARG;
        }
        P.lo = c->range_start;
        P.hi = c->range_end;
        P.times = 1;
    }
    int rc = build_index(c, d_buf, n, format, st);
    uint64_t total = 0, kept = 0, status = 0;
    for (int attempt = 0;; ++attempt) {
        if (P.fastq && !c->norm_active &&
            (rc == BSK_ERR_MULTILINE_FASTQ || (rc != BSK_OK && (c->last_kernel_flags & STRICT_FASTQ_FLAGS)))) {
            // FASTQ records on more than four lines (helper.go:252-269; at the head of the shard, or -- the strict reader
            // complained -- further down): these operators print the record TEXT, wrapped as it stands, so the multi-line
            // reader only says where the records begin and the text leaves like FASTA text does: from one record start to
            // the next, minus the final newline
            const std::string msg = c->last_error;
            const int rc0 = rc;
            size_t n_eff = n;
            HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 2 * sizeof(uint64_t), st));
            rc = normalize_multiline_fastq(c, d_buf, n, st, nullptr, &n_eff);
            if (rc != BSK_OK) {
                if (rc0 != BSK_ERR_MULTILINE_FASTQ) { c->set_error(msg); return rc0; }  // (not FASTQ either way: the first complaint stands)
                return rc;
            }
            P.fastq = 0;
            n = n_eff;
        }
        if (rc != BSK_OK) return rc;
        if (c->table.n == 0) return empty_result(c, out);
        rc = ensure_record_scratch(c);
        if (rc != BSK_OK) return rc;
        HIP_TRYX(c, launch_records_size(d_buf, n, c->table, P, c->d_out_len, c->d_status, st));
        HIP_TRYX(c, launch_scan_u32(c->d_out_len, c->d_out_off, c->table.n, c->d_scan_tmp, st));
        HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
        HIP_TRYX(c, launch_count_nonzero(c->d_out_len, c->table.n, c->d_counter, st));
        HIP_TRYX(c, hipMemcpyAsync(&total, c->d_out_off + c->table.n, sizeof total, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipMemcpyAsync(&kept, c->d_counter, sizeof kept, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        rc = kernel_error_to_status(c, status);
        // (the index pass leaves its complaints in the status word: a shard that is wrapped behind its head is seen here)
        if (rc != BSK_OK && attempt == 0 && P.fastq && !c->norm_active && (status & STRICT_FASTQ_FLAGS)) continue;
        if (rc != BSK_OK) return rc;
        break;
    }
    out->d_data = nullptr;
    out->len = 0;
    out->records = 0;
    if (total == 0) return BSK_OK;
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    const char* sg = c->tune.get("segcopy");
    if (P.times == 1 && !(sg && strcmp(sg, "off") == 0)) {
        // range / head: the kept records are verbatim segments of the shard (ops_segcopy.hip)
        const RecordTable& t = c->table;
        rc = grow(c, &c->d_seg_src, &c->seg_src_cap, t.n + 1, t.n / 8 + 16);
        if (rc != BSK_OK) return rc;
        rc = grow(c, &c->d_seg_first, &c->seg_first_cap, seg_tiles(total) + 1, 64);
        if (rc != BSK_OK) return rc;
        uint64_t* d_other = c->d_seg_src + t.n;
        HIP_TRYX(c, hipMemsetAsync(d_other, 0, sizeof(uint64_t), st));
        HIP_TRYX(c, launch_seg_build_text(d_buf, n, t, c->d_out_len, c->d_seg_src, d_other, st));
        HIP_TRYX(c, launch_seg_first(c->d_out_off, t.n, c->d_seg_first, st));
        HIP_TRYX(c, launch_seg_copy(c->d_seg_src, c->d_out_off, t.n, c->d_seg_first, c->d_out, total, d_buf, d_buf + n, st));
        uint64_t other = 0;
        HIP_TRYX(c, hipMemcpyAsync(&other, d_other, sizeof other, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (other) HIP_TRYX(c, launch_seg_fix_text(d_buf, t, c->d_out_len, c->d_out_off, c->d_seg_src, c->d_out, st));
    } else if (P.times <= 4 && !(sg && strcmp(sg, "off") == 0)) {
        // duplicate -n 2..4: `times` segments per record (more copies: the tile copy below, whose tables do not grow with n)
        const RecordTable& t = c->table;
        const uint64_t ns = t.n * P.times;
        rc = grow(c, &c->d_seg_src, &c->seg_src_cap, 2 * ns + 2, ns / 4 + 16);
        if (rc != BSK_OK) return rc;
        rc = grow(c, &c->d_seg_first, &c->seg_first_cap, seg_tiles(total) + 1, 64);
        if (rc != BSK_OK) return rc;
        uint64_t* seg_src = c->d_seg_src;
        uint64_t* seg_off2 = c->d_seg_src + ns;       // [ns + 1]
        uint64_t* d_other = c->d_seg_src + 2 * ns + 1;
        HIP_TRYX(c, hipMemsetAsync(d_other, 0, sizeof(uint64_t), st));
        HIP_TRYX(c, launch_seg_build_text_times(d_buf, n, t, c->d_out_len, c->d_out_off, P.times, seg_src, seg_off2, d_other, st));
        HIP_TRYX(c, launch_seg_first(seg_off2, ns, c->d_seg_first, st));
        HIP_TRYX(c, launch_seg_copy(seg_src, seg_off2, ns, c->d_seg_first, c->d_out, total, d_buf, d_buf + n, st));
        uint64_t other = 0;
        HIP_TRYX(c, hipMemcpyAsync(&other, d_other, sizeof other, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (other) HIP_TRYX(c, launch_seg_fix_text_times(d_buf, t, c->d_out_len, c->d_out_off, P.times, seg_src, c->d_out, st));
    } else {
        rc = grow(c, &c->d_tile_first, &c->tile_first_cap, records_copy_tiles(total), 64);
        if (rc != BSK_OK) return rc;
        HIP_TRYX(c, launch_records_copy(d_buf, c->table, P, c->d_out_off, c->d_tile_first, c->d_out, total, st));
    }
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept * P.times;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// rename: RenamePrepare + GroupByKey + Rename (bigseqkit-lib/rename.go:39-131).
// The k-th further record (k >= 1, file order) of an ID -- or of a whole name with ByName -- is printed as
// "<ID>_<k> <Desc>"; everything is re-formatted with Format(LineWidth).  Global like rmdup: one call sees the whole
// input.  Output in file order (the reference's group order is whatever GroupByKey yields); PARITY.md REN.
// ---------------------------------------------------------------------------
int rename_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    int rc = build_index(c, d_buf, n, format, st);
    if (rc == BSK_OK) rc = check_u32_records(c, "rename");  // (group << 32 | index) keys, u32 permutations
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    TextTableH tt;
    rc = prepare_text(c, d_buf, format, st, &tt);
    if (rc != BSK_OK) return rc;
    RmDupParams P;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    P.by_name = o.b("ByName");
    P.id_mode = id_mode_of(c);
    P.line_width = fastq ? 0 : (int)o.ci("LineWidth");
    P.buf_end = d_buf + n;
    const uint64_t N = c->table.n;
    rc = grow(c, &c->d_keys, &c->keys_cap, N, N / 8 + 16);
    if (rc != BSK_OK) return rc;
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    // groups: the rmdup machinery (XXH64 of the ID / name, first occurrence wins, exact verification of every other one)
    HIP_TRYX(c, launch_rmdup_hash(d_buf, n, c->table, tt, P, c->d_keys, nullptr, st));
    Arena A;
    const uint64_t o_has = A.take(N), o_ord = A.take(N * 4);
    rc = arena_reserve(c, &A);
    if (rc != BSK_OK) return rc;
    uint8_t* d_has = A.at<uint8_t>(o_has);   // (launch_rmdup_group also marks the groups of two or more; not needed here)
    uint32_t* d_ord = A.at<uint32_t>(o_ord);
    uint64_t* d_list = nullptr;
    void* d_tmp = nullptr;
    auto cleanup = [&]() {};
    auto fail = [&](int code) { return code; };
    HIP_TRYX(c, hipMemsetAsync(d_has, 0, N, st));
    HIP_TRYX(c, hipMemsetAsync(d_ord, 0, N * 4, st));
    // collisions checked, and d_keys[i] := first record of i's group
    rc = group_resolve(c, d_buf, tt, P, d_has, st);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
    // how many records are not the first of their group
    uint64_t status = 0, m = 0;
    {
        // count first (the list is allocated to size): out_len of resolve is 0 exactly for the dropped records
        HIP_TRYX(c, launch_count_nonzero(c->d_out_len, N, c->d_counter, st));
        uint64_t firsts = 0;
        HIP_TRYX(c, hipMemcpyAsync(&firsts, c->d_counter, 8, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, 8, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (status & ERR_HASH_COLLISION) {
            c->set_error("libbsk: two distinct subjects share one 64-bit XXH64 key; refusing to guess (rerun on the CPU path)");
            return fail(BSK_ERR_UNSUPPORTED);
        }
        rc = kernel_error_to_status(c, status);
        if (rc != BSK_OK) return fail(rc);
        m = N - firsts;
    }
    if (m) {
        size_t tmp_bytes = 0;
        if (group_sort_temp_bytes(m, &tmp_bytes) != hipSuccess) { c->set_error("libbsk: rocPRIM sort size query failed"); return BSK_ERR_HIP; }
        // the arena may move when it grows: d_ord has to survive, so it is re-derived after the reservation
        const uint64_t o_list = A.take(2 * m * 8), o_tmp = A.take(tmp_bytes ? tmp_bytes : 16);
        if (A.used > c->arena_cap) {
            // grow by hand, keeping the first part (has / ord)
            uint8_t* nb = nullptr;
            const uint64_t cap = A.used + A.used / 8 + 256;
            if (hipMalloc((void**)&nb, cap) != hipSuccess) { c->set_error("libbsk: out of device memory (rename)"); return BSK_ERR_HIP; }
            HIP_TRYX(c, hipMemcpyAsync(nb, c->d_arena, o_list, hipMemcpyDeviceToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            hipFree(c->d_arena);
            c->d_arena = nb;
            c->arena_cap = cap;
        }
        A.base = c->d_arena;
        d_has = A.at<uint8_t>(o_has);
        d_ord = A.at<uint32_t>(o_ord);
        d_list = A.at<uint64_t>(o_list);
        d_tmp = A.at<uint8_t>(o_tmp);
        HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 8, st));
        HIP_TRYX(c, launch_group_compact(c->d_keys, N, d_list, c->d_counter, st));
        HIP_TRYX(c, launch_group_sort(d_tmp, tmp_bytes, d_list, d_list + m, m, st, N));
        HIP_TRYX(c, launch_group_ordinals(d_list + m, m, d_ord, st));
    }
    SeqParams F = format_params(c, fastq);
    F.text_w = tt.text_w; F.lin_off = tt.lin_off; F.lin = tt.lin;
    F.buf_end = d_buf + n;
    F.ren_ord = d_ord;
    HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 2 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_seq_size(d_buf, c->table, F, c->d_out_len, c->d_status, st));
    uint64_t total = 0, kept = 0;
    rc = finish_sizes(c, st, &total, &kept);
    if (rc != BSK_OK) return fail(rc);
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return fail(rc);
    apply_long(c, &F);
    if (m * 4 > N) {
        // many renamed records: their heads are rewritten record by record anyway, and one kernel over all records beats
        // the segmented copy of the rest plus that kernel (every ID twice: 59 vs 63 ms)
        HIP_TRYX(c, launch_seq_emit(d_buf, c->table, F, c->d_out_len, c->d_out_off, c->d_out, st, total, kept));
    } else {
        const int rce = emit_records(c, d_buf, n, F, total, kept, st);
        if (rce != BSK_OK) return rce;
    }
    cleanup();
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// sort: driver bigseqkit/sort.go:91-147, executor bigseqkit-lib/sort.go:38-166
// ---------------------------------------------------------------------------
void validate_sort_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    int k = 0;  // sort.go:105-119 (ByBases implies ByLength)
    if (o.b("BySeq")) ++k;
    if (o.b("ByName")) ++k;
    if (o.b("ByLength") || o.b("ByBases")) ++k;
    if (k > 1) throw OptError("only one of the options (byLength), (byName) and (bySeq) is allowed");
    if (o.i("SeqPrefixLength") < 0) throw OptError("value of flag -L (--seq-prefix-length) should be >= 0");
}

int sort_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    int rc = build_index(c, d_buf, n, format, st);
    if (rc == BSK_OK) rc = check_u32_records(c, "sort");  // (group << 32 | index) keys, u32 permutations
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    TextTableH tt;
    rc = prepare_text(c, d_buf, format, st, &tt);
    if (rc != BSK_OK) return rc;
    const uint64_t N = c->table.n;
    SortParams P;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    P.mode = o.b("ByBases") ? 4 : o.b("ByLength") ? 3 : o.b("BySeq") ? 2 : o.b("ByName") ? 1 : 0;
    P.ignore_case = o.b("IgnoreCase");
    P.id_mode = id_mode_of(c);
    P.prefix_len = (uint32_t)std::min<int64_t>(o.i("SeqPrefixLength"), 0xFFFFFFFFll);
    set_bits(P.gap_set, o.s("GapLetters"));
    P.buf_end = d_buf + n;
    const bool desc = o.b("Reverse");  // SortByKey(!reverse, ...)
    // scratch: keys x2, perm x2, key lengths, rocPRIM temporary storage
    size_t tmp_bytes = 0;
    if (sort_pairs_temp_bytes(N, &tmp_bytes) != hipSuccess) { c->set_error("libbsk: rocPRIM sort size query failed"); return BSK_ERR_HIP; }
    Arena A;
    const uint64_t o_keys = A.take(2 * N * 8), o_perm = A.take(2 * N * 4), o_klen = A.take((N + 1) * 4),
                   o_tmp = A.take(tmp_bytes ? tmp_bytes : 16);
    // string keys of more than three chunks: the tie pass (below)
    const bool may_tie = P.mode < 3;
    const uint64_t o_tied = A.take(may_tie ? N * 4 : 0), o_start = A.take(may_tie ? N * 4 : 0), o_rank = A.take(may_tie ? (N + 1) * 8 : 0),
                   o_run = A.take(may_tie ? (N + 1) * 8 : 0), o_runof = A.take(may_tie ? N * 8 : 0), o_subpos = A.take(may_tie ? N * 4 : 0),
                   o_subperm = A.take(may_tie ? 2 * N * 4 : 0);
    rc = arena_reserve(c, &A);
    if (rc != BSK_OK) return rc;
    uint64_t* d_keys2 = A.at<uint64_t>(o_keys);   // [2 N]
    uint32_t* d_perm2 = A.at<uint32_t>(o_perm);   // [2 N]
    uint32_t* d_klen = A.at<uint32_t>(o_klen);    // [N + 1]   (last: max)
    void* d_tmp = A.at<uint8_t>(o_tmp);
    auto cleanup = [&]() {};
    auto fail = [&](int code) { return code; };
    uint64_t* kin = d_keys2;
    uint64_t* kout = d_keys2 + N;
    uint32_t* pin = d_perm2;
    uint32_t* pout = d_perm2 + N;
    if (launch_sort_iota(pin, N, st) != hipSuccess) return fail(BSK_ERR_HIP);
    if (P.mode >= 3) {
        if (launch_sort_intkeys(d_buf, c->table, tt, P, kin, st) != hipSuccess ||
            launch_sort_pairs(d_tmp, tmp_bytes, kin, kout, pin, pout, N, desc, 32, st) != hipSuccess) return fail(BSK_ERR_HIP);
        std::swap(pin, pout);
    } else {
        uint8_t* d_nat = nullptr;
        if (o.b("InNaturalOrder") && P.mode <= 1) {
            // natural order (sort.go:130-133: IDs / names only): keys rewritten so that byte order is natural order
            uint32_t* d_nlen = c->d_out_len;     // scratch of N entries, free until the size pass
            uint64_t* d_noff = c->d_out_off;     // [N + 1]
            rc = ensure_record_scratch(c);
            if (rc != BSK_OK) return fail(rc);
            d_nlen = c->d_out_len; d_noff = c->d_out_off;
            uint64_t nat_bytes = 0;
            if (launch_sort_natlen(d_buf, c->table, P, d_nlen, st) != hipSuccess ||
                launch_scan_u32(d_nlen, d_noff, N, c->d_scan_tmp, st) != hipSuccess ||
                hipMemcpyAsync(&nat_bytes, d_noff + N, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) return fail(BSK_ERR_HIP);
            // the offsets must survive the size pass below: keep them (and the keys) in their own allocation
            if (hipMalloc((void**)&d_nat, nat_bytes + 16 + (N + 1) * 8) != hipSuccess) { c->set_error("libbsk: out of device memory (sort -N)"); return fail(BSK_ERR_HIP); }
            uint64_t* d_noff2 = reinterpret_cast<uint64_t*>(d_nat);
            uint8_t* d_keys_nat = d_nat + (N + 1) * 8;
            if (hipMemcpyAsync(d_noff2, d_noff, (N + 1) * 8, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                launch_sort_natkeys(d_buf, c->table, P, d_noff2, d_keys_nat, st) != hipSuccess) { hipFree(d_nat); return fail(BSK_ERR_HIP); }
            P.nat = d_keys_nat;
            P.nat_off = d_noff2;
        }
        struct FreeNat { uint8_t* p; ~FreeNat() { if (p) hipFree(p); } } free_nat{d_nat};
        uint32_t maxlen = 0;
        if (hipMemsetAsync(d_klen + N, 0, 4, st) != hipSuccess ||
            launch_sort_keylen(d_buf, c->table, P, d_klen, d_klen + N, st) != hipSuccess ||
            hipMemcpyAsync(&maxlen, d_klen + N, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) return fail(BSK_ERR_HIP);
        const uint32_t nchunks = (maxlen + 7) / 8;
        const char* se = c->tune.get("sort");  // lsd: every chunk for every record (the round-1 path)
        if (nchunks <= 3 || (se && strcmp(se, "lsd") == 0)) {
            // LSD over the 8-byte chunks of the keys, last chunk first; every pass is stable
            for (uint32_t ch = nchunks; ch-- > 0;) {
                if (launch_sort_chunk(d_buf, c->table, tt, P, d_klen, pin, ch, kin, st) != hipSuccess ||
                    launch_sort_pairs(d_tmp, tmp_bytes, kin, kout, pin, pout, N, desc, 64, st) != hipSuccess) return fail(BSK_ERR_HIP);
                std::swap(pin, pout);
            }
        } else {
            // long keys (sequences: 19 chunks for 150 bases): the order by the two LEADING chunks first -- 16 key bytes
            // separate nearly all records -- then only the positions whose 16 bytes equal a neighbour's are ordered by the
            // rest of the key (LSD over chunks n-1 .. 2 on that subset, then a stable pass by run number puts every run
            // back in its place).  Same result as the full LSD sweep, ties in file order included.
            rc = ensure_record_scratch(c);  // (the scan scratch)
            if (rc != BSK_OK) return fail(rc);
            for (uint32_t ch = 2; ch-- > 0;) {
                if (launch_sort_chunk(d_buf, c->table, tt, P, d_klen, pin, ch, kin, st) != hipSuccess ||
                    launch_sort_pairs(d_tmp, tmp_bytes, kin, kout, pin, pout, N, desc, 64, st) != hipSuccess) return fail(BSK_ERR_HIP);
                std::swap(pin, pout);
            }
            // kout = chunk 0 in the new order; chunk 1 in that order once more
            if (launch_sort_chunk(d_buf, c->table, tt, P, d_klen, pin, 1, kin, st) != hipSuccess) return fail(BSK_ERR_HIP);
            uint32_t* d_tied = A.at<uint32_t>(o_tied);
            uint32_t* d_start = A.at<uint32_t>(o_start);
            uint64_t* d_rank = A.at<uint64_t>(o_rank);
            uint64_t* d_run = A.at<uint64_t>(o_run);
            uint64_t* d_run_of = A.at<uint64_t>(o_runof);
            uint32_t* d_sub_pos = A.at<uint32_t>(o_subpos);
            uint32_t* sp_in = A.at<uint32_t>(o_subperm);
            uint32_t* sp_out = sp_in + N;
            uint64_t m = 0;
            if (launch_sort_tie_flags(kout, kin, N, d_tied, d_start, st) != hipSuccess ||
                launch_scan_u32(d_tied, d_rank, N, c->d_scan_tmp, st) != hipSuccess ||
                launch_scan_u32(d_start, d_run, N, c->d_scan_tmp, st) != hipSuccess ||
                hipMemcpyAsync(&m, d_rank + N, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) return fail(BSK_ERR_HIP);
            if (m) {
                if (launch_sort_tie_gather(d_tied, d_rank, d_run, d_start, pin, N, d_sub_pos, sp_in, d_run_of, st) != hipSuccess)
                    return fail(BSK_ERR_HIP);
                uint64_t* sk_in = kin;   // the key buffers are free again
                uint64_t* sk_out = kout;
                for (uint32_t ch = nchunks; ch-- > 2;) {
                    if (launch_sort_chunk(d_buf, c->table, tt, P, d_klen, sp_in, ch, sk_in, st, m) != hipSuccess ||
                        launch_sort_pairs(d_tmp, tmp_bytes, sk_in, sk_out, sp_in, sp_out, m, desc, 64, st) != hipSuccess) return fail(BSK_ERR_HIP);
                    std::swap(sp_in, sp_out);
                }
                if (launch_sort_gather_keys(d_run_of, sp_in, m, sk_in, st) != hipSuccess ||
                    launch_sort_pairs(d_tmp, tmp_bytes, sk_in, sk_out, sp_in, sp_out, m, false, 64, st) != hipSuccess ||
                    launch_sort_tie_scatter(d_sub_pos, sp_out, m, pin, st) != hipSuccess) return fail(BSK_ERR_HIP);
            }
        }
    }
    // sizes in file order, offsets in sorted order
    SeqParams F = format_params(c, fastq);
    F.text_w = tt.text_w; F.lin_off = tt.lin_off; F.lin = tt.lin;
    F.buf_end = d_buf + n;
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return fail(rc);
    if (launch_seq_size(d_buf, c->table, F, c->d_out_len, c->d_status, st) != hipSuccess) return fail(BSK_ERR_HIP);
    uint64_t total = 0, kept = 0;
    rc = finish_sizes(c, st, &total, &kept);
    if (rc != BSK_OK) return fail(rc);
    uint32_t* len_perm = pout;  // the other permutation buffer is free now
    uint64_t* off_perm = kin;   // [N + 1] fits: kin and kout are adjacent (2 N entries)
    if (kin != d_keys2) off_perm = d_keys2;
    if (launch_sort_gather(c->d_out_len, pin, N, len_perm, st) != hipSuccess ||
        launch_scan_u32(len_perm, off_perm, N, c->d_scan_tmp, st) != hipSuccess ||
        launch_sort_scatter(off_perm, pin, N, c->d_out_off, st) != hipSuccess) return fail(BSK_ERR_HIP);
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return fail(rc);
    apply_long(c, &F);
    // the offsets follow the SORTED order: the segments of the copy are the records in that order (FASTQ records that leave
    // unchanged; ops_segcopy.hip), the record-wise emit writes what is left
    const char* sg = c->tune.get("segcopy");
    bool seg_done = false;
    if (fastq && !F.ren_ord && !(sg && strcmp(sg, "off") == 0) && ((sg && strcmp(sg, "force") == 0) || total >= (4u << 20))) {
        if (grow(c, &c->d_seg_src, &c->seg_src_cap, 2 * N + 1, N / 4 + 16) != BSK_OK ||
            grow(c, &c->d_seg_first, &c->seg_first_cap, seg_tiles(total) + 1, 64) != BSK_OK) return fail(BSK_ERR_HIP);
        uint64_t* seg_sorted = c->d_seg_src;
        uint64_t* seg_rec = c->d_seg_src + N;
        uint64_t* d_other = c->d_seg_src + 2 * N;
        uint64_t other = 0;
        if (hipMemsetAsync(d_other, 0, sizeof(uint64_t), st) != hipSuccess ||
            launch_seg_build_fastq_perm(d_buf, n, c->table, c->d_out_len, pin, seg_sorted, seg_rec, d_other, st) != hipSuccess ||
            launch_seg_first(off_perm, N, c->d_seg_first, st) != hipSuccess ||
            launch_seg_copy(seg_sorted, off_perm, N, c->d_seg_first, c->d_out, total, d_buf, d_buf + n, st) != hipSuccess ||
            hipMemcpyAsync(&other, d_other, sizeof other, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) return fail(BSK_ERR_HIP);
        if (other == 0) seg_done = true;
        else F.seg_src = seg_rec;
    }
    if (!seg_done && launch_seq_emit(d_buf, c->table, F, c->d_out_len, c->d_out_off, c->d_out, st, total, kept) != hipSuccess) return fail(BSK_ERR_HIP);
    cleanup();
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// faidx index rows: Faidx.Before / Call, bigseqkit-lib/faidx.go:63-229.  PARITY.md FAI.
// ---------------------------------------------------------------------------
// parseRegion (bigseqkit-lib/faidx.go:536-567): "id:b-e", "id:b", "id:b-", "id:-e", else the whole record; the id is
// the shortest prefix whose remainder has one of those shapes (lazy `(.+?)`), tried shape by shape
static void parse_faidx_region(const std::string& region, std::string* id, long long* begin, long long* end) {
    auto number = [](const std::string& t, size_t* at, bool neg_ok, long long* v) {
        size_t i = *at;
        bool neg = false;
        if (neg_ok && i < t.size() && t[i] == '-') { neg = true; ++i; }
        const size_t d0 = i;
        long long x = 0;
        while (i < t.size() && t[i] >= '0' && t[i] <= '9') { x = x * 10 + (t[i] - '0'); ++i; }
        if (i == d0) return false;
        *v = neg ? -x : x;
        *at = i;
        return true;
    };
    for (int shape = 0; shape < 4; ++shape)
        for (size_t c0 = 1; c0 < region.size(); ++c0) {
            if (region[c0] != ':') continue;
            const std::string t = region.substr(c0 + 1);
            size_t at = 0;
            long long b = 0, e = 0;
            bool ok = false;
            if (shape == 0) ok = number(t, &at, true, &b) && at < t.size() && t[at] == '-' && (++at, number(t, &at, true, &e)) && at == t.size();
            else if (shape == 1) { ok = number(t, &at, false, &b) && at == t.size(); e = b; }
            else if (shape == 2) { ok = number(t, &at, true, &b) && at + 1 == t.size() && t[at] == '-'; e = -1; }
            else { ok = !t.empty() && t[0] == '-' && (at = 1, number(t, &at, true, &e)) && at == t.size(); b = 1; }
            if (ok) { *id = region.substr(0, c0); *begin = b; *end = e; return; }
        }
    *id = region;
    *begin = 1;
    *end = -1;
}

void validate_faidx_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    if (!o.b("FullHead")) check_id_regexp(c);  // -f swaps the ID regexp for ^(.+)$ (faidx.go:69-73)
    // region queries (FaidxQuery.Before, faidx.go:246-329): the region file first, then Regions
    c->features.clear();
    c->features_uploaded = false;
    std::vector<std::string> queries;
    if (!o.s("RegionFile").empty())
        for (auto& r : read_pattern_lines(o.s("RegionFile"))) if (!r.empty()) queries.push_back(r);
    for (auto& r : o.sl("Regions")) queries.push_back(r);
    if (queries.empty()) return;
    c->regexes.clear();
    c->patterns_uploaded = false;
    if (o.b("UseRegexp")) {  // :312-319: every query is a regular expression on the ID, the region is the whole record
        for (const std::string& q : queries) {
            try { c->regexes.push_back(compile_regex(q)); }
            catch (const OptError& e) {
                if (std::string(e.what()).rfind("error parsing regexp", 0) == 0) throw OptError("invalid regular expression: " + q);
                throw;
            }
        }
        return;
    }
    std::unordered_set<std::string> seen;
    for (const std::string& q : queries) {
        std::string id;
        long long begin = 1, end = -1;
        parse_faidx_region(q, &id, &begin, &end);
        if (o.b("IgnoreCase")) for (auto& ch : id) if (ch >= 'A' && ch <= 'Z') ch += 32;  // strings.ToLower(id), :323
        if (!seen.insert(id).second) continue;  // the first query of an ID is the one a record meets (:373-380)
        bsk_ctx::Feature f;
        f.name_lower = id;
        const bool whole = (begin == 1 && end == -1) || (begin > 0 && end < 0);  // :388
        f.suffix = whole ? std::string() : ":" + std::to_string(begin) + "-" + std::to_string(end);
        f.minus = !whole && begin > end;  // :403-418, PARITY.md FAI: the region [end, begin], reverse complemented
        f.s = f.minus ? end : begin;
        f.e = f.minus ? begin : end;
        c->features.push_back(f);
    }
}

int faidx_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    int rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    FaidxParams P;
    memset(&P, 0, sizeof P);
    P.fastq = format == BSK_FORMAT_FASTQ;
    P.full_head = o.b("FullHead");
    P.id_mode = id_mode_of(c);
    P.base_offset = c->cur_base_offset;
    P.buf_end = d_buf + n;
    const uint64_t N = c->table.n;
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    Arena A;
    const uint64_t o_lb = A.take(N * 4);
    rc = arena_reserve(c, &A);
    if (rc != BSK_OK) return rc;
    uint32_t* d_lb = A.at<uint32_t>(o_lb);
    auto fail = [&](int code) { return code; };
    if (hipMemsetAsync(c->d_status + 1, 0xFF, 8, st) != hipSuccess ||
        launch_faidx_size(d_buf, c->table, P, c->d_out_len, d_lb, c->d_status, st) != hipSuccess ||
        launch_scan_u32(c->d_out_len, c->d_out_off, N, c->d_scan_tmp, st) != hipSuccess) return fail(BSK_ERR_HIP);
    uint64_t total = 0, status[2] = {0, 0};
    if (hipMemcpyAsync(&total, c->d_out_off + N, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(status, c->d_status, 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return fail(BSK_ERR_HIP);
    if (status[0] & ERR_LINE_LENGTHS) {
        // the first offending record names the error (faidx.go:131)
        uint64_t start = 0;
        uint32_t lh = 0;
        hipMemcpy(&start, c->table.start + status[1], 8, hipMemcpyDeviceToHost);
        hipMemcpy(&lh, c->table.l_head + status[1], 4, hipMemcpyDeviceToHost);
        std::string head(lh > 0 ? lh - 1 : 0, '\0');
        if (!head.empty()) hipMemcpy(&head[0], d_buf + start + 1, head.size(), hipMemcpyDeviceToHost);
        std::string id = head;
        if (!P.full_head && P.id_mode == 0) {
            size_t sp = head.find(' ');
            if (sp != std::string::npos && sp > 0) id = head.substr(0, sp);
            else { sp = head.find('\t'); if (sp != std::string::npos && sp > 0) id = head.substr(0, sp); }
        } else if (!P.full_head) {  // --id-ncbi: first match of \|([^\|]+)\|<space>
            for (size_t i = 0; i < head.size(); ++i) {
                if (head[i] != '|') continue;
                size_t j = i + 1;
                while (j < head.size() && head[j] != '|') ++j;
                if (j > i + 1 && j + 1 < head.size() && head[j + 1] == ' ') { id = head.substr(i + 1, j - i - 1); break; }
            }
        }
        c->set_error("different line length in sequence: " + id + ". Please format the file with 'seqkit seq'");
        return fail(BSK_ERR_FORMAT);
    }
    rc = kernel_error_to_status(c, status[0]);
    if (rc != BSK_OK) return fail(rc);
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return fail(rc);
    if (launch_faidx_rows(d_buf, c->table, P, d_lb, c->d_out_off, c->d_out, st) != hipSuccess) return fail(BSK_ERR_HIP);
    out->d_data = c->d_out;
    out->len = total;
    out->records = N;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// pair: PairPrepare x2 + Union + GroupByKey + Pair (bigseqkit/pair.go:34-100,
// bigseqkit-lib/pair.go:37-121).  The shard is file 1 followed by file 2; the k-th record of an ID in file 1 is paired
// with the k-th of file 2.  outs[0] / outs[1]: the pairs, both in the file-1 order of their first mates; outs[2] /
// outs[3]: the records without a mate (SaveUnpaired), file order.  PARITY.md PAIR.
// ---------------------------------------------------------------------------
int pair_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, size_t n_first, int format, hipStream_t st, bsk_out* outs) {
    const Options& o = c->opts;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    for (int k = 0; k < 4; ++k) { outs[k].d_data = nullptr; outs[k].len = 0; outs[k].records = 0; }
    int rc = build_index(c, d_buf, n, format, st);
    if (rc == BSK_OK) rc = check_u32_records(c, "pair");  // (group << 32 | index) keys, u32 permutations
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) {
        bsk_out tmp;
        return empty_result(c, &tmp);
    }
    TextTableH tt;
    rc = prepare_text(c, d_buf, format, st, &tt);
    if (rc != BSK_OK) return rc;
    const uint64_t N = c->table.n;
    RmDupParams P;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    P.id_mode = id_mode_of(c);
    P.line_width = fastq ? 0 : (int)o.ci("LineWidth");
    P.buf_end = d_buf + n;
    rc = grow(c, &c->d_keys, &c->keys_cap, N, N / 8 + 16);
    if (rc != BSK_OK) return rc;
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    size_t tmp_bytes = 0;
    if (group_sort_temp_bytes(N, &tmp_bytes) != hipSuccess) { c->set_error("libbsk: rocPRIM sort size query failed"); return BSK_ERR_HIP; }
    Arena A;
    const uint64_t o_has = A.take(N), o_list = A.take(2 * N * 8), o_tmp = A.take(tmp_bytes ? tmp_bytes : 16),
                   o_state = A.take(N), o_partner = A.take(N * 4), o_fmt = A.take(N * 4), o_len = A.take(N * 4),
                   o_off = A.take((N + 1) * 8), o_offw = A.take((N + 1) * 8), o_tot = A.take(8 * 8);
    rc = arena_reserve(c, &A);
    if (rc != BSK_OK) return rc;
    uint8_t* d_has = A.at<uint8_t>(o_has);
    uint64_t* d_list = A.at<uint64_t>(o_list);
    uint8_t* d_state = A.at<uint8_t>(o_state);
    uint32_t* d_partner = A.at<uint32_t>(o_partner);
    uint32_t* d_fmt = A.at<uint32_t>(o_fmt);
    uint32_t* d_len = A.at<uint32_t>(o_len);
    uint64_t* d_off = A.at<uint64_t>(o_off);
    uint64_t* d_offw = A.at<uint64_t>(o_offw);
    uint64_t* d_tot = A.at<uint64_t>(o_tot);
    // groups by ID (XXH64, first occurrence, exact verification of every other member)
    HIP_TRYX(c, launch_rmdup_hash(d_buf, n, c->table, tt, P, c->d_keys, nullptr, st));
    HIP_TRYX(c, hipMemsetAsync(d_has, 0, N, st));
    rc = group_resolve(c, d_buf, tt, P, d_has, st);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, hipMemsetAsync(d_tot, 0, 8 * 8, st));
    HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_count_below(c->table.start, N, n_first, c->d_counter, st));
    uint64_t first2 = 0, status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&first2, c->d_counter, 8, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, 8, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status & ERR_HASH_COLLISION) {
        c->set_error("libbsk: two distinct subjects share one 64-bit XXH64 key; refusing to guess (rerun on the CPU path)");
        return BSK_ERR_UNSUPPORTED;
    }
    rc = kernel_error_to_status(c, status);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, launch_group_all(c->d_keys, N, d_list, st));
    HIP_TRYX(c, launch_group_sort(A.at<uint8_t>(o_tmp), tmp_bytes, d_list, d_list + N, N, st, N, /*index_ordered=*/true));
    HIP_TRYX(c, launch_pair_classify(d_list + N, N, (uint32_t)first2, d_state, d_partner, st));
    // formatted size of every record, totals per output
    SeqParams F = format_params(c, fastq);
    F.text_w = tt.text_w; F.lin_off = tt.lin_off; F.lin = tt.lin;
    F.buf_end = d_buf + n;
    HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 2 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_seq_size(d_buf, c->table, F, d_fmt, c->d_status, st));
    HIP_TRYX(c, launch_pair_totals(d_state, d_fmt, N, d_tot, st));
    uint64_t tot[8];
    HIP_TRYX(c, hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, 8, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    rc = kernel_error_to_status(c, status);
    if (rc != BSK_OK) return rc;
    const bool unpaired = o.b("SaveUnpaired");
    uint64_t base[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) base[k + 1] = base[k] + ((k < 2 || unpaired) ? ((tot[k] + 255) & ~255ull) : 0);
    rc = ensure_out(c, base[4]);
    if (rc != BSK_OK) return rc;
    for (int k = 0; k < 4; ++k) {
        if (!(k < 2 || unpaired) || tot[k] == 0) continue;
        HIP_TRYX(c, launch_pair_select(d_state, d_fmt, N, (uint8_t)(k + 1), d_len, st));
        if (k == 1) {
            // the second mates follow the order of the first ones
            uint32_t* d_w = c->d_out_len;  // free scratch of N entries
            HIP_TRYX(c, launch_pair_partner_len(d_state, d_partner, d_fmt, N, d_w, st));
            HIP_TRYX(c, launch_scan_u32(d_w, d_offw, N, c->d_scan_tmp, st));
            HIP_TRYX(c, launch_pair_partner_off(d_state, d_partner, d_offw, N, d_off, st));
        } else {
            HIP_TRYX(c, launch_scan_u32(d_len, d_off, N, c->d_scan_tmp, st));
        }
        if (k == 1) {  // offsets in the order of the first mates: the record-wise emit
            HIP_TRYX(c, launch_seq_emit(d_buf, c->table, F, d_len, d_off, c->d_out + base[k], st, tot[k], tot[4 + k]));
        } else {
            rc = emit_records_at(c, d_buf, n, F, d_len, d_off, c->d_out + base[k], tot[k], tot[4 + k], st);
            if (rc != BSK_OK) return rc;
        }
        outs[k].d_data = c->d_out + base[k];
        outs[k].len = tot[k];
        outs[k].records = tot[4 + k];
    }
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// common: the records of the first file whose key occurs in every file.  The reference's
// CommonPrepare / CommonJoin (bigseqkit-lib/common.go:31-212) cannot run as written; PARITY.md COMMON states what is
// kept (keys, options, error texts) and what follows seqkit's documented behaviour instead.
// ---------------------------------------------------------------------------
void validate_common_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    if (o.b("BySeq") && o.b("ByName"))  // common.go:37-39
        throw OptError("only one/none of the flags -s (--by-seq) and -n (--by-name) is allowed");
    if (o.b("OnlyPositiveStrand") && !o.b("BySeq"))  // :43-45
        throw OptError("flag -s (--by-seq) needed when using -P (--only-positive-strand)");
    // -s -P: as written nothing is hashed on that branch (every record would get key 0); the flag's meaning -- compare the
    // sequences as they are -- is what -s already does here (PARITY.md COMMON), so it is accepted and changes nothing
}

int common_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, const uint64_t* file_ends, uint32_t nfiles, int format,
                      hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    int rc = build_index(c, d_buf, n, format, st);
    if (rc == BSK_OK) rc = check_u32_records(c, "common");  // (group << 32 | index) keys, u32 permutations
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    TextTableH tt;
    rc = prepare_text(c, d_buf, format, st, &tt);
    if (rc != BSK_OK) return rc;
    const uint64_t N = c->table.n;
    RmDupParams P;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    P.by_seq = o.b("BySeq");
    P.by_name = o.b("ByName");
    P.ignore_case = o.b("IgnoreCase");
    P.id_mode = id_mode_of(c);
    P.line_width = fastq ? 0 : (int)o.ci("LineWidth");
    P.buf_end = d_buf + n;
    rc = grow(c, &c->d_keys, &c->keys_cap, N, N / 8 + 16);
    if (rc != BSK_OK) return rc;
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    Arena A;
    const uint64_t o_has = A.take(N), o_masks = A.take(N * 8), o_ends = A.take((uint64_t)nfiles * 8);
    rc = arena_reserve(c, &A);
    if (rc != BSK_OK) return rc;
    uint8_t* d_has = A.at<uint8_t>(o_has);
    uint64_t* d_masks = A.at<uint64_t>(o_masks);
    uint64_t* d_ends = A.at<uint64_t>(o_ends);
    HIP_TRYX(c, hipMemcpyAsync(d_ends, file_ends, (size_t)nfiles * 8, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, launch_rmdup_hash(d_buf, n, c->table, tt, P, c->d_keys, nullptr, st));
    HIP_TRYX(c, hipMemsetAsync(d_has, 0, N, st));
    rc = group_resolve(c, d_buf, tt, P, d_has, st);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, hipMemsetAsync(d_masks, 0, N * 8, st));
    HIP_TRYX(c, launch_common_masks(c->d_keys, c->table.start, N, d_ends, nfiles, d_masks, st));
    uint64_t status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, 8, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));  // (file_ends is the caller's memory)
    if (status & ERR_HASH_COLLISION) {
        c->set_error("libbsk: two distinct subjects share one 64-bit XXH64 key; refusing to guess (rerun on the CPU path)");
        return BSK_ERR_UNSUPPORTED;
    }
    rc = kernel_error_to_status(c, status);
    if (rc != BSK_OK) return rc;
    SeqParams F = format_params(c, fastq);
    F.text_w = tt.text_w; F.lin_off = tt.lin_off; F.lin = tt.lin;
    F.buf_end = d_buf + n;
    HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 2 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_seq_size(d_buf, c->table, F, c->d_out_len, c->d_status, st));
    HIP_TRYX(c, launch_common_select(c->d_keys, c->table.start, N, d_ends, nfiles, d_masks, c->d_out_len, st));
    uint64_t total = 0, kept = 0;
    rc = finish_sizes(c, st, &total, &kept);
    if (rc != BSK_OK) return rc;
    out->d_data = nullptr;
    out->len = 0;
    out->records = 0;
    if (total == 0) return BSK_OK;
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    apply_long(c, &F);
    { const int rce = emit_records(c, d_buf, n, F, total, kept, st); if (rce != BSK_OK) return rce; }
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// concat: ConcatPrepare x2 + Union + GroupByKey + ConcatJoin (bigseqkit/concat.go:41-90,
// bigseqkit-lib/concat.go:39-165).  PARITY.md CONCAT.
// ---------------------------------------------------------------------------
int concat_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, size_t n_first, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    int rc = build_index(c, d_buf, n, format, st);
    if (rc == BSK_OK) rc = check_u32_records(c, "concat");  // (group << 32 | index) keys, u32 permutations
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    TextTableH tt;
    rc = prepare_text(c, d_buf, format, st, &tt);
    if (rc != BSK_OK) return rc;
    const uint64_t N = c->table.n;
    RmDupParams P;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    P.id_mode = id_mode_of(c);
    P.line_width = fastq ? 0 : (int)o.ci("LineWidth");
    P.buf_end = d_buf + n;
    rc = grow(c, &c->d_keys, &c->keys_cap, N, N / 8 + 16);
    if (rc != BSK_OK) return rc;
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    size_t tmp_bytes = 0;
    if (group_sort_temp_bytes(N, &tmp_bytes) != hipSuccess) { c->set_error("libbsk: rocPRIM sort size query failed"); return BSK_ERR_HIP; }
    Arena A;
    const uint64_t o_has = A.take(N), o_list = A.take(2 * N * 8), o_tmp = A.take(tmp_bytes ? tmp_bytes : 16),
                   o_seg = A.take(3 * N * 4), o_cnt = A.take(N * 4), o_cntoff = A.take((N + 1) * 8);
    rc = arena_reserve(c, &A);
    if (rc != BSK_OK) return rc;
    uint8_t* d_has = A.at<uint8_t>(o_has);
    uint64_t* d_list = A.at<uint64_t>(o_list);
    uint32_t* d_seg = A.at<uint32_t>(o_seg);
    uint32_t* d_cnt = A.at<uint32_t>(o_cnt);
    uint64_t* d_cntoff = A.at<uint64_t>(o_cntoff);
    HIP_TRYX(c, launch_rmdup_hash(d_buf, n, c->table, tt, P, c->d_keys, nullptr, st));
    HIP_TRYX(c, hipMemsetAsync(d_has, 0, N, st));
    rc = group_resolve(c, d_buf, tt, P, d_has, st);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_count_below(c->table.start, N, n_first, c->d_counter, st));
    uint64_t first2 = 0, status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&first2, c->d_counter, 8, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, 8, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status & ERR_HASH_COLLISION) {
        c->set_error("libbsk: two distinct subjects share one 64-bit XXH64 key; refusing to guess (rerun on the CPU path)");
        return BSK_ERR_UNSUPPORTED;
    }
    rc = kernel_error_to_status(c, status);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, launch_group_all(c->d_keys, N, d_list, st));
    HIP_TRYX(c, launch_group_sort(A.at<uint8_t>(o_tmp), tmp_bytes, d_list, d_list + N, N, st, N, /*index_ordered=*/true));
    HIP_TRYX(c, launch_concat_segments(d_list + N, N, (uint32_t)first2, d_seg, st));
    ConcatParams Q;
    memset(&Q, 0, sizeof Q);
    Q.fastq = fastq;
    Q.full = o.b("Full");
    Q.id_mode = P.id_mode;
    Q.line_width = P.line_width;
    Q.first2 = (uint32_t)first2;
    Q.buf_end = d_buf + n;
    HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 2 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_concat_size(d_buf, c->table, Q, d_list + N, d_seg, c->d_out_len, d_cnt, c->d_status, st));
    HIP_TRYX(c, launch_scan_u32(c->d_out_len, c->d_out_off, N, c->d_scan_tmp, st));
    HIP_TRYX(c, launch_scan_u32(d_cnt, d_cntoff, N, c->d_scan_tmp, st));
    uint64_t total = 0, elements = 0;
    HIP_TRYX(c, hipMemcpyAsync(&total, c->d_out_off + N, 8, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&elements, d_cntoff + N, 8, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, 8, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    rc = kernel_error_to_status(c, status);
    if (rc != BSK_OK) return rc;
    out->d_data = nullptr;
    out->len = 0;
    out->records = 0;
    if (total == 0) return BSK_OK;
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    const uint8_t* seg_done = nullptr;
    bool emit_old = true;
    {
        // FASTQ: the elements are slices of the shard -- the segmented copy writes them (ops_segcopy.hip, k_concat_segs)
        const char* sg = c->tune.get("segcopy");
        const bool force = sg && strcmp(sg, "force") == 0;
        if (fastq && elements > 0 && !(sg && strcmp(sg, "off") == 0) && (force || total >= (4u << 20))) {
            const uint64_t ns = 5 * elements;
            rc = grow(c, &c->d_seg_src, &c->seg_src_cap, 2 * ns + 2 + (N + 7) / 8 + 1, ns / 4 + 16);
            if (rc != BSK_OK) return rc;
            rc = grow(c, &c->d_seg_first, &c->seg_first_cap, seg_tiles(total) + 1, 64);
            if (rc != BSK_OK) return rc;
            uint64_t* seg_src = c->d_seg_src;
            uint64_t* seg_off2 = c->d_seg_src + ns;  // [ns + 1]
            uint64_t* d_other = c->d_seg_src + 2 * ns + 1;
            uint8_t* d_done = reinterpret_cast<uint8_t*>(c->d_seg_src + 2 * ns + 2);
            HIP_TRYX(c, hipMemsetAsync(d_other, 0, sizeof(uint64_t), st));
            HIP_TRYX(c, hipMemcpyAsync(seg_off2 + ns, &total, sizeof total, hipMemcpyHostToDevice, st));
            HIP_TRYX(c, launch_concat_segs(d_buf, n, c->table, Q, d_list + N, d_seg, c->d_out_len, c->d_out_off, d_cntoff, seg_src,
                                           seg_off2, d_done, d_other, st));
            HIP_TRYX(c, launch_seg_first(seg_off2, ns, c->d_seg_first, st));
            HIP_TRYX(c, launch_seg_copy(seg_src, seg_off2, ns, c->d_seg_first, c->d_out, total, d_buf, d_buf + n, st));
            uint64_t other = 0;
            HIP_TRYX(c, hipMemcpyAsync(&other, d_other, sizeof other, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            seg_done = d_done;
            emit_old = other != 0;
        }
    }
    if (emit_old)
        HIP_TRYX(c, launch_concat_emit(d_buf, c->table, tt, Q, d_list + N, d_seg, c->d_out_len, c->d_out_off, c->d_out,
                                       elements ? total / elements : 0, st, seg_done));
    out->d_data = c->d_out;
    out->len = total;
    out->records = elements;
    return BSK_OK;
}

// FaidxQuery.Call (bigseqkit-lib/faidx.go:331-432): every record whose ID has a query comes back as FASTA --
// ">ID" or ">ID:b-e" and the region, wrapped at LineWidth; PARITY.md FAI
int faidx_query_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const bool fastq = format == BSK_FORMAT_FASTQ;
    int rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0 || (c->features.empty() && c->regexes.empty())) return empty_result(c, out);
    SeqParams P = format_params(c, fastq);
    P.print_qual = 0;
    P.fasta_out = 1;
    P.line_width = (int)c->opts.ci("LineWidth");
    P.buf_end = d_buf + n;
    if (!c->regexes.empty()) {
        // -r: hits = records whose ID matches one of the expressions (the grep -r automaton), printed as ">ID" + sequence
        const uint64_t N = c->table.n;
        if (!c->patterns_uploaded) {
            rc = grow(c, &c->d_regex, &c->regex_cap, c->regexes.size());
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemcpyAsync(c->d_regex, c->regexes.data(), c->regexes.size() * sizeof(RegexProgram),
                                       hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            c->patterns_uploaded = true;
        }
        TextTableH tt{nullptr, nullptr, nullptr};
        rc = prepare_text(c, d_buf, format, st, &tt);
        if (rc != BSK_OK) return rc;
        P.text_w = tt.text_w; P.lin_off = tt.lin_off; P.lin = tt.lin;
        P.only_id = 1;
        P.min_len = 1;  // SubLocation of an empty sequence is not ok: the record is skipped (faidx.go:391-395)
        rc = ensure_record_scratch(c);
        if (rc != BSK_OK) return rc;
        Arena A;
        const uint64_t o_hits = A.take(N * 4);
        rc = arena_reserve(c, &A);
        if (rc != BSK_OK) return rc;
        uint32_t* d_hits = A.at<uint32_t>(o_hits);
        GrepParams G;
        memset(&G, 0, sizeof G);
        G.fastq = fastq;
        G.id_mode = P.id_mode;
        G.line_width = P.line_width;
        G.npat = (int)c->regexes.size();
        G.regex = c->d_regex;
        HIP_TRYX(c, launch_grep_match(d_buf, n, c->table, &tt, G, d_hits, st));
        HIP_TRYX(c, launch_seq_size(d_buf, c->table, P, c->d_out_len, c->d_status, st));
        HIP_TRYX(c, launch_mask_u32(c->d_out_len, d_hits, N, st));
        uint64_t total = 0, kept = 0;
        rc = finish_sizes(c, st, &total, &kept);
        if (rc != BSK_OK) return rc;
        rc = ensure_out(c, total);
        if (rc != BSK_OK) return rc;
        apply_long(c, &P);
        { const int rce = emit_records(c, d_buf, n, P, total, kept, st); if (rce != BSK_OK) return rce; }
        out->d_data = c->d_out;
        out->len = total;
        out->records = kept;
        return BSK_OK;
    }
    rc = bind_features(c, d_buf, n, format, st, &P);
    if (rc != BSK_OK) return rc;
    P.feat_query = 1;
    P.feat_fold = c->opts.b("IgnoreCase");
    TextTableH tt{nullptr, nullptr, nullptr};
    rc = prepare_text(c, d_buf, format, st, &tt);
    if (rc != BSK_OK) return rc;
    P.text_w = tt.text_w; P.lin_off = tt.lin_off; P.lin = tt.lin;
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, launch_seq_size(d_buf, c->table, P, c->d_out_len, c->d_status, st));
    uint64_t total = 0, kept = 0;
    rc = finish_sizes(c, st, &total, &kept);
    if (rc != BSK_OK) return rc;
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    apply_long(c, &P);
    { const int rce = emit_records(c, d_buf, n, P, total, kept, st); if (rce != BSK_OK) return rce; }
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    return BSK_OK;
}

}  // namespace bsk
