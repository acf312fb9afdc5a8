// Host side of the record-table operators: index construction and `seq`
// (SeqTransform, /root/reference/bigseqkit-lib/seq.go).  C-ABI in include/bsk.h.
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>
#include <cerrno>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
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
#include "ops_locate.hpp"
#include "ops_mlfq.hpp"
#include "ops_records.hpp"
#include "ops_rmdup.hpp"
#include "ops_text.hpp"
#include "ops_translate.hpp"
#include "ops_segcopy.hpp"
#include "ops_seq.hpp"
#include "ops_sort.hpp"
#include "stream_fasta_light.hpp"
#include "stream_filter.hpp"
#include "stream_names.hpp"
#include "stream_subseq.hpp"
#include "stream_rmdup.hpp"
#include "stream_stats.hpp"

namespace bsk {

// Open-addressing table of the key-grouping operators (rmdup, rename, pair, common, concat, grep --delete-matched):
// d_keys for N records, `cap` slots (a power of two >= 2 N) of keys (zeroed) and first-record indices (0xFF-filled).
int key_table(bsk_ctx* c, uint64_t N, uint64_t* cap_out, uint64_t** tk, hipStream_t st) {
    int rc = grow(c, &c->d_keys, &c->keys_cap, N, N / 8 + 16);
    if (rc != BSK_OK) return rc;
    uint64_t cap = 1024;
    while (cap < 2 * N) cap <<= 1;
    if (2 * cap > c->table_cap || !c->d_table) {
        if (c->d_table) HIP_TRYX(c, hipFree(c->d_table));
        c->d_table = nullptr;
        HIP_TRYX(c, hipMalloc((void**)&c->d_table, 2 * cap * sizeof(uint64_t)));
        c->table_cap = 2 * cap;
    }
    *tk = c->d_table;
    *cap_out = cap;
    HIP_TRYX(c, hipMemsetAsync(*tk, 0, 2 * cap * sizeof(uint64_t), st));  // slot = {key, ~first}: all zero = empty
    return BSK_OK;
}

static void complement_table(Alphabet ab, uint8_t m[256]);

// Groups of equal subjects for rename / pair / common / concat.  In: c->d_keys[i] = XXH64 of the subject of record i.
// Out: c->d_keys[i] = first record of i's group, d_has[first] = 1 for the groups of two or more, c->d_out_len[i] = formatted
// size of record i if it is the first of its group (else 0); every later member is byte-compared with the first
// (ERR_HASH_COLLISION in the status word).  Radix buckets + one LDS table per bucket (ops_rmdup.hip) out of a scratch of
// its own -- the callers hold the arena --, the one big table in HBM when a bucket overflows, with BSK_RMDUP=table, or
// from 2^32 records.
int group_resolve(bsk_ctx* c, const uint8_t* d_buf, const TextTableH& tt, const RmDupParams& P, uint8_t* d_has, hipStream_t st) {
    const uint64_t N = c->table.n;
    bool by_buckets = N < (1ull << 32);
    {
        const char* e = c->tune.get("rmdup");
        if (e && strcmp(e, "table") == 0) by_buckets = false;
    }
    if (by_buckets) {
        size_t tmp_bytes = 0;
        HIP_TRYX(c, sort_pairs_bits_temp_bytes(N, 0, 16, &tmp_bytes));
        Arena A;  // (used for its offset arithmetic only: the memory is c->d_group)
        const uint64_t o_sk = A.take(N * 8), o_vi = A.take(N * 4), o_vo = A.take(N * 4), o_first = A.take(N * 4),
                       o_bs = A.take((65536 + 2) * 4), o_hist = A.take(65536 * 4), o_tmp = A.take(tmp_bytes + 256);
        int rc = grow(c, &c->d_group, &c->group_cap, A.used, A.used / 8 + 256);
        if (rc != BSK_OK) return rc;
        A.base = c->d_group;
        uint64_t* d_sk = A.at<uint64_t>(o_sk);
        uint32_t* d_vi = A.at<uint32_t>(o_vi);
        uint32_t* d_vo = A.at<uint32_t>(o_vo);
        uint32_t* d_first = A.at<uint32_t>(o_first);
        if (!c->tune.is("rmdup_buckets", "hand")) {  // the device radix sort of the pairs (two 8-bit digit passes: 1.5 ms per 79 M pairs)
            HIP_TRYX(c, launch_sort_iota(d_vi, N, st));
            HIP_TRYX(c, launch_sort_iota(d_first, N, st));
            HIP_TRYX(c, launch_sort_pairs_bits(A.at<uint8_t>(o_tmp), tmp_bytes, c->d_keys, d_sk, d_vi, d_vo, N, 0, 16, st));
            HIP_TRYX(c, launch_bucket_dedupe(d_sk, d_vo, N, A.at<uint32_t>(o_bs), d_first, c->d_status, st));
        } else {  // one 16-bit histogram + scatter by hand (ops_rmdup.hip): 6.8 ms -- kept for the comparison
            HIP_TRYX(c, launch_bucket_pass(c->d_keys, N, A.at<uint32_t>(o_hist), A.at<uint32_t>(o_bs), d_first, d_sk, d_vo, st));
            HIP_TRYX(c, launch_bucket_dedupe(d_sk, d_vo, N, A.at<uint32_t>(o_bs), d_first, c->d_status, st, nullptr, nullptr, 0, true));
        }
        uint64_t status = 0;
        HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (!(status & ERR_BUCKET_OVERFLOW)) {
            HIP_TRYX(c, launch_rmdup_resolve_first(d_buf, c->table, tt, P, d_first, c->d_keys, c->d_out_len, c->d_status, d_has, st));
            return BSK_OK;
        }
        status &= ~(uint64_t)ERR_BUCKET_OVERFLOW;
        HIP_TRYX(c, hipMemcpyAsync(c->d_status, &status, sizeof status, hipMemcpyHostToDevice, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
    }
    uint64_t cap = 0;
    uint64_t* tk = nullptr;
    int rc = key_table(c, N, &cap, &tk, st);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, launch_rmdup_insert(c->d_keys, N, 0, tk, cap, st));
    HIP_TRYX(c, launch_rmdup_resolve_group(d_buf, c->table, tt, P, c->d_keys, tk, cap, c->d_out_len, c->d_status, d_has, st));
    return BSK_OK;
}



int kernel_error_to_status(bsk_ctx* c, uint64_t f) {
    c->last_kernel_flags = f;
    if (!f) return BSK_OK;
    int code = BSK_ERR_FORMAT;
    std::string m;
    if (f & ERR_BAD_HEADER) {
        code = BSK_ERR_UNSUPPORTED;
        m = "record does not start with '>' / '@' at a line start (leading blank lines, multi-line FASTQ and "
            "blank lines between records are not accepted by the HIP path)";
    } else if (f & ERR_BAD_PLUS) {
        code = BSK_ERR_UNSUPPORTED;
        m = "FASTQ is not in the strict 4-line layout (third line must start with '+')";
    } else if (f & ERR_LEN_MISMATCH) m = "unmatched length of sequence and quality";
    else if (f & ERR_TRUNCATED) m = "FASTQ ends inside a record";
    else if (f & ERR_ANCHOR) {
        code = BSK_ERR_UNSUPPORTED;
        m = "FASTQ is not in the strict 4-line layout (a range did not end on a record boundary)";
    } else if (f & ERR_LINE_TOO_LONG) {
        code = BSK_ERR_UNSUPPORTED;
        m = "a line longer than 2^31 bytes (or a FASTA record longer than 2^32 bytes)";
    } else if (f & ERR_INVALID_LETTER) m = "seq: invalid letter for the sequence alphabet";
    else if (f & ERR_LINE_LENGTHS) m = "different line length in sequence";  // the caller adds the ID
    else if (f & ERR_RECORD_TOO_LARGE) { code = BSK_ERR_UNSUPPORTED; m = "duplicate: the copies of one record exceed 4 GiB"; }
    else if (f & ERR_CAPACITY) { code = BSK_ERR_CAPACITY; m = "libbsk: internal table capacity exceeded"; }
    else if (f & ERR_HASH_COLLISION) {
        code = BSK_ERR_UNSUPPORTED;
        m = "libbsk: two distinct subjects share one 64-bit XXH64 key; refusing to guess (rerun on the CPU path)";
    } else {
        char hex[32];
        snprintf(hex, sizeof hex, "0x%llx", (unsigned long long)f);
        m = std::string("unknown kernel error (flags ") + hex + ")";
    }
    c->set_error(m);
    return code;
}

// ---------------------------------------------------------------------------
// record table of one device-resident shard (count pass, scan, write pass)
// ---------------------------------------------------------------------------
int build_index(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st) {
    return build_index_filtered(c, d_buf, n, format, st, nullptr);
}

// ranges of a streaming pass over the shard: anchors[nranges + 1] (+ the queue word behind them) in ctx-owned memory.
// FASTA ranges begin on line starts (a chromosome spans many ranges); the records that cross range boundaries are
// completed by k_index_stitch from the per-range parts
static int prep_ranges(bsk_ctx* c, const uint8_t* d_buf, size_t n, bool fastq, int blocks, hipStream_t st, uint32_t* nranges_out,
                       uint64_t* chunk_out) {
    const uint64_t waves = (uint64_t)blocks * 4;
    const uint64_t nr = pick_nranges(n, waves, c->min_range_bytes, (int)c->tune.num("ranges_per_wave"));
    const uint32_t nranges = (uint32_t)nr;
    uint64_t chunk = (n + nranges - 1) / nranges;
    chunk = (chunk + 15) & ~(uint64_t)15;
    if (nranges > c->cap_ranges || !c->d_anchors || !c->d_range_count) {
        if (c->d_anchors) HIP_TRYX(c, hipFree(c->d_anchors));
        if (c->d_range_count) HIP_TRYX(c, hipFree(c->d_range_count));
        if (c->d_range_base) HIP_TRYX(c, hipFree(c->d_range_base));
        c->d_anchors = nullptr; c->d_range_count = nullptr; c->d_range_base = nullptr;
        HIP_TRYX(c, hipMalloc((void**)&c->d_anchors, 2 * ((size_t)nranges + 2) * sizeof(uint64_t)));  // (+ k_prep's raw anchors)
        HIP_TRYX(c, hipMalloc((void**)&c->d_range_count, ((size_t)nranges + 1) * sizeof(uint64_t)));
        HIP_TRYX(c, hipMalloc((void**)&c->d_range_base, ((size_t)nranges + 2) * sizeof(uint64_t)));
        c->cap_ranges = nranges;
    }
    uint32_t* queue = reinterpret_cast<uint32_t*>(c->d_anchors + (size_t)nranges + 1);
    HIP_TRYX(c, launch_prep(fastq, d_buf, n, chunk, nranges, c->d_anchors, queue, st, /*line_mode=*/!fastq,
                            /*raw=*/fastq ? nullptr : c->d_anchors + (size_t)nranges + 2));
    *nranges_out = nranges;
    *chunk_out = chunk;
    return BSK_OK;
}

// With a FilterDev (FASTQ only): the table holds ONLY the records whose sequence line contains one of the filter's
// patterns (or only the others, with invert) -- stream_filter.hip.  BSK_ERR_FILTER_FALLBACK: the filter gave up
// (pending-hit list full); the caller then takes the unfiltered path.
int build_index_filtered(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, const FilterDev* F) {
    return build_index_ex(c, d_buf, n, format, st, F, nullptr);
}

// hash != null (FASTQ): the same pass also leaves the two keys of every record's sequence (hash_dev.hpp) in c->d_keys /
// c->d_keys2 (stream_rmdup.hip); hash->fold: keys of the lower-cased sequence (-i)
int build_index_ex(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, const FilterDev* F, const HashReq* hash) {
    const bool fastq = format == BSK_FORMAT_FASTQ;
    c->table.n = 0;
    c->avg_record_bytes = 0;
    if (n == 0) return BSK_OK;
    if (F && !fastq) { c->set_error("libbsk: the pattern filter runs on FASTQ only"); return BSK_ERR_INVALID_ARG; }
    if (hash && (!fastq || F)) { c->set_error("libbsk: the hashing pass runs on unfiltered FASTQ only"); return BSK_ERR_INVALID_ARG; }
    HashDev HD{nullptr, nullptr};
    uint64_t chunk = 0;  // nominal bytes per range (prep_ranges)
    auto launch_pass = [&](int blocks, const uint64_t* anchors, uint32_t nranges, uint32_t* queue, const IndexDev& D) -> hipError_t {
        if (F) return launch_filter(c->use_dpp, blocks, d_buf, n, anchors, nranges, queue, D, *F, st);
        if (hash) return launch_rmdup_stream(c->use_dpp, hash->fold, blocks, d_buf, n, anchors, nranges, queue, D, HD, st);
        return launch_index(fastq, c->use_dpp, blocks, d_buf, n, anchors, nranges, queue, D, st, fastq ? 0 : chunk);
    };
    const int per_cu = F ? filter_max_blocks_per_cu(c->use_dpp)
                         : (hash ? rmdup_stream_max_blocks_per_cu(c->use_dpp, hash->fold) : index_max_blocks_per_cu(fastq, c->use_dpp));
    const int blocks = std::max(1, c->num_cus * per_cu);
    uint32_t nranges = 0;
    int rcp = prep_ranges(c, d_buf, n, fastq, blocks, st, &nranges, &chunk);
    if (rcp != BSK_OK) return rcp;
    uint64_t* anchors = c->d_anchors;
    uint32_t* queue = reinterpret_cast<uint32_t*>(c->d_anchors + (size_t)nranges + 1);
    IndexDev D;
    D.parts = nullptr;
    if (!fastq) {
        if (nranges > c->parts_cap || !c->d_parts) {
            if (c->d_parts) HIP_TRYX(c, hipFree(c->d_parts));
            c->d_parts = nullptr;
            HIP_TRYX(c, hipMalloc((void**)&c->d_parts, (size_t)nranges * sizeof(RangePart)));
            c->parts_cap = nranges;
        }
        HIP_TRYX(c, hipMemsetAsync(c->d_parts, 0, (size_t)nranges * sizeof(RangePart), st));
        D.parts = c->d_parts;
    }
    D.range_count = c->d_range_count;
    D.range_base = c->d_range_base;
    D.status = c->d_status;
    D.sparse_cap = 0;
    auto alloc_table = [&](RecordTable& t, uint64_t cap) -> int {
        if (cap <= t.cap && t.start) return BSK_OK;
        for (void* p : {(void*)t.start, (void*)t.l_head, (void*)t.l_seq, (void*)t.aux, (void*)t.text_w})
            if (p) HIP_TRYX(c, hipFree(p));
        t = RecordTable();
        HIP_TRYX(c, hipMalloc((void**)&t.start, (cap + 1) * sizeof(uint64_t)));
        HIP_TRYX(c, hipMalloc((void**)&t.l_head, cap * sizeof(uint32_t)));
        HIP_TRYX(c, hipMalloc((void**)&t.l_seq, cap * sizeof(uint32_t)));
        HIP_TRYX(c, hipMalloc((void**)&t.aux, cap * sizeof(uint32_t)));
        HIP_TRYX(c, hipMalloc((void**)&t.text_w, cap * sizeof(uint32_t)));
        t.cap = cap;
        return BSK_OK;
    };
    uint64_t total = 0;
    bool done = false;
    // ---- one-pass path: every range writes into its own slice of a sparse table sized from the
    // record density of the shard head; slices are then gathered (6 % of the data volume).
    // Falls back to the exact count + write passes when a slice overflows.
    const char* ix = c->tune.get("index");
    if (!(ix && strcmp(ix, "twopass") == 0)) {
        const size_t hb = std::min<size_t>(n, 256 * 1024);
        std::vector<uint8_t> head(hb);
        HIP_TRYX(c, hipMemcpyAsync(head.data(), d_buf, hb, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (fastq && !c->norm_active && fastq_head_multiline(head.data(), hb)) return BSK_ERR_MULTILINE_FASTQ;
        uint64_t recs = 0;
        if (fastq) { for (size_t i = 0; i < hb; ++i) recs += head[i] == '\n'; recs /= 4; }
        else { for (size_t i = 0; i + 1 < hb; ++i) recs += (head[i] == '\n' && head[i + 1] == '>'); recs += 1; }
        const double avg = (double)hb / (double)std::max<uint64_t>(recs, 1);
        c->avg_record_bytes = (uint64_t)avg;  // lanes per record of the per-record kernels (a filtered table is no measure)
        const uint64_t sparse_cap = (uint64_t)((double)chunk / std::max(avg * 0.5, 6.0)) + 64;
        const uint64_t need = sparse_cap * nranges;
        if (need * 20 <= (uint64_t)n + (64ull << 20)) {  // never reserve more than the shard itself
            int rc2 = alloc_table(c->sparse, need);
            if (rc2 != BSK_OK) return rc2;
            D.t = c->sparse;
            D.write = 2;
            D.sparse_cap = sparse_cap;
            if (hash) {  // keys in the same slices: k1 ++ k2
                rc2 = grow(c, &c->d_keys_sparse, &c->keys_sparse_cap, 2 * c->sparse.cap, 16);
                if (rc2 != BSK_OK) return rc2;
                HD.k1 = c->d_keys_sparse;
                HD.k2 = c->d_keys_sparse + c->sparse.cap;
            }
            {
                Timed t(c, F ? "k_filter" : (hash ? "k_rmdup_stream" : "k_index"), st);
                HIP_TRYX(c, launch_pass(blocks, anchors, nranges, queue, D));
            }
            HIP_TRYX(c, launch_scan_small(c->d_range_count, c->d_range_base, nranges, st));
            uint64_t status = 0;
            HIP_TRYX(c, hipMemcpyAsync(&total, c->d_range_base + nranges, sizeof total, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            if (status & ERR_FILTER_OVERFLOW) {
                status &= ~(uint64_t)(ERR_FILTER_OVERFLOW | ERR_CAPACITY);
                HIP_TRYX(c, hipMemcpyAsync(c->d_status, &status, sizeof status, hipMemcpyHostToDevice, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                return BSK_ERR_FILTER_FALLBACK;
            }
            if (status & ERR_CAPACITY) {
                status &= ~(uint64_t)ERR_CAPACITY;  // retry exactly
                HIP_TRYX(c, hipMemcpyAsync(c->d_status, &status, sizeof status, hipMemcpyHostToDevice, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                HIP_TRYX(c, launch_reset_queue(queue, st));
                if (D.parts) HIP_TRYX(c, hipMemsetAsync(c->d_parts, 0, (size_t)nranges * sizeof(RangePart), st));
            } else {
                int rc3 = alloc_table(c->table, total + total / 8 + 16);
                if (rc3 != BSK_OK) return rc3;
                c->table.n = total;
                if (total && hash) {
                    rc3 = grow(c, &c->d_keys, &c->keys_cap, total, total / 8 + 16);
                    if (rc3 == BSK_OK) rc3 = grow(c, &c->d_keys2, &c->keys2_cap, total, total / 8 + 16);
                    if (rc3 != BSK_OK) return rc3;
                    Timed t(c, "k_rmdup_compact", st);
                    HIP_TRYX(c, launch_rmdup_compact(c->sparse, sparse_cap, c->d_range_count, c->d_range_base, nranges, c->table, HD,
                                                     HashDev{c->d_keys, c->d_keys2}, st));
                } else if (total)
                    HIP_TRYX(c, launch_index_compact(c->sparse, sparse_cap, c->d_range_count, c->d_range_base, nranges,
                                                     c->table, st));
                done = true;
            }
        }
    }
    if (!done) {
        D.t = c->table;
        D.write = 0;
        HIP_TRYX(c, launch_pass(blocks, anchors, nranges, queue, D));
        HIP_TRYX(c, launch_scan_small(c->d_range_count, c->d_range_base, nranges, st));
        HIP_TRYX(c, hipMemcpyAsync(&total, c->d_range_base + nranges, sizeof total, hipMemcpyDeviceToHost, st));
        if (F) {
            uint64_t status = 0;
            HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            if (status & ERR_FILTER_OVERFLOW) {
                status &= ~(uint64_t)(ERR_FILTER_OVERFLOW | ERR_CAPACITY);
                HIP_TRYX(c, hipMemcpyAsync(c->d_status, &status, sizeof status, hipMemcpyHostToDevice, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                return BSK_ERR_FILTER_FALLBACK;
            }
        }
        HIP_TRYX(c, hipStreamSynchronize(st));
        int rc4 = alloc_table(c->table, total + total / 8 + 16);
        if (rc4 != BSK_OK) return rc4;
        c->table.n = total;
        if (total == 0) return BSK_OK;
        if (hash) {
            rc4 = grow(c, &c->d_keys, &c->keys_cap, total, total / 8 + 16);
            if (rc4 == BSK_OK) rc4 = grow(c, &c->d_keys2, &c->keys2_cap, total, total / 8 + 16);
            if (rc4 != BSK_OK) return rc4;
            HD.k1 = c->d_keys;
            HD.k2 = c->d_keys2;
        }
        D.t = c->table;
        D.write = 1;
        HIP_TRYX(c, launch_reset_queue(queue, st));
        HIP_TRYX(c, launch_pass(blocks, anchors, nranges, queue, D));
    }
    if (total == 0) return BSK_OK;
    if (D.parts) HIP_TRYX(c, launch_index_stitch(c->table, c->d_parts, c->d_range_count, c->d_range_base, nranges, c->d_status, st));
    // start[n] = effective end of the shard (anchors[nranges])
    HIP_TRYX(c, hipMemcpyAsync(c->table.start + total, anchors + nranges, sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    return id_spans(c, d_buf, st);
}

// The record table of a FASTA shard from its '>' bytes alone (stream_fasta_light.hip): starts, header lengths and regions
// are exact; l_seq and text_w hold what they WOULD be if every line of a record but the last were as long as its first.
// Only for a caller that has every byte of the text validated against that layout afterwards (translate: k_translate_wide)
// and falls back to build_index otherwise.  BSK_ERR_FILTER_FALLBACK: not this path's input (a slice overflowed, a first
// line shorter than 16 bases, a custom --id-regexp, text that does not begin with '>').
int build_index_light(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st) {
    c->table.n = 0;
    c->avg_record_bytes = 0;
    if (n == 0 || c->id_custom) return BSK_ERR_FILTER_FALLBACK;
    const size_t hb = std::min<size_t>(n, 256 * 1024);
    std::vector<uint8_t> head(hb);
    HIP_TRYX(c, hipMemcpyAsync(head.data(), d_buf, hb, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (head[0] != '>') return BSK_ERR_FILTER_FALLBACK;
    uint64_t recs = 1;
    for (size_t i = 0; i + 1 < hb; ++i) recs += (head[i] == '\n' && head[i + 1] == '>');
    // the sample must look like this path's text: letters A C G T (any case) on the sequence lines
    {
        bool in_head = true;
        for (size_t i = 0; i < hb; ++i) {
            const uint8_t ch = head[i];
            if (ch == '\n') { in_head = i + 1 < hb && head[i + 1] == '>'; continue; }
            if (in_head) continue;
            const uint8_t u = ch & 0xDFu;
            if (!(u == 'A' || u == 'C' || u == 'G' || u == 'T')) return BSK_ERR_FILTER_FALLBACK;
        }
    }
    const double avg = (double)hb / (double)recs;
    c->avg_record_bytes = (uint64_t)avg;
    const int blocks = std::max(1, c->num_cus * fasta_starts_max_blocks_per_cu());
    const uint64_t waves = (uint64_t)blocks * 4;
    const uint32_t nranges = (uint32_t)pick_nranges(n, waves, c->min_range_bytes, (int)c->tune.num("ranges_per_wave"));
    uint64_t chunk = (n + nranges - 1) / nranges;
    chunk = (chunk + 15) & ~(uint64_t)15;
    if (nranges > c->cap_ranges || !c->d_anchors || !c->d_range_count) {
        if (c->d_anchors) HIP_TRYX(c, hipFree(c->d_anchors));
        if (c->d_range_count) HIP_TRYX(c, hipFree(c->d_range_count));
        if (c->d_range_base) HIP_TRYX(c, hipFree(c->d_range_base));
        c->d_anchors = nullptr; c->d_range_count = nullptr; c->d_range_base = nullptr;
        HIP_TRYX(c, hipMalloc((void**)&c->d_anchors, 2 * ((size_t)nranges + 2) * sizeof(uint64_t)));
        HIP_TRYX(c, hipMalloc((void**)&c->d_range_count, ((size_t)nranges + 1) * sizeof(uint64_t)));
        HIP_TRYX(c, hipMalloc((void**)&c->d_range_base, ((size_t)nranges + 2) * sizeof(uint64_t)));
        c->cap_ranges = nranges;
    }
    uint32_t* queue = reinterpret_cast<uint32_t*>(c->d_anchors + (size_t)nranges + 1);
    const uint64_t sparse_cap = (uint64_t)((double)chunk / std::max(avg * 0.5, 6.0)) + 64;
    const uint64_t need = sparse_cap * nranges;
    if (need * 8 > (uint64_t)n / 4 + (64ull << 20)) return BSK_ERR_FILTER_FALLBACK;  // (tiny records: the full pass is the better one)
    int rc = grow(c, &c->d_keys_sparse, &c->keys_sparse_cap, need, 16);  // (the slices: plain u64 scratch of the context)
    if (rc != BSK_OK) return rc;
    uint64_t n_eff = n;
    {   // effective end of the shard (trailing blank lines dropped), as k_prep computes it -- from the last bytes on the host
        const size_t tb = std::min<size_t>(n, 4096);
        std::vector<uint8_t> tail(tb);
        HIP_TRYX(c, hipMemcpyAsync(tail.data(), d_buf + n - tb, tb, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        size_t e = tb;
        while (e >= 2 && tail[e - 1] == '\n' && tail[e - 2] == '\n') --e;
        if (e < 2 && tb < n) return BSK_ERR_FILTER_FALLBACK;  // (kilobytes of blank lines: leave it to k_prep)
        n_eff = n - (tb - e);
    }
    HIP_TRYX(c, hipMemsetAsync(queue, 0, sizeof(uint32_t), st));
    {
        Timed t(c, "k_fasta_starts", st);
        HIP_TRYX(c, launch_fasta_starts(blocks, d_buf, n_eff, chunk, nranges, queue, c->d_keys_sparse, sparse_cap, c->d_range_count,
                                        c->d_status, st));
    }
    HIP_TRYX(c, launch_scan_small(c->d_range_count, c->d_range_base, nranges, st));
    uint64_t total = 0, status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&total, c->d_range_base + nranges, sizeof total, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    auto give_up = [&](uint64_t bits) -> int {
        status &= ~bits;
        HIP_TRYX(c, hipMemcpyAsync(c->d_status, &status, sizeof status, hipMemcpyHostToDevice, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        return BSK_ERR_FILTER_FALLBACK;
    };
    if (status & ERR_CAPACITY) return give_up(ERR_CAPACITY);
    if (total == 0 || total >= (1ull << 32)) return BSK_ERR_FILTER_FALLBACK;
    RecordTable& t = c->table;
    const uint64_t cap = total + total / 8 + 16;
    if (cap > t.cap || !t.start) {
        for (void* p : {(void*)t.start, (void*)t.l_head, (void*)t.l_seq, (void*)t.aux, (void*)t.text_w})
            if (p) HIP_TRYX(c, hipFree(p));
        t = RecordTable();
        HIP_TRYX(c, hipMalloc((void**)&t.start, (cap + 1) * sizeof(uint64_t)));
        HIP_TRYX(c, hipMalloc((void**)&t.l_head, cap * sizeof(uint32_t)));
        HIP_TRYX(c, hipMalloc((void**)&t.l_seq, cap * sizeof(uint32_t)));
        HIP_TRYX(c, hipMalloc((void**)&t.aux, cap * sizeof(uint32_t)));
        HIP_TRYX(c, hipMalloc((void**)&t.text_w, cap * sizeof(uint32_t)));
        t.cap = cap;
    }
    t.n = total;
    t.id_off = nullptr;
    t.id_len = nullptr;
    {
        Timed tm(c, "k_fasta_heads", st);
        HIP_TRYX(c, launch_fasta_starts_compact(c->d_keys_sparse, sparse_cap, c->d_range_count, c->d_range_base, nranges, n_eff, total, t, st));
        HIP_TRYX(c, launch_fasta_heads(d_buf, n_eff, t, c->d_status, st));
    }
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status & (ERR_LIGHT_UNFIT | ERR_LINE_TOO_LONG)) { t.n = 0; return give_up(ERR_LIGHT_UNFIT | ERR_LINE_TOO_LONG); }
    return BSK_OK;
}

// custom --id-regexp: the ID span of every record of the new table, once per shard (ops_idre.hip)
int id_spans(bsk_ctx* c, const uint8_t* d_buf, hipStream_t st) {
    c->table.id_off = nullptr;
    c->table.id_len = nullptr;
    if (!c->id_custom || c->table.n == 0) return BSK_OK;
    if (!c->d_id_prog) {
        HIP_TRYX(c, hipMalloc((void**)&c->d_id_prog, sizeof(VmProgram)));
        HIP_TRYX(c, hipMemcpy(c->d_id_prog, &c->id_prog, sizeof(VmProgram), hipMemcpyHostToDevice));
    }
    if (c->table.n > c->id_cap || !c->d_id_off) {
        if (c->d_id_off) HIP_TRYX(c, hipFree(c->d_id_off));
        if (c->d_id_len) HIP_TRYX(c, hipFree(c->d_id_len));
        c->d_id_off = c->d_id_len = nullptr;
        const uint64_t cap = c->table.n + c->table.n / 8 + 16;
        HIP_TRYX(c, hipMalloc((void**)&c->d_id_off, cap * 4));
        HIP_TRYX(c, hipMalloc((void**)&c->d_id_len, cap * 4));
        c->id_cap = cap;
    }
    HIP_TRYX(c, launch_id_spans(d_buf, c->table, c->d_id_prog, c->d_id_off, c->d_id_len, st));
    c->table.id_off = c->d_id_off;
    c->table.id_len = c->d_id_len;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// seq
// ---------------------------------------------------------------------------
void set_bits(uint32_t* set, const std::string& letters) {
    for (int k = 0; k < 8; ++k) set[k] = 0;
    for (unsigned char ch : letters) set[ch >> 5] |= 1u << (ch & 31);
}

static const char* alphabet_letters(Alphabet a) {
    switch (a) {
        case AB_DNA: return "acgtACGT -.nN";
        case AB_RNA: return "acguACGU -.nN";
        case AB_DNAredundant: return "acgtryswkmbdhvACGTRYSWKMBDHV -.nN";
        case AB_RNAredundant: return "acguryswkmbdhvACGURYSWKMBDHV -.nN";
        case AB_PROTEIN: return "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ -xX*_.";
        default: return nullptr;
    }
}

void validate_seq_opts(bsk_ctx* c) {  // SeqTransform.Before, seq.go:28-79
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    const std::string& g = o.s("GapLetters");
    if (g.empty()) throw OptError("value of flag -G (--gap-letters) should not be empty");
    for (unsigned char ch : g)
        if (ch > 127) throw OptError("value of -G (--gap-letters) contains non-ASCII characters");
    if (o.i("MinLen") >= 0 && o.i("MaxLen") >= 0 && o.i("MinLen") > o.i("MaxLen"))
        throw OptError("value of flag -m (--min-len) should be >= value of flag -M (--max-len)");
    if (o.f("MinQual") >= 0 && o.f("MaxQual") >= 0 && o.f("MinQual") > o.f("MaxQual"))
        throw OptError("value of flag -Q (--min-qual) should be <= value of flag -R (--max-qual)");
    if (o.b("LowerCase") && o.b("UpperCase"))
        throw OptError("could not give both flags -l (--lower-case) and -u (--upper-case)");
    check_id_regexp(c);
    // the messages of seq.go:52-69 (log.Warn always, the log.Info unless --quiet)
    if ((o.i("MinLen") >= 0 || o.i("MaxLen") >= 0) && !o.b("RemoveGaps")) c->warn("you may switch on flag -g/--remove-gaps to remove spaces");
    if (o.b("Complement") && (c->alphabet == AB_NONE || c->alphabet == AB_PROTEIN))
        c->warn("flag -t (--seq-type) (DNA/RNA) is recommended for computing complement sequences");
    if (!o.b("ValidateSeq") && !(c->alphabet == AB_NONE || c->alphabet == AB_UNLIMIT))
        c->info("when flag -t (--seq-type) given, flag -v (--validate-seq) is automatically switched on", /*unless_quiet=*/true);
}

// `\{[^\}]*$|^[^\{]*\}` (grep.go:38): an opening brace without its closing one, or the reverse -- what is left of "A{2,}"
// when the command line cut it at the comma
static bool has_unquoted_comma(const std::string& p) {
    const size_t open = p.rfind('{');
    if (open != std::string::npos && p.find('}', open) == std::string::npos) return true;
    const size_t close = p.find('}');
    return close != std::string::npos && p.find('{') > close;
}
static const char* const HELP_UNQUOTED_COMMA =
    "possible unquoted comma detected, please use double quotation marks for patterns containing comma, e.g., -p '\"A{2,}\"' "
    "or -p \"\\\"A{2,}\\\"\"";

// sequence bytes of the first record of a shard head (type guess, helper.go:286-291)
static std::vector<uint8_t> head_first_seq(const std::vector<uint8_t>& b, int format, size_t limit) {
    std::vector<uint8_t> s;
    const size_t n = b.size();
    size_t p = 0;
    while (p < n && b[p] != '\n') ++p;
    ++p;
    if (format == BSK_FORMAT_FASTQ) {
        while (p < n && b[p] != '\n' && s.size() < limit) s.push_back(b[p++]);
        return s;
    }
    while (p < n && s.size() < limit) {
        if (b[p] == '>' && b[p - 1] == '\n') break;
        if (b[p] != '\n') s.push_back(b[p]);
        ++p;
    }
    return s;
}

Alphabet partition_alphabet(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, int* rc) {
    *rc = BSK_OK;
    if (c->alphabet != AB_NONE) return c->alphabet;
    const int64_t thr = c->opts.ci("AlphabetGuessSeqLength");
    size_t want = (size_t)std::max<int64_t>(thr, 10000) * 2 + 65536;
    want = std::min(want, n);
    std::vector<uint8_t> h(want);
    if (want) {
        hipError_t e = hipMemcpyAsync(h.data(), d_buf, want, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            c->set_error(std::string("hipMemcpy: ") + hipGetErrorString(e));
            *rc = BSK_ERR_HIP;
            return AB_UNLIMIT;
        }
    }
    std::vector<uint8_t> s = head_first_seq(h, format, (size_t)std::max<int64_t>(thr, 1) );
    if (thr == 0) s = head_first_seq(h, format, h.size());
    return guess_alphabet_less_conservatively(s.data(), s.size(), thr);
}

// (slack so that a slightly larger next result does not reallocate; capped: 1/8 of a 100 GB output is 12 GB of HBM)
int ensure_out(bsk_ctx* c, uint64_t bytes) { return grow(c, &c->d_out, &c->out_cap, bytes, std::min<uint64_t>(bytes / 8, 256ull << 20) + 256); }

int ensure_record_scratch(bsk_ctx* c) {
    const uint64_t n = c->table.n;
    int rc = BSK_OK;
    if (n + 1 > c->out_len_cap || !c->d_out_len) {
        if (c->d_out_len) HIP_TRYX(c, hipFree(c->d_out_len));
        if (c->d_out_off) HIP_TRYX(c, hipFree(c->d_out_off));
        c->d_out_len = nullptr; c->d_out_off = nullptr;
        const uint64_t cap = n + n / 8 + 16;
        HIP_TRYX(c, hipMalloc((void**)&c->d_out_len, cap * sizeof(uint32_t)));
        HIP_TRYX(c, hipMalloc((void**)&c->d_out_off, (cap + 1) * sizeof(uint64_t)));
        c->out_len_cap = cap;
    }
    const uint64_t need = 2 * ((n + 2047) / 2048) + 4;
    rc = grow(c, &c->d_scan_tmp, &c->scan_tmp_cap, need, 16);
    if (rc != BSK_OK) return rc;
    if (!c->d_counter) HIP_TRYX(c, hipMalloc((void**)&c->d_counter, 4 * sizeof(uint64_t)));
    return BSK_OK;
}

// size array -> scan -> total / kept / kernel status; then the caller emits
int finish_sizes(bsk_ctx* c, hipStream_t st, uint64_t* total, uint64_t* kept) {
    HIP_TRYX(c, launch_scan_u32(c->d_out_len, c->d_out_off, c->table.n, c->d_scan_tmp, st));
    HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_count_nonzero(c->d_out_len, c->table.n, c->d_counter, st));
    // records with a very large output are written by whole blocks (k_seq_emit<.., LONG>): list them now, the
    // synchronisation below is needed anyway
    int rc = grow(c, &c->d_long_list, &c->long_list_cap, c->table.n, c->table.n / 8 + 16);
    if (rc != BSK_OK) return rc;
    {
        const char* e = c->tune.get("long_bytes");
        c->long_thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
    }
    HIP_TRYX(c, launch_find_long(c->d_out_len, c->table.n, c->long_thresh, c->d_long_list, c->d_counter + 2, st));
    uint64_t status = 0, lc[2] = {0, 0};
    HIP_TRYX(c, hipMemcpyAsync(total, c->d_out_off + c->table.n, sizeof *total, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(kept, c->d_counter, sizeof *kept, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(lc, c->d_counter + 2, sizeof lc, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    c->long_count = lc[0];
    c->long_max = lc[1];
    return kernel_error_to_status(c, status);
}

// tell the emit kernel which records it must leave to the block-per-chunk launch
int emit_records_at(bsk_ctx* c, const uint8_t* d_buf, size_t n, const SeqParams& Pin, const uint32_t* d_len, const uint64_t* d_off,
                    uint8_t* d_out, uint64_t total, uint64_t kept, hipStream_t st) {
    SeqParams P = Pin;
    P.seg_src = nullptr;
    const RecordTable& t = c->table;
    const char* env = c->tune.get("segcopy");  // off: never; force: whenever the records qualify (tests)
    const bool verbatim = P.fastq && !P.fasta_out && P.print_name && P.print_seq && P.print_qual && !P.qual_only && !P.only_id &&
                          !P.reverse && !P.use_lut && !P.region_on && !P.feat_on && !P.remove_gaps;  // (rename: per record, below)
    bool seg = verbatim && t.n > 0 && total > 0 && ((uintptr_t)d_out & 15u) == 0 && !(env && strcmp(env, "off") == 0);
    if (seg && !(env && strcmp(env, "force") == 0)) seg = kept * 2 >= t.n && total >= (4u << 20);
    if (seg) {
        int rc = grow(c, &c->d_seg_src, &c->seg_src_cap, t.n + 1, t.n / 8 + 16);
        if (rc != BSK_OK) return rc;
        rc = grow(c, &c->d_seg_first, &c->seg_first_cap, seg_tiles(total) + 1, 64);
        if (rc != BSK_OK) return rc;
        uint64_t* d_other = c->d_seg_src + t.n;
        HIP_TRYX(c, hipMemsetAsync(d_other, 0, sizeof(uint64_t), st));
        HIP_TRYX(c, launch_seg_build_fastq(d_buf, n, t, d_len, c->d_seg_src, d_other, st, P.ren_ord));
        HIP_TRYX(c, launch_seg_first(d_off, t.n, c->d_seg_first, st));
        {
            Timed tm(c, "k_seg_copy", st);
            HIP_TRYX(c, launch_seg_copy(c->d_seg_src, d_off, t.n, c->d_seg_first, d_out, total, d_buf, d_buf + n, st));
        }
        uint64_t other = 0;
        HIP_TRYX(c, hipMemcpyAsync(&other, d_other, sizeof other, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (other == 0) return BSK_OK;
        P.seg_src = c->d_seg_src;  // the few records the copy left out
    }
    HIP_TRYX(c, launch_seq_emit(d_buf, t, P, d_len, d_off, d_out, st, total, kept));
    return BSK_OK;
}

int emit_records(bsk_ctx* c, const uint8_t* d_buf, size_t n, const SeqParams& P, uint64_t total, uint64_t kept, hipStream_t st) {
    return emit_records_at(c, d_buf, n, P, c->d_out_len, c->d_out_off, c->d_out, total, kept, st);
}

void apply_long(const bsk_ctx* c, SeqParams* P) {
    P->long_list = c->long_count ? c->d_long_list : nullptr;
    P->long_count = c->long_count;
    P->long_max = c->long_max;
    P->long_thresh = c->long_count ? c->long_thresh : 0u;
}

int empty_result(bsk_ctx* c, bsk_out* out) {
    out->d_data = nullptr;
    out->len = 0;
    out->records = 0;
    uint64_t status = 0;
    HIP_TRYX(c, hipMemcpy(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost));
    return kernel_error_to_status(c, status);
}

// SeqParams that print the whole record unchanged == fastx.Record.Format(lineWidth)
SeqParams format_params(bsk_ctx* c, bool fastq) {
    SeqParams P;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    P.print_name = 1;
    P.print_seq = 1;
    P.print_qual = fastq;
    P.line_width = fastq ? 0 : (int)c->opts.ci("LineWidth");
    P.id_mode = id_mode_of(c);
    return P;
}

static void parse_region_opt(const std::string& region, const char* cmd, int* start, int* end) {
    // reRegion `\-?\d+:\-?\d+` (bigseqkit-lib/helper.go:20) + grep.go:103-118 / subseq.go:83-97
    bool ok = false;
    for (size_t i = 0; i < region.size() && !ok; ++i) {
        size_t p = i;
        if (region[p] == '-') ++p;
        size_t d0 = p;
        while (p < region.size() && isdigit((unsigned char)region[p])) ++p;
        if (p == d0 || p >= region.size() || region[p] != ':') continue;
        ++p;
        if (p < region.size() && region[p] == '-') ++p;
        size_t d1 = p;
        while (p < region.size() && isdigit((unsigned char)region[p])) ++p;
        if (p > d1) ok = true;
    }
    if (!ok) throw OptError("invalid region: " + region + ". type \"seqkit " + cmd + " -h\" for more examples");
    const size_t c = region.find(':');
    const std::string a = region.substr(0, c), b = region.substr(c + 1);
    char* endp = nullptr;
    const long sa = strtol(a.c_str(), &endp, 10);
    if (a.empty() || *endp) throw OptError("strconv.Atoi: parsing \"" + a + "\": invalid syntax");
    const long sb = strtol(b.c_str(), &endp, 10);
    if (b.empty() || *endp) throw OptError("strconv.Atoi: parsing \"" + b + "\": invalid syntax");
    if (sa == 0 || sb == 0) throw OptError("both start and end should not be 0");
    if (sa < 0 && sb > 0) throw OptError("when start < 0, end should not > 0");
    *start = (int)sa;
    *end = (int)sb;
}

// --id-regexp (bigseqkit-lib/helper.go:179-198): the default and the --id-ncbi expression have their own code; any other
// expression is compiled for the position-reporting matcher (regex_vm.hpp) and its spans are computed per shard
void check_id_regexp(bsk_ctx* c) {
    const std::string& re = c->opts.cs("IDRegexp");
    c->id_custom = false;
    if (re.empty() || re == "^(\\S+)\\s?" || re == "\\|([^\\|]+)\\| ") return;
    // reCheckIDregexpStr = `\(.+\)` (helper.go:156)
    const size_t a = re.find('(');
    const size_t b = re.rfind(')');
    if (a == std::string::npos || b == std::string::npos || b < a + 2)
        throw OptError("fastx: regular expression must contain \"(\" and \")\" to capture matched ID. default: ^(\\S+)\\s?");
    try {
        c->id_prog = compile_vm(re);
    } catch (const OptError& e) {
        if (std::string(e.what()).rfind("libbsk:", 0) == 0) throw;  // syntax this matcher does not take: say so
        throw OptError("fastx: fail to compile regexp: " + re);
    }
    if (c->id_prog.ngroups == 0)
        throw OptError("fastx: regular expression must contain \"(\" and \")\" to capture matched ID. default: ^(\\S+)\\s?");
    c->id_custom = true;
}

int id_mode_of(const bsk_ctx* c) {  // 0 default regexp, 1 --id-ncbi, 2 custom (spans in the record table; no description)
    if (c->id_custom) return 2;
    return c->opts.cs("IDRegexp") == "\\|([^\\|]+)\\| " ? 1 : 0;
}

// ---------------------------------------------------------------------------
// pattern helpers shared by grep and locate
// ---------------------------------------------------------------------------
using ByteSet = std::array<uint32_t, 8>;
static inline void set_add(ByteSet& s, uint8_t b) { s[b >> 5] |= 1u << (b & 31); }
static inline bool set_has(const ByteSet& s, uint8_t b) { return (s[b >> 5] >> (b & 31)) & 1u; }

// Seq.Degenerate2Regexp [shenwei356/bio v0.7.0, not in tree; PARITY.md DEG]: the letters a degenerate
// base / residue stands for; nullptr = the byte stays a literal of the regular expression
static std::string degenerate_letters(char c, bool protein) {
    const bool low = c >= 'a' && c <= 'z';
    const char u = low ? (char)(c - 32) : c;
    std::string r;
    if (!protein) {
        switch (u) {
            case 'A': case 'C': case 'G': case 'T': case 'U': r = std::string(1, u); break;
            case 'R': r = "AG"; break; case 'Y': r = "CT"; break; case 'M': r = "AC"; break; case 'K': r = "GT"; break;
            case 'S': r = "CG"; break; case 'W': r = "AT"; break; case 'H': r = "ACT"; break; case 'B': r = "CGT"; break;
            case 'V': r = "ACG"; break; case 'D': r = "AGT"; break; case 'N': r = "ACGT"; break;
            default: return "";
        }
    } else {
        if (u < 'A' || u > 'Z') return "";
        switch (u) {
            case 'B': r = "DN"; break; case 'Z': r = "EQ"; break; case 'J': r = "IL"; break;
            case 'X': r = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"; break;
            default: r = std::string(1, u);
        }
    }
    if (low) for (auto& ch : r) ch = (char)(ch + 32);
    return r;
}

static std::vector<ByteSet> class_sets(const std::string& p, bool degenerate, bool protein, bool icase) {
    std::vector<ByteSet> out;
    for (char ch : p) {
        ByteSet s{};
        std::string letters = degenerate ? degenerate_letters(ch, protein) : std::string();
        if (letters.empty()) {
            if (degenerate && !((ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z')))
                throw OptError("libbsk: with -d the HIP path takes patterns made of letters only (regular-expression "
                               "syntax is not supported): " + p);
            letters = std::string(1, ch);
        }
        for (char l : letters) {
            set_add(s, (uint8_t)l);
            if (icase && l >= 'A' && l <= 'Z') set_add(s, (uint8_t)(l + 32));
            if (icase && l >= 'a' && l <= 'z') set_add(s, (uint8_t)(l - 32));
        }
        out.push_back(s);
    }
    return out;
}

static void complement_table(Alphabet ab, uint8_t m[256]) {
    for (int i = 0; i < 256; ++i) m[i] = (uint8_t)i;
    const char *from = nullptr, *to = nullptr;
    if (ab == AB_DNA || ab == AB_DNAredundant) { from = "acgtryswkmbdhvACGTRYSWKMBDHV"; to = "tgcayrswmkvhdbTGCAYRSWMKVHDB"; }
    else if (ab == AB_RNA || ab == AB_RNAredundant) { from = "acguryswkmbdhvACGURYSWKMBDHV"; to = "ugcayrswmkvhdbUGCAYRSWMKVHDB"; }
    if (from) for (size_t k = 0; from[k]; ++k) m[(uint8_t)from[k]] = (uint8_t)to[k];
}

// class pattern that matches on the forward text exactly where the original matches on RevCom(text)
static std::vector<ByteSet> revcom_sets(const std::vector<ByteSet>& s, Alphabet ab) {
    uint8_t comp[256];
    complement_table(ab, comp);
    std::vector<ByteSet> out(s.size());
    for (size_t q = 0; q < s.size(); ++q)
        for (int b = 0; b < 256; ++b)
            if (set_has(s[s.size() - 1 - q], comp[b])) set_add(out[q], (uint8_t)b);
    return out;
}

static std::string read_whole_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw OptError("open " + path + ": no such file or directory");
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

// breader.NewDefaultBufferedReader: one pattern per line, line ends trimmed (grep.go:126-140)
std::vector<std::string> read_pattern_lines(const std::string& path) {
    std::vector<std::string> out;
    const std::string s = read_whole_file(path);
    for (size_t i = 0; i < s.size();) {
        size_t j = s.find('\n', i);
        if (j == std::string::npos) j = s.size();
        size_t e = j;
        while (e > i && (s[e - 1] == '\r' || s[e - 1] == '\n')) --e;
        out.emplace_back(s, i, e - i);
        i = j + 1;
    }
    return out;
}

// fastx.GetSeqsMap(file, seq.Unlimit, ...) (locate.go:86): full name -> sequence, file order (PARITY.md Q11);
// a repeated name keeps the later sequence, like the Go map assignment
static std::vector<std::pair<std::string, std::string>> read_pattern_fasta(const std::string& path) {
    std::vector<std::pair<std::string, std::string>> out;
    const std::string s = read_whole_file(path);
    bool have = false;
    for (size_t i = 0; i < s.size();) {
        size_t j = s.find('\n', i);
        if (j == std::string::npos) j = s.size();
        size_t e = j;
        while (e > i && s[e - 1] == '\r') --e;
        if (e > i && s[i] == '>') {
            const std::string name(s, i + 1, e - i - 1);
            have = true;
            size_t k = 0;
            for (; k < out.size(); ++k) if (out[k].first == name) break;
            if (k < out.size()) out.erase(out.begin() + (long)k);
            out.emplace_back(name, "");
        } else if (have) {
            out.back().second.append(s, i, e - i);
        }
        i = j + 1;
    }
    return out;
}

// ---------------------------------------------------------------------------
// grep  (Grep.Before, bigseqkit-lib/grep.go:41-253)
// ---------------------------------------------------------------------------
void validate_grep_opts(bsk_ctx* c) {
    Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    bool any = !o.s("PatternFile").empty();
    for (auto& p : o.sl("Pattern")) if (!p.empty()) any = true;
    // PARITY.md Q17: the default Pattern [""] must not defeat this guard (grep.go:53)
    if (!any) throw OptError("one of flags -p (--pattern) and -f (--pattern-file) needed");
    // the log lines of Grep.Before (grep.go:57-98), in its order
    for (auto& p : o.sl("Pattern"))
        if (has_unquoted_comma(p)) { c->warn(HELP_UNQUOTED_COMMA); break; }
    if (o.b("Degenerate") && !o.b("BySeq")) c->info("when flag -d (--degenerate) given, flag -s (--by-seq) is automatically on");
    if (o.b("Degenerate")) o.mut("BySeq").b = true;
    if (o.i("MaxMismatch") > 0) {
        if (o.b("UseRegexp") || o.b("Degenerate"))
            throw OptError("flag -r (--use-regexp) or -d (--degenerate) not allowed when giving flag -m (--max-mismatch)");
        if (!o.b("BySeq")) c->info("when value of flag -m (--max-mismatch) > 0, flag -s (--by-seq) is automatically on");
        o.mut("BySeq").b = true;
        if (o.i("MaxMismatch") > 4) c->warn("large value flag -m/--max-mismatch will slow down the search");
    }
    if (o.b("UseRegexp") && o.b("Degenerate"))
        throw OptError("could not give both flags -d (--degenerate) and -r (--use-regexp)");
    c->region_on = false;
    if (!o.s("Region").empty()) {
        c->region_on = true;
        if (!o.b("BySeq")) c->info("when flag -R (--region) given, flag -s (--by-seq) is automatically on");
        o.mut("BySeq").b = true;
        parse_region_opt(o.s("Region"), "grep", &c->region_start, &c->region_end);
    }
    c->patterns.clear();
    c->regexes.clear();
    c->pattern_cls.clear();
    c->max_mm = (int)o.i("MaxMismatch");
    c->general = o.b("Degenerate") || c->max_mm > 0;
    c->patterns_uploaded = false;
    // grep.go:122-252: the pattern file replaces -p when given
    const std::vector<std::string> given = !o.s("PatternFile").empty() ? read_pattern_lines(o.s("PatternFile")) : o.sl("Pattern");
    std::unordered_set<std::string> seen;
    const bool default_id_re = o.cs("IDRegexp") == "^(\\S+)\\s?" && !o.cb("IDNCBI");
    for (std::string p : given) {
        if (p.empty()) continue;
        // grep.go:140-147, 199-207 (unless --quiet)
        if (p[0] == '>') c->warn("symbol \">\" detected, it should not be a part of the sequence ID/name: " + p, true);
        else if (p[0] == '@') c->warn("symbol \"@\" detected, it should not be a part of the sequence ID/name. " + p, true);
        else if (!o.b("ByName") && default_id_re && p.find_first_of("\t ") != std::string::npos)
            c->warn("space found in pattern, you may need use -n/--by-name: " + p, true);
        if (o.b("UseRegexp")) {  // grep.go:148-153, 211-225: "(?i)" + p with -i, then regexp.Compile
            if (o.b("IgnoreCase")) p = "(?i)" + p;
            if (!seen.insert(p).second) continue;
            c->regexes.push_back(compile_regex(p));
            c->patterns.push_back(p);
            continue;
        }
        if (o.b("Degenerate")) {
            // Degenerate2Regexp with the alphabet of -t (nil for auto => nucleotide map), "(?i)" with -i
            if (!seen.insert(p).second) continue;
            c->pattern_cls.push_back(class_sets(p, true, c->alphabet == AB_PROTEIN, o.b("IgnoreCase")));
            c->patterns.push_back(p);
            continue;
        }
        if (o.b("BySeq")) {
            if (c->max_mm > 0 && c->max_mm > (int)p.size()) throw OptError("mismatch should be <= length of sequence: " + p);
            const uint8_t* b = (const uint8_t*)p.data();
            if (!(alphabet_valid_letters(AB_DNAredundant, b, p.size()) || alphabet_valid_letters(AB_RNAredundant, b, p.size()) ||
                  alphabet_valid_letters(AB_PROTEIN, b, p.size())))
                throw OptError("illegal DNA/RNA/Protein sequence: " + p);
        }
        if (o.b("IgnoreCase"))
            for (auto& ch : p) if (ch >= 'A' && ch <= 'Z') ch += 32;
        if (!seen.insert(p).second) continue;
        if (c->general) c->pattern_cls.push_back(class_sets(p, false, false, o.b("IgnoreCase")));
        c->patterns.push_back(p);
    }
    if (!o.s("PatternFile").empty()) {  // grep.go:191-197 (unless --quiet; a warning when the file held none)
        const size_t np = o.b("UseRegexp") ? c->regexes.size() : c->patterns.size();
        const std::string m = std::to_string(np) + " patterns loaded from file";
        if (np == 0) c->warn(m, true); else c->info(m, true);
    }
    if (o.b("DeleteMatched") && !o.b("InvertMatch")) {  // PARITY.md DEL
        // with -m the reference takes grepBySeqMismatches (grep.go:255-365), which never deletes a pattern, and the driver
        // returns its records as they are (bigseqkit/grep.go:141-143): --delete-matched is a no-op there
        if (o.b("BySeq") && c->max_mm > 0) o.mut("DeleteMatched").b = false;
        const size_t np = o.b("UseRegexp") ? c->regexes.size() : c->patterns.size();
        if ((o.b("BySeq") || o.b("UseRegexp")) && np > 255)  // (15 per hit-bit array, 17 arrays; round 2 stopped at 15)
            throw OptError("libbsk: --delete-matched with more than 255 sequence / regexp patterns is not provided");
    }
}

static std::string revcom_pattern(const std::string& p, Alphabet ab) {
    uint8_t m[256];
    complement_table(ab, m);
    std::string r(p.rbegin(), p.rend());
    for (auto& ch : r) ch = (char)m[(uint8_t)ch];
    return r;
}

// class sets of all patterns (forward, then reverse-complemented when `rc`), 8 dwords per position
static int upload_classes(bsk_ctx* c, bool rc, Alphabet ab, hipStream_t st) {
    std::vector<uint32_t> flat;
    for (int pass = 0; pass < (rc ? 2 : 1); ++pass)
        for (auto& sets : c->pattern_cls) {
            const std::vector<ByteSet> use = pass ? revcom_sets(sets, ab) : sets;
            for (auto& s : use) flat.insert(flat.end(), s.begin(), s.end());
        }
    int r = grow(c, &c->d_cls, &c->cls_cap, flat.size() + 8);
    if (r != BSK_OK) return r;
    if (!flat.empty()) HIP_TRYX(c, hipMemcpyAsync(c->d_cls, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    return BSK_OK;
}

// open-addressing set of the ID / name patterns keyed by fnv1a64 (pattern_match_dev.hpp)
static int upload_pattern_set(bsk_ctx* c, hipStream_t st) {
    uint64_t slots = 16;
    while (slots < 2 * c->patterns.size()) slots <<= 1;
    std::vector<uint64_t> keys(slots, 0);
    std::vector<uint32_t> idx(slots, 0);
    for (size_t k = 0; k < c->patterns.size(); ++k) {
        const std::string& p = c->patterns[k];
        uint64_t h = 1469598103934665603ull;
        for (unsigned char ch : p) h = (h ^ ch) * 1099511628211ull;  // patterns are already lower-cased with -i
        if (!h) h = 1;
        uint64_t s = h & (slots - 1);
        while (keys[s]) s = (s + 1) & (slots - 1);
        keys[s] = h;
        idx[s] = (uint32_t)k;
    }
    int r = grow(c, &c->d_set_keys, &c->set_keys_cap, slots);
    if (r != BSK_OK) return r;
    r = grow(c, &c->d_set_idx, &c->set_idx_cap, slots);
    if (r != BSK_OK) return r;
    HIP_TRYX(c, hipMemcpyAsync(c->d_set_keys, keys.data(), slots * 8, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipMemcpyAsync(c->d_set_idx, idx.data(), slots * 4, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    c->set_slots = slots;
    return BSK_OK;
}

static int upload_patterns(bsk_ctx* c, const std::vector<std::string>& all, hipStream_t st) {
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> off{0};
    for (auto& p : all) {
        bytes.insert(bytes.end(), p.begin(), p.end());
        off.push_back((uint32_t)bytes.size());
    }
    int rc = grow(c, &c->d_pat, &c->pat_cap, bytes.size() + 16);
    if (rc != BSK_OK) return rc;
    rc = grow(c, &c->d_pat_off, &c->pat_off_cap, off.size());
    if (rc != BSK_OK) return rc;
    if (!bytes.empty()) HIP_TRYX(c, hipMemcpyAsync(c->d_pat, bytes.data(), bytes.size(), hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipMemcpyAsync(c->d_pat_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// the fused pattern filter (stream_filter.hip): host side
// ---------------------------------------------------------------------------
// `all` = the pattern strings as uploaded to c->d_pat (forward, then reverse-complemented); the first `nuse` of them
// are searched.  Builds the collision-free pair-hash table and uploads it.  false: not applicable (pattern lengths,
// too many patterns, no collision-free table found, BSK_FILTER=off) -- the caller keeps the record-table path.
static bool make_filter(bsk_ctx* c, const std::vector<std::string>& all, size_t nuse, bool invert, bool icase, hipStream_t st,
                        FilterDev* F, int* rc) {
    *rc = BSK_OK;
    const char* env = c->tune.get("filter");
    if (env && strcmp(env, "off") == 0) return false;
    if (nuse == 0 || nuse * 4 > FILTER_MAX_ENTRIES) return false;
    for (size_t k = 0; k < nuse; ++k)
        if (all[k].size() < FILTER_MIN_LEN || all[k].size() > FILTER_MAX_LEN) return false;
    std::vector<uint32_t> tab(512, 0u);  // T1 ++ T2
    std::vector<uint16_t> ent(FILTER_MAX_ENTRIES, 0);
    std::vector<uint8_t> padded(FILTER_MAX_PATTERNS * FILTER_MAX_LEN, 0);
    uint32_t e = 0;
    for (size_t k = 0; k < nuse; ++k) {
        memcpy(padded.data() + k * FILTER_MAX_LEN, all[k].data(), all[k].size());
        for (uint32_t j = 0; j < 4; ++j, ++e) {
            uint32_t first, second;
            memcpy(&first, all[k].data() + j, 4);  // little-endian dwords, as the kernel loads the text
            memcpy(&second, all[k].data() + j + 4, 4);
            tab[filter_code(first)] |= 1u << e;
            tab[256 + filter_code(second)] |= 1u << e;
            ent[e] = (uint16_t)(k | (j << 5) | (all[k].size() << 8));  // (FILTER_MAX_LEN = 64 fits the high byte)
        }
    }
    const size_t o_ent = 512 * 4, o_pat = o_ent + FILTER_MAX_ENTRIES * 2;
    int r = grow(c, &c->d_ftab, &c->ftab_cap, o_pat + padded.size() + 64);
    if (r != BSK_OK) { *rc = r; return false; }
    hipError_t he = hipMemcpyAsync(c->d_ftab, tab.data(), 512 * 4, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(c->d_ftab + o_ent, ent.data(), FILTER_MAX_ENTRIES * 2, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(c->d_ftab + o_pat, padded.data(), padded.size(), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);  // the vectors live in this frame
    if (he != hipSuccess) { c->set_error(std::string("hipMemcpy: ") + hipGetErrorString(he)); *rc = BSK_ERR_HIP; return false; }
    F->t1 = reinterpret_cast<const uint32_t*>(c->d_ftab);
    F->ent = reinterpret_cast<const uint16_t*>(c->d_ftab + o_ent);
    F->pat_padded = reinterpret_cast<const uint32_t*>(c->d_ftab + o_pat);
    F->ignore_case = icase ? 1 : 0;
    F->invert = invert ? 1 : 0;
    return true;
}

int grep_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    c->last_count = 0;
    int rc = BSK_OK;
    // exact sequence patterns on FASTQ: the streaming pass itself selects the records (stream_filter.hip) and the
    // per-record kernels below run on the selected ones only.  Everything else (and any shard on which the filter gives
    // up) goes through the table of all records.
    bool filtered = false;
    if (fastq && n > 0 && o.b("BySeq") && !c->general && c->regexes.empty() && !c->region_on && !o.b("Circular") &&
        !o.b("DeleteMatched") && !c->patterns.empty()) {
        Alphabet fab = partition_alphabet(c, d_buf, n, format, st, &rc);
        if (rc != BSK_OK) return rc;
        if (fab == AB_NONE) fab = AB_UNLIMIT;
        const bool both = !(o.b("OnlyPositiveStrand") || fab == AB_UNLIMIT || fab == AB_PROTEIN);
        std::vector<std::string> all = c->patterns;
        if (both)
            for (auto& p : c->patterns) all.push_back(revcom_pattern(p, fab));
        rc = upload_patterns(c, all, st);
        if (rc != BSK_OK) return rc;
        FilterDev F;
        if (make_filter(c, all, all.size(), o.b("InvertMatch"), o.b("IgnoreCase"), st, &F, &rc)) {
            rc = build_index_filtered(c, d_buf, n, format, st, &F);
            if (rc == BSK_OK) filtered = true;
            else if (rc != BSK_ERR_FILTER_FALLBACK) return rc;
        } else if (rc != BSK_OK) return rc;
    }
    if (!filtered) rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    uint64_t total = 0, kept = 0;
    TextTableH tt{nullptr, nullptr, nullptr};
    if (c->table.n > 0) {
        Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);
        if (rc != BSK_OK) return rc;
        if (ab == AB_NONE) ab = AB_UNLIMIT;
        GrepParams G;
        memset(&G, 0, sizeof G);
        G.fastq = fastq;
        G.by_seq = o.b("BySeq");
        G.by_name = o.b("ByName");
        G.invert = o.b("InvertMatch");
        G.ignore_case = o.b("IgnoreCase");
        G.circular = o.b("Circular") && !c->region_on;  // the region branch wins (grep.go:447-456)
        G.region_on = c->region_on;
        G.region_start = c->region_start;
        G.region_end = c->region_end;
        // grep.go:404-409: protein / unlimit sequences are searched on the '+' strand only
        const bool only_pos = o.b("OnlyPositiveStrand") || ab == AB_UNLIMIT || ab == AB_PROTEIN;
        G.both_strands = G.by_seq && !only_pos;
        G.id_mode = id_mode_of(c);
        G.line_width = fastq ? 0 : (int)o.ci("LineWidth");
        G.npat = (int)c->patterns.size();
        // (uses d_out_len as scratch: before the match kernel.)  A search in the sequences reads them many times at
        // arbitrary offsets: wrapped FASTA records get a linear copy first (the emit below goes back to the views)
        const bool flat_text = !fastq && G.by_seq;
        rc = prepare_text(c, d_buf, format, st, &tt, flat_text, false, n);
        if (rc != BSK_OK) return rc;
        if (!c->regexes.empty()) {
            if (!c->patterns_uploaded) {
                rc = grow(c, &c->d_regex, &c->regex_cap, c->regexes.size());
                if (rc != BSK_OK) return rc;
                HIP_TRYX(c, hipMemcpyAsync(c->d_regex, c->regexes.data(), c->regexes.size() * sizeof(RegexProgram),
                                           hipMemcpyHostToDevice, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                c->patterns_uploaded = true;
            }
            uint8_t comp[256];
            complement_table(ab, comp);
            if (!c->d_lut) HIP_TRYX(c, hipMalloc((void**)&c->d_lut, 256));
            HIP_TRYX(c, hipMemcpyAsync(c->d_lut, comp, 256, hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));  // comp lives on the host stack
            G.regex = c->d_regex;
            G.comp = c->d_lut;
        } else if (!G.by_seq) {
            // ID / name: the patterns do not depend on the shard, upload once per context
            if (!c->patterns_uploaded) {
                rc = upload_patterns(c, c->patterns, st);
                if (rc != BSK_OK) return rc;
                if (c->patterns.size() > 8) {
                    rc = upload_pattern_set(c, st);
                    if (rc != BSK_OK) return rc;
                } else c->set_slots = 0;
                c->patterns_uploaded = true;
            }
            if (c->set_slots) { G.set_keys = c->d_set_keys; G.set_idx = c->d_set_idx; G.set_mask = c->set_slots - 1; }
        } else {
            std::vector<std::string> all = c->patterns;
            if (G.both_strands)
                for (auto& p : c->patterns) all.push_back(revcom_pattern(p, ab));
            rc = upload_patterns(c, all, st);
            if (rc != BSK_OK) return rc;
            if (c->general) {
                rc = upload_classes(c, G.both_strands, ab, st);
                if (rc != BSK_OK) return rc;
                G.general = 1;
                G.max_mm = c->max_mm;
                G.cls = c->d_cls;
                size_t longest = 0;
                for (auto& p : all) longest = std::max(longest, p.size());
                G.sa_ok = longest <= 64 && all.size() <= 8 && c->max_mm <= 3 && !G.circular && !c->tune.is("grep_shiftand", "off");
            }
        }
        G.pat = c->d_pat;
        G.pat_off = c->d_pat_off;
        rc = ensure_record_scratch(c);
        if (rc != BSK_OK) return rc;
        if (filtered) {
            // the table holds exactly the records the command prints (the streaming pass verified every occurrence,
            // stream_filter.hip): no second search, only their formatted sizes
            SeqParams FP = format_params(c, fastq);
            FP.buf_end = d_buf + n;
            HIP_TRYX(c, launch_seq_size(d_buf, c->table, FP, c->d_out_len, c->d_status, st));
        } else if (G.by_seq && G.general && G.sa_ok) {
            // one lane per record (k_grep_shiftand): not for chromosomes
            const char* e = c->tune.get("long_bytes");
            const uint32_t thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
            rc = grow(c, &c->d_long_list, &c->long_list_cap, c->table.n, c->table.n / 8 + 16);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
            HIP_TRYX(c, launch_find_long(c->table.l_seq, c->table.n, thresh, c->d_long_list, c->d_counter + 2, st));
            uint64_t lc[2] = {0, 0};
            HIP_TRYX(c, hipMemcpyAsync(lc, c->d_counter + 2, sizeof lc, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            if (lc[0]) G.sa_ok = 0;
        } else if (G.by_seq && !G.general && !G.regex) {
            // chromosome-sized sequences are searched by whole blocks (k_grep_seq<.., LONG>): list them
            const char* e = c->tune.get("long_bytes");
            const uint32_t thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
            rc = grow(c, &c->d_long_list, &c->long_list_cap, c->table.n, c->table.n / 8 + 16);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
            HIP_TRYX(c, launch_find_long(c->table.l_seq, c->table.n, thresh, c->d_long_list, c->d_counter + 2, st));
            uint64_t lc[2] = {0, 0};
            HIP_TRYX(c, hipMemcpyAsync(lc, c->d_counter + 2, sizeof lc, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            if (lc[0]) {
                rc = grow(c, &c->d_hit_list, &c->hit_list_cap, lc[0], 64);
                if (rc != BSK_OK) return rc;
                HIP_TRYX(c, hipMemsetAsync(c->d_hit_list, 0, lc[0] * sizeof(uint32_t), st));
                G.long_list = c->d_long_list;
                G.long_hit = c->d_hit_list;
                G.long_count = lc[0];
                G.long_max = lc[1];
                G.long_thresh = thresh;
            }
        }
        if (!filtered) HIP_TRYX(c, launch_grep_match(d_buf, n, c->table, &tt, G, c->d_out_len, st, c->avg_record_bytes));
        if (o.b("DeleteMatched") && !G.invert) {
            // grep.go:463-511 + bigseqkit/grep.go:144-156: a pattern is dropped at its first hit and the driver keeps
            // the lowest partition per pattern, so every pattern selects its FIRST record in file order (PARITY.md DEL)
            const uint64_t N = c->table.n;
            const bool exact_key = !G.by_seq && !o.b("UseRegexp");
            if (exact_key) {
                // all records with the ID / name of a hit are hits: "first per pattern" = hit AND first of its key group
                RmDupParams R;
                memset(&R, 0, sizeof R);
                R.fastq = fastq;
                R.by_name = G.by_name;
                R.ignore_case = G.ignore_case;
                R.id_mode = G.id_mode;
                R.line_width = G.line_width;
                R.buf_end = d_buf + n;
                uint64_t cap = 0;
                uint64_t* tk = nullptr;
                rc = key_table(c, N, &cap, &tk, st);
                if (rc != BSK_OK) return rc;
                Arena A;
                const uint64_t o_first = A.take(N * 4);
                rc = arena_reserve(c, &A);
                if (rc != BSK_OK) return rc;
                uint32_t* d_firsts = A.at<uint32_t>(o_first);
                HIP_TRYX(c, launch_rmdup_hash(d_buf, n, c->table, tt, R, c->d_keys, nullptr, st));
                HIP_TRYX(c, launch_rmdup_insert(c->d_keys, N, 0, tk, cap, st));
                HIP_TRYX(c, launch_rmdup_resolve(d_buf, c->table, tt, R, c->d_keys, tk, cap, d_firsts, c->d_status, st));
                HIP_TRYX(c, launch_mask_u32(c->d_out_len, d_firsts, N, st));
            } else if (G.npat == 1) {
                // one pattern: only its first hit survives
                HIP_TRYX(c, hipMemsetAsync(c->d_counter + 3, 0xFF, 8, st));
                HIP_TRYX(c, launch_first_nonzero(c->d_out_len, N, c->d_counter + 3, st));
                HIP_TRYX(c, launch_keep_only(c->d_out_len, N, c->d_counter + 3, st));
            } else {
                // several sequence / regexp patterns (grep.go:463-511): the records are visited in file order, a record
                // is a hit when one of the REMAINING patterns matches it, and that pattern -- the first one in the order
                // the patterns were given (PARITY.md Q11; the reference walks a Go map) -- is dropped.  At most one
                // record per pattern is selected, so the walk is: hit bits of every pattern (one match launch each),
                // then <= npat rounds of "first record after the last selected one that still matches something".
                const int np = G.npat;
                // hit bits in arrays of 15 patterns each (bit k = pattern on '+', bit 16 + k = on '-', bit 31 of array 0 =
                // selected); round 2 had one array and refused more than 15 patterns
                const int nblk = (np + 14) / 15;
                Arena A;
                const uint64_t o_masks = A.take((uint64_t)nblk * N * 4), o_hit = A.take(N * 4);
                rc = arena_reserve(c, &A);
                if (rc != BSK_OK) return rc;
                uint32_t* d_masks = A.at<uint32_t>(o_masks);
                uint32_t* d_hit = A.at<uint32_t>(o_hit);
                HIP_TRYX(c, hipMemsetAsync(d_masks, 0, (uint64_t)nblk * N * 4, st));
                const std::vector<std::string> all_patterns = c->patterns;
                const auto all_cls = c->pattern_cls;
                // the reference asks the '+' strand about every remaining pattern before it turns to the '-' strand
                // (grep.go:420-433)
                const int nstrands = G.both_strands ? 2 : 1;
                for (int k = 0; k < np; ++k) {
                    GrepParams G1 = G;
                    G1.npat = 1;
                    if (G.regex) {
                        G1.regex = c->d_regex + k;
                    } else {
                        c->patterns.assign(1, all_patterns[k]);
                        if (c->general) c->pattern_cls.assign(1, all_cls[k]);
                        std::vector<std::string> one = c->patterns;
                        if (G.both_strands) one.push_back(revcom_pattern(all_patterns[k], ab));
                        rc = upload_patterns(c, one, st);
                        if (rc == BSK_OK && c->general) rc = upload_classes(c, G.both_strands, ab, st);
                        c->patterns = all_patterns;
                        c->pattern_cls = all_cls;
                        if (rc != BSK_OK) return rc;
                    }
                    for (int sd = 0; sd < nstrands; ++sd) {
                        G1.strand_only = sd + 1;
                        if (G1.long_hit) HIP_TRYX(c, hipMemsetAsync(c->d_hit_list, 0, G.long_count * sizeof(uint32_t), st));
                        HIP_TRYX(c, launch_grep_match(d_buf, n, c->table, &tt, G1, d_hit, st, c->avg_record_bytes));
                        HIP_TRYX(c, launch_or_bit(d_masks + (uint64_t)(k / 15) * N, d_hit, N, 1u << (16 * sd + k % 15), st));
                    }
                }
                std::vector<uint32_t> remaining(nblk);
                for (int b = 0; b < nblk; ++b) remaining[b] = (1u << std::min(15, np - 15 * b)) - 1u;
                uint64_t from = 0;
                auto any_left = [&] { for (uint32_t r : remaining) if (r) return true; return false; };
                while (any_left() && from < N) {
                    // the first record at or after `from` that one of the remaining patterns matches: per array, then the lowest
                    uint64_t idx = ~0ull;
                    std::vector<uint64_t> first(nblk, ~0ull);
                    HIP_TRYX(c, hipMemsetAsync(c->d_counter + 3, 0xFF, 8, st));
                    if (nblk == 1) {
                        HIP_TRYX(c, launch_first_masked(d_masks, N, remaining[0] | (remaining[0] << 16), from, c->d_counter + 3, st));
                        HIP_TRYX(c, hipMemcpyAsync(&idx, c->d_counter + 3, 8, hipMemcpyDeviceToHost, st));
                        HIP_TRYX(c, hipStreamSynchronize(st));
                    } else {
                        for (int b = 0; b < nblk; ++b) {
                            if (!remaining[b]) continue;
                            HIP_TRYX(c, hipMemsetAsync(c->d_counter + 3, 0xFF, 8, st));
                            HIP_TRYX(c, launch_first_masked(d_masks + (uint64_t)b * N, N, remaining[b] | (remaining[b] << 16), from, c->d_counter + 3, st));
                            HIP_TRYX(c, hipMemcpyAsync(&first[b], c->d_counter + 3, 8, hipMemcpyDeviceToHost, st));
                            HIP_TRYX(c, hipStreamSynchronize(st));
                            idx = std::min(idx, first[b]);
                        }
                    }
                    if (idx == ~0ull) break;
                    // the first remaining pattern (in the order given) that matched, '+' strand before '-'
                    std::vector<uint32_t> m(nblk);
                    for (int b = 0; b < nblk; ++b) HIP_TRYX(c, hipMemcpy(&m[b], d_masks + (uint64_t)b * N + idx, 4, hipMemcpyDeviceToHost));
                    int drop = -1;
                    for (int pass = 0; pass < 2 && drop < 0; ++pass)
                        for (int b = 0; b < nblk && drop < 0; ++b) {
                            const uint32_t hit = (pass ? (m[b] >> 16) : m[b]) & remaining[b];
                            if (hit) drop = 15 * b + (__builtin_ffs((int)hit) - 1);
                        }
                    if (drop < 0) break;  // (cannot happen: idx matched something)
                    remaining[drop / 15] &= ~(1u << (drop % 15));
                    m[0] |= 0x80000000u;  // bit 31 of array 0: selected
                    HIP_TRYX(c, hipMemcpy(d_masks + idx, &m[0], 4, hipMemcpyHostToDevice));
                    from = idx + 1;
                }
                HIP_TRYX(c, launch_keep_selected(c->d_out_len, d_masks, N, st));
            }
        }
        rc = finish_sizes(c, st, &total, &kept);
        if (rc == BSK_OK && o.b("DeleteMatched") && !G.invert) {
            uint64_t status = 0;
            HIP_TRYX(c, hipMemcpy(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost));
            if (status & ERR_HASH_COLLISION) {
                c->set_error("libbsk: two distinct subjects share one 64-bit XXH64 key; refusing to guess (rerun on the CPU path)");
                return BSK_ERR_UNSUPPORTED;
            }
        }
        if (rc != BSK_OK) return rc;
    } else {
        rc = empty_result(c, out);
        if (rc != BSK_OK) return rc;
    }
    c->last_count = kept;
    if (o.b("Count")) {  // grep.go:526-540: one element holding the decimal count
        const std::string txt = std::to_string(kept) + "\n";
        rc = ensure_out(c, txt.size());
        if (rc != BSK_OK) return rc;
        HIP_TRYX(c, hipMemcpy(c->d_out, txt.data(), txt.size(), hipMemcpyHostToDevice));
        out->d_data = c->d_out;
        out->len = txt.size();
        out->records = 1;
        return BSK_OK;
    }
    out->d_data = nullptr;
    out->len = 0;
    out->records = 0;
    if (total == 0) return BSK_OK;
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    SeqParams P = format_params(c, fastq);
    if (!fastq && tt.text_w == c->d_text_w) {  // the search ran on linear copies: the emit reads the wrapped text in place
        rc = prepare_text(c, d_buf, format, st, &tt, false, /*keep_out_len=*/true);
        if (rc != BSK_OK) return rc;
    }
    P.text_w = tt.text_w; P.lin_off = tt.lin_off; P.lin = tt.lin;
    apply_long(c, &P);
    { const int rce = emit_records(c, d_buf, n, P, total, kept, st); if (rce != BSK_OK) return rce; }
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// locate  (Locate.Before, bigseqkit-lib/locate.go:33-193; exact patterns)
// ---------------------------------------------------------------------------
void validate_locate_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    bool any = !o.s("PatternFile").empty();
    for (auto& p : o.sl("Pattern")) if (!p.empty()) any = true;
    if (!any) throw OptError("one of flags -p (--pattern) and -f (--pattern-file) needed");  // PARITY.md Q17
    for (auto& p : o.sl("Pattern"))  // locate.go:50-59
        if (has_unquoted_comma(p)) { c->warn(HELP_UNQUOTED_COMMA); break; }
    if (o.i("MaxMismatch") > 0) {
        if (o.b("Degenerate")) throw OptError("flag -d (--degenerate) not allowed when giving flag -m (--max-mismatch)");
        if (o.b("UseRegexp")) throw OptError("flag -r (--use-regexp) not allowed when giving flag -m (--use-regexp)");
        if (o.b("NonGreedy")) c->info("flag -G (--non-greedy) ignored when giving flag -m (--max-mismatch)", true);  // :68-70
    }
    if (o.b("UseFmi")) {
        if (o.b("Degenerate")) throw OptError("flag -d (--degenerate) ignored when giving flag -F (--use-fmi)");
        if (o.b("UseRegexp")) throw OptError("flag -r (--use-regexp) ignored when giving flag -F (--use-fmi)");
    }
    c->patterns.clear();
    c->pattern_names.clear();
    c->pattern_disp.clear();
    c->pattern_cls.clear();
    c->max_mm = (int)o.i("MaxMismatch");
    c->fmi_order = c->max_mm > 0 || o.b("UseFmi");
    c->general = o.b("Degenerate") || o.b("UseRegexp") || c->fmi_order;
    c->locate_vm = false;
    std::vector<std::pair<std::string, std::string>> given;  // (name, sequence)
    const bool from_file = !o.s("PatternFile").empty();
    if (from_file) {
        given = read_pattern_fasta(o.s("PatternFile"));
        if (given.empty()) throw OptError("no FASTA sequences found in pattern file: " + o.s("PatternFile"));
    } else {
        for (const std::string& p : o.sl("Pattern")) if (!p.empty()) given.emplace_back(p, p);
    }
    // locate.go:96-98 (a pattern file: bytes.Contains(seq, "\t ") -- the two bytes in a row, as written), :143-145 (-p: any)
    for (auto& g : given) {
        if (from_file) { if (g.second.find("\t ") != std::string::npos) c->warn("space found in sequence: " + g.first, true); }
        else if (g.second.find_first_of(" \t") != std::string::npos) c->warn("space found in sequence: '" + g.first + "'", true);
    }
    if (o.b("UseRegexp")) {
        // locate.go:102-121, 153-172: the regexp branch shares the search loop of -d (FindSubmatchIndex from a moving
        // offset).  Expressions that are a fixed-length chain of literals, '.', classes and escapes become class patterns (leftmost-
        // first matching has nothing to choose there, and 16 start positions are tested per step); as soon as one
        // expression has quantifiers, alternation, groups with choices or anchors, ALL of them run on the position-
        // reporting matcher instead (regex_vm.hpp: Go's leftmost-first priorities, matches of any length).
        c->locate_vm = false;
        c->vm_progs.clear();
        std::vector<std::pair<std::string, std::string>> uniq;
        for (auto& g : given) {
            bool seen = false;
            for (auto& u : uniq) seen |= u.first == g.first;
            if (!seen) uniq.push_back(g);
        }
        std::vector<std::vector<ByteSet>> chains;
        for (auto& g : uniq) {
            const std::string expr = o.b("IgnoreCase") ? "(?i)" + g.second : g.second;  // :104-106
            bool chain = false;
            RegexProgram pr;
            try {
                pr = compile_regex(expr);
                chain = pr.npos > 0 && !pr.nullable && pr.first == 1ull && pr.last == (1ull << (pr.npos - 1)) &&
                        pr.accept[RE_SYM_BEGIN] == 0 && pr.accept[RE_SYM_END] == 0;
                for (uint32_t q = 0; chain && q < pr.npos; ++q)
                    chain = pr.follow[q >> 3][1u << (q & 7)] == (q + 1 < pr.npos ? (1ull << (q + 1)) : 0ull);
            } catch (const OptError& e) {
                if (std::string(e.what()).rfind("libbsk:", 0) != 0) throw;  // a syntax error is one in any engine
            }
            if (chain) {
                std::vector<ByteSet> sets(pr.npos);
                for (uint32_t q = 0; q < pr.npos; ++q) {
                    sets[q].fill(0);
                    for (int b = 0; b < 256; ++b)
                        if ((pr.accept[b] >> q) & 1ull) set_add(sets[q], (uint8_t)b);
                }
                chains.push_back(sets);
            } else {
                c->locate_vm = true;
            }
        }
        c->locate_pre.clear();
        if (c->locate_vm) {
            // the position-reporting matcher costs ~35 ns per base and lane; most records hold no match at all, and WHETHER
            // one exists is what the boolean automaton of grep -r answers ten times faster: it goes first (expressions it
            // does not take -- more than 64 positions -- leave the matcher alone with every record)
            try {
                for (auto& g : uniq) c->locate_pre.push_back(compile_regex(o.b("IgnoreCase") ? "(?i)" + g.second : g.second));
            } catch (const OptError&) {
                c->locate_pre.clear();
            }
        }
        for (size_t k = 0; k < uniq.size(); ++k) {
            auto& g = uniq[k];
            c->pattern_names.push_back(g.first);
            c->pattern_disp.push_back(g.second);
            if (c->locate_vm) {
                c->vm_progs.push_back(compile_vm(o.b("IgnoreCase") ? "(?i)" + g.second : g.second));
                c->patterns.push_back("N");  // (the match length comes from the matcher)
                c->pattern_cls.push_back(std::vector<ByteSet>(1, ByteSet{}));
            } else {
                c->pattern_cls.push_back(chains[k]);
                c->patterns.push_back(std::string(chains[k].size(), 'N'));  // carries the match length only
            }
        }
        return;
    }
    for (auto& g : given) {  // locate.go:86-190
        std::string eff = g.second;
        if (!o.b("Degenerate") && o.b("IgnoreCase"))
            for (auto& ch : eff) if (ch >= 'A' && ch <= 'Z') ch += 32;
        const uint8_t* b = (const uint8_t*)eff.data();
        const bool legal = alphabet_valid_letters(AB_DNAredundant, b, eff.size()) || alphabet_valid_letters(AB_RNAredundant, b, eff.size()) ||
                           alphabet_valid_letters(AB_PROTEIN, b, eff.size());
        if (c->max_mm > 0) {
            if (c->max_mm > (int)eff.size()) throw OptError("mismatch should be <= length of sequence: " + g.second);
            if (!legal) throw OptError("illegal DNA/RNA/Protein sequence: " + g.first);
        } else if (!o.b("Degenerate") && (eff.find('.') != std::string::npos || !legal)) {
            throw OptError("illegal DNA/RNA/Protein sequence: " + g.first + ", you may switch on -d/--degenerate or -r/--use-regexp");
        }
        if (std::find(c->pattern_names.begin(), c->pattern_names.end(), g.first) != c->pattern_names.end()) continue;
        if (c->general)
            // -d: Degenerate2Regexp with the alphabet of -t (records of a pattern file are seq.Unlimit => nucleotide map)
            c->pattern_cls.push_back(class_sets(eff, o.b("Degenerate"), !from_file && c->alphabet == AB_PROTEIN,
                                                o.b("IgnoreCase")));
        c->pattern_names.push_back(g.first);
        c->patterns.push_back(eff);
    }
}

int locate_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    int rc = BSK_OK;
    // exact patterns on FASTQ: only the records that hold an occurrence produce rows, and the streaming filter finds them
    // (stream_filter.hip); k_locate then computes the rows of those records exactly as before
    bool filtered = false;
    if (format == BSK_FORMAT_FASTQ && n > 0 && !c->general && !o.b("UseRegexp") && !o.b("Circular") && !c->patterns.empty() &&
        c->pattern_disp.empty()) {
        Alphabet fab = partition_alphabet(c, d_buf, n, format, st, &rc);
        if (rc != BSK_OK) return rc;
        if (fab == AB_NONE) fab = AB_UNLIMIT;
        std::vector<std::string> all = c->patterns;
        for (auto& p : c->patterns) all.push_back(revcom_pattern(p, fab));
        rc = upload_patterns(c, all, st);
        if (rc != BSK_OK) return rc;
        FilterDev F;
        const size_t nuse = o.b("OnlyPositiveStrand") ? c->patterns.size() : all.size();  // locate.go:669 tests the option only
        if (make_filter(c, all, nuse, false, o.b("IgnoreCase"), st, &F, &rc)) {
            rc = build_index_filtered(c, d_buf, n, format, st, &F);
            if (rc == BSK_OK) filtered = true;
            else if (rc != BSK_ERR_FILTER_FALLBACK) return rc;
        } else if (rc != BSK_OK) return rc;
    }
    if (!filtered) rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    // header row of partition 0 (locate.go:198-204)
    std::string header;
    const bool tsv = !(o.b("Gtf") || o.b("Bed"));
    if (tsv && c->cur_pid == 0)
        header = o.b("HideMatched") ? "seqID\tpatternName\tpattern\tstrand\tstart\tend\n"
                                    : "seqID\tpatternName\tpattern\tstrand\tstart\tend\tmatched\n";
    uint64_t total = 0, nrows = 0;
    LocateParams P;
    memset(&P, 0, sizeof P);
    TextTableH tt{nullptr, nullptr, nullptr};
    if (c->table.n > 0) {
        Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);
        if (rc != BSK_OK) return rc;
        if (ab == AB_NONE) ab = AB_UNLIMIT;
        rc = prepare_text(c, d_buf, format, st, &tt, /*flatten=*/format != BSK_FORMAT_FASTQ, false, n);  // (see grep)
        if (rc != BSK_OK) return rc;
        P.fastq = format == BSK_FORMAT_FASTQ;
        P.ignore_case = o.b("IgnoreCase");
        P.circular = o.b("Circular");
        P.non_greedy = o.b("NonGreedy");
        P.both_strands = !o.b("OnlyPositiveStrand");  // sic: locate.go:669 tests the option, not the alphabet
        if (c->fmi_order) {  // the FM-index branch does consult the alphabet (locate.go:222-227, 308-310)
            P.both_strands = !(o.b("OnlyPositiveStrand") || ab == AB_UNLIMIT || ab == AB_PROTEIN);
            P.non_greedy = 0;  // "flag -G (--non-greedy) ignored when giving flag -m" (locate.go:67-69)
        }
        P.format = o.b("Gtf") ? 2 : (o.b("Bed") ? 3 : (o.b("HideMatched") ? 1 : 0));
        P.id_mode = id_mode_of(c);
        P.npat = (int)c->patterns.size();
        std::vector<std::string> all = c->patterns;
        for (auto& p : c->patterns) all.push_back(revcom_pattern(p, ab));
        rc = upload_patterns(c, all, st);
        if (rc != BSK_OK) return rc;
        P.pat = c->d_pat;
        P.pat_off = c->d_pat_off;
        if (c->general) {
            rc = upload_classes(c, true, ab, st);
            if (rc != BSK_OK) return rc;
            uint8_t comp[256];
            complement_table(ab, comp);
            if (!c->d_lut) HIP_TRYX(c, hipMalloc((void**)&c->d_lut, 256));
            HIP_TRYX(c, hipMemcpyAsync(c->d_lut, comp, 256, hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));  // comp lives on the host stack
            P.general = 1;
            P.max_mm = c->max_mm;
            P.cls = c->d_cls;
            P.fmi_order = c->fmi_order;
            P.matched_lower = !o.b("Degenerate") && !o.b("UseRegexp") && o.b("IgnoreCase");  // locate.go:430-432 lower-cases the text
            P.comp = c->d_lut;
        }
        {
            std::vector<uint8_t> bytes;
            std::vector<uint32_t> off{0};
            for (auto& p : c->pattern_names) {
                bytes.insert(bytes.end(), p.begin(), p.end());
                off.push_back((uint32_t)bytes.size());
            }
            for (auto& p : c->pattern_disp) {  // after the names: off[npat + k] .. off[npat + k + 1]
                bytes.insert(bytes.end(), p.begin(), p.end());
                off.push_back((uint32_t)bytes.size());
            }
            rc = grow(c, &c->d_names, &c->names_cap, bytes.size() + 16);
            if (rc != BSK_OK) return rc;
            rc = grow(c, &c->d_names_off, &c->names_off_cap, off.size());
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemcpyAsync(c->d_names, bytes.data(), bytes.size(), hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipMemcpyAsync(c->d_names_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
        }
        P.name = c->d_names;
        P.name_off = c->d_names_off;
        if (!c->pattern_disp.empty()) { P.disp = c->d_names; P.disp_off = c->d_names_off + c->pattern_names.size(); }
        rc = ensure_record_scratch(c);
        if (rc != BSK_OK) return rc;
        rc = grow(c, &c->d_hit_list, &c->hit_list_cap, c->table.n, c->table.n / 8 + 16);
        if (rc != BSK_OK) return rc;
        P.hit_list = c->d_hit_list;
        P.hit_count = c->d_counter;
        // chromosome-sized sequences: one wave per (pattern, strand, chunk) cell instead of one group per record
        // (not with --non-greedy, whose search position depends on the previous match)
        uint64_t ncells_total = 0;
        const uint64_t per_cells = (uint64_t)P.npat * (P.both_strands ? 2 : 1);
        bool long_checked = false;
        if (!P.non_greedy && per_cells < 32768 && !c->locate_vm) {  // (the matcher of variable-length -r walks every record itself)  // (cells of one record are counted in 32 bits: chunks <= 2^17)
            const char* e = c->tune.get("long_bytes");
            const uint32_t thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
            rc = grow(c, &c->d_long_list, &c->long_list_cap, c->table.n, c->table.n / 8 + 16);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
            HIP_TRYX(c, launch_find_long(c->table.l_seq, c->table.n, thresh, c->d_long_list, c->d_counter + 2, st));
            uint64_t lc[2] = {0, 0};
            HIP_TRYX(c, hipMemcpyAsync(lc, c->d_counter + 2, sizeof lc, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            long_checked = true;
            if (lc[0]) {
                const uint64_t nl = lc[0];
                auto al = [](uint64_t b) { return (b + 15) & ~15ull; };
                // cell counts -> cellbase (record order does not matter: every record has its own rows)
                const uint64_t o_nc = 0, o_cb = al(nl * 4), meta = o_cb + al((nl + 1) * 8);
                rc = grow(c, &c->d_cellmeta, &c->cellmeta_cap, meta, meta / 8 + 64);
                if (rc != BSK_OK) return rc;
                rc = grow(c, &c->d_scan_tmp, &c->scan_tmp_cap, 2 * ((nl + 2047) / 2048) + 4, 16);
                if (rc != BSK_OK) return rc;
                P.long_list = c->d_long_list;
                P.long_count = nl;
                P.cellbase = (const uint64_t*)(c->d_cellmeta + o_cb);
                HIP_TRYX(c, launch_locate_long_cells(c->table, P, (uint32_t*)(c->d_cellmeta + o_nc), st));
                HIP_TRYX(c, launch_scan_u32((const uint32_t*)(c->d_cellmeta + o_nc), const_cast<uint64_t*>(P.cellbase), nl,
                                            c->d_scan_tmp, st));
                HIP_TRYX(c, hipMemcpyAsync(&ncells_total, P.cellbase + nl, 8, hipMemcpyDeviceToHost, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                const uint64_t o_off = al(ncells_total * 4), need = o_off + al((ncells_total + 1) * 8);
                rc = grow(c, &c->d_cells, &c->cells_cap, need, need / 8 + 64);
                if (rc != BSK_OK) return rc;
                rc = grow(c, &c->d_scan_tmp, &c->scan_tmp_cap, 2 * ((ncells_total + 2047) / 2048) + 4, 16);
                if (rc != BSK_OK) return rc;
                P.cell_bytes = (uint32_t*)c->d_cells;
                P.cell_off = (const uint64_t*)(c->d_cells + o_off);
                P.long_cells = ncells_total;
                P.long_thresh = thresh;
            }
        }
        HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 2 * sizeof(uint64_t), st));
        if (c->locate_vm) {
            rc = grow(c, &c->d_vm_progs, &c->vm_progs_cap, c->vm_progs.size());
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemcpyAsync(c->d_vm_progs, c->vm_progs.data(), c->vm_progs.size() * sizeof(VmProgram), hipMemcpyHostToDevice, st));
            bool pre = false;
            // (--circular: an occurrence across the origin is invisible to the boolean pass over the plain text)
            if (!c->locate_pre.empty() && !c->tune.get("locate_nopre") && !P.circular) {
                rc = grow(c, &c->d_regex, &c->regex_cap, c->locate_pre.size());
                if (rc != BSK_OK) return rc;
                HIP_TRYX(c, hipMemcpyAsync(c->d_regex, c->locate_pre.data(), c->locate_pre.size() * sizeof(RegexProgram), hipMemcpyHostToDevice, st));
                GrepParams G;
                memset(&G, 0, sizeof G);
                G.fastq = P.fastq;
                G.by_seq = 1;
                G.both_strands = P.both_strands;
                G.npat = (int)c->locate_pre.size();
                G.regex = c->d_regex;
                G.comp = P.comp;
                HIP_TRYX(c, launch_grep_match(d_buf, n, c->table, &tt, G, c->d_out_len, st, c->avg_record_bytes));  // != 0: some match exists
                pre = true;
                P.pre_regex = c->d_regex;
            }
            uint64_t ncand = 0;
            if (pre) {  // the candidates as a list: the matcher then runs with every lane busy
                HIP_TRYX(c, launch_compact_hits(c->d_out_len, c->table.n, c->d_hit_list, c->d_counter, st));
                HIP_TRYX(c, hipMemcpyAsync(&ncand, c->d_counter, sizeof ncand, hipMemcpyDeviceToHost, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 2 * sizeof(uint64_t), st));
            }
            HIP_TRYX(c, launch_locate_vm(false, d_buf, n, c->table, tt, P, c->d_vm_progs, c->d_out_len, nullptr, nullptr, c->d_counter + 1, st,
                                         pre ? c->d_hit_list : nullptr, ncand));
        } else {
            // -d / -m: whether a record holds an occurrence at all is what grep's Shift-And answers at one table lookup per
            // base; the position-reporting search (one class test per start position and pattern byte) then runs on the
            // few records that do.  Long records have their own cell launches and are not prefiltered.
            size_t longest = 0;
            for (auto& p : all) longest = std::max(longest, p.size());
            if (c->general && long_checked && !P.long_count && !P.circular && longest <= 64 && all.size() <= 8 && c->max_mm <= 3 &&
                !c->tune.get("locate_nopre")) {
                GrepParams G;
                memset(&G, 0, sizeof G);
                G.fastq = P.fastq;
                G.by_seq = 1;
                G.both_strands = P.both_strands;
                G.npat = P.npat;
                G.pat = P.pat;
                G.pat_off = P.pat_off;
                G.general = 1;
                G.max_mm = P.max_mm;
                G.cls = P.cls;
                G.sa_ok = 1;
                HIP_TRYX(c, launch_grep_match(d_buf, n, c->table, &tt, G, c->d_out_len, st, c->avg_record_bytes));
                HIP_TRYX(c, launch_compact_hits(c->d_out_len, c->table.n, c->d_hit_list, c->d_counter, st));
                HIP_TRYX(c, hipMemcpyAsync(&P.ncand, c->d_counter, sizeof P.ncand, hipMemcpyDeviceToHost, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 2 * sizeof(uint64_t), st));
                P.cand = c->d_hit_list;
            }
            if (!P.cand || P.ncand)
                HIP_TRYX(c, launch_locate(false, d_buf, n, c->table, tt, P, c->d_out_len, nullptr, nullptr, c->d_counter + 1, st, c->avg_record_bytes));
            P.cand = nullptr;
        }
        if (P.long_count) {
            // place every cell inside its record's rows, then the record sizes
            HIP_TRYX(c, launch_scan_u32(P.cell_bytes, const_cast<uint64_t*>(P.cell_off), ncells_total, c->d_scan_tmp, st));
            HIP_TRYX(c, launch_locate_long_sizes(P, c->d_out_len, st));
        }
        HIP_TRYX(c, launch_compact_hits(c->d_out_len, c->table.n, c->d_hit_list, c->d_counter, st));
        HIP_TRYX(c, launch_scan_u32(c->d_out_len, c->d_out_off, c->table.n, c->d_scan_tmp, st));
        uint64_t status = 0;
        HIP_TRYX(c, hipMemcpyAsync(&total, c->d_out_off + c->table.n, sizeof total, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipMemcpyAsync(&P.nhit, c->d_counter, sizeof P.nhit, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        rc = kernel_error_to_status(c, status);
        if (rc != BSK_OK) return rc;
    } else {
        rc = empty_result(c, out);
        if (rc != BSK_OK) return rc;
    }
    out->d_data = nullptr;
    out->len = 0;
    out->records = 0;
    if (total + header.size() == 0) return BSK_OK;
    rc = ensure_out(c, total + header.size());
    if (rc != BSK_OK) return rc;
    if (!header.empty()) HIP_TRYX(c, hipMemcpyAsync(c->d_out, header.data(), header.size(), hipMemcpyHostToDevice, st));
    if (total && c->locate_vm)
        HIP_TRYX(c, launch_locate_vm(true, d_buf, n, c->table, tt, P, c->d_vm_progs, c->d_out_len, c->d_out_off, c->d_out + header.size(),
                                     c->d_counter + 1, st, P.hit_list, P.nhit));
    else if (total)
        HIP_TRYX(c, launch_locate(true, d_buf, n, c->table, tt, P, c->d_out_len, c->d_out_off, c->d_out + header.size(),
                                  c->d_counter + 1, st, c->avg_record_bytes));
    if (total) HIP_TRYX(c, hipMemcpyAsync(&nrows, c->d_counter + 1, sizeof nrows, hipMemcpyDeviceToHost, st));  // counted by the emit pass
    HIP_TRYX(c, hipStreamSynchronize(st));  // header lives on the host stack
    out->d_data = c->d_out;
    out->len = total + header.size();
    out->records = nrows + (header.empty() ? 0 : 1);
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// subseq --gtf / --bed: feature files (host side of SubseqTransform.Before, subseq.go:100-165)
//   ReadBedFilteredFeatures  bigseqkit-lib/subseq.go:242-310 (in tree)
//   gtf.ReadFilteredFeatures shenwei356/bio featio/gtf (not in tree; PARITY.md GTF)
// Only the FIRST feature of a sequence name is ever used (subseq.go:426, 523 return inside the loop, Q7), so that
// is all the context keeps: name -> (flank-adjusted start, end, strand, header suffix).
// ---------------------------------------------------------------------------
static std::vector<std::string> split_tabs(const std::string& line) {
    std::vector<std::string> items;
    for (size_t i = 0;;) {
        size_t j = line.find('\t', i);
        if (j == std::string::npos) { items.emplace_back(line, i); break; }
        items.emplace_back(line, i, j - i);
        i = j + 1;
    }
    return items;
}

static bool atoi_strict(const std::string& s, long long* v) {
    if (s.empty() || isspace((unsigned char)s[0])) return false;
    char* e = nullptr;
    *v = strtoll(s.c_str(), &e, 10);
    return *e == 0;
}

static std::string lower_str(std::string s) {
    for (auto& ch : s) if (ch >= 'A' && ch <= 'Z') ch += 32;
    return s;
}

static void load_features(bsk_ctx* c) {
    const Options& o = c->opts;
    const bool gtf = !o.s("Gtf").empty();
    const std::vector<std::string>& chrs = o.sl("Chr");
    std::vector<std::string> feats;
    for (auto& f : o.sl("Feature")) feats.push_back(lower_str(f));
    if (!gtf && !feats.empty()) throw OptError("when given flag -b (--bed), flag -f (--feature) is not allowed");
    const int64_t up = o.i("UpStream"), down = o.i("DownStream");
    const bool only = o.b("OnlyFlank");
    std::string flank;
    if (up > 0) {
        if (only) flank = "_usf:" + std::to_string(up);
        else if (down > 0) flank = "_us:" + std::to_string(up) + "_ds:" + std::to_string(down);
        else flank = "_us:" + std::to_string(up);
    } else if (down > 0) {
        flank = only ? "_dsf:" + std::to_string(down) : "_ds:" + std::to_string(down);
    }
    c->features.clear();
    c->features_uploaded = false;
    std::unordered_set<std::string> seen;
    c->info(gtf ? "read GTF file ..." : "read BED file ...", /*unless_quiet=*/true);  // subseq.go:98-100, 131-133
    size_t nloaded = 0;  // len(features) of the reference: every accepted line, before the per-name map keeps the first
    for (const std::string& line : read_pattern_lines(gtf ? o.s("Gtf") : o.s("Bed"))) {
        if (line.empty() || line[0] == '#') continue;
        if (!gtf && ((line.size() > 7 && line.compare(0, 7, "browser") == 0) || (line.size() > 5 && line.compare(0, 5, "track") == 0)))
            continue;
        const auto items = split_tabs(line);
        if (gtf ? items.size() != 9 : items.size() < 3) continue;
        if (!chrs.empty() && std::find(chrs.begin(), chrs.end(), items[0]) == chrs.end()) continue;
        if (gtf && !feats.empty() && std::find(feats.begin(), feats.end(), lower_str(items[2])) == feats.end()) continue;
        long long st, en;
        const std::string &sst = items[gtf ? 3 : 1], &sen = items[gtf ? 4 : 2];
        if (!atoi_strict(sst, &st)) throw OptError(items[0] + ": bad start: " + sst);
        if (!atoi_strict(sen, &en)) throw OptError(items[0] + ": bad end: " + sen);
        std::string strand = ".", label;
        if (gtf) {
            if (st > en) throw OptError(items[0] + ": start (" + std::to_string(st) + ") must be < end (" + std::to_string(en) + ")");
            if (items[6] != "+" && items[6] != "-" && items[6] != ".") throw OptError("bad strand: " + items[6]);
            strand = items[6];
            const std::string& at = items[8];  // tag "value"; tag "value";
            for (size_t i = 0; i < at.size();) {
                size_t j = at.find(';', i);
                if (j == std::string::npos) j = at.size();
                std::string item(at, i, j - i);
                i = j + 1;
                const size_t a0 = item.find_first_not_of(' ');
                if (a0 == std::string::npos) continue;
                item.erase(0, a0);
                const size_t sp = item.find(' ');
                if (sp == std::string::npos) continue;
                std::string v(item, sp + 1);
                while (!v.empty() && v.back() == ' ') v.pop_back();
                if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);
                if (item.compare(0, sp, o.s("GtfTag")) == 0 && sp == o.s("GtfTag").size()) { label = v; break; }
            }
        } else {
            if (st >= en) throw OptError(items[0] + ": start (" + std::to_string(st) + ") must be <= end (" + std::to_string(en) + ")");
            st += 1;  // BED start is 0-based (subseq.go:294)
            if (items.size() >= 4) label = items[3];
            if (items.size() >= 6) {
                if (items[5] != "+" && items[5] != "-" && items[5] != ".") throw OptError("bad strand: " + items[5]);
                strand = items[5];
            }
        }
        ++nloaded;
        const std::string key = lower_str(items[0]);
        if (!seen.insert(key).second) continue;  // a later feature of the same name is never reached
        bsk_ctx::Feature f;
        f.name_lower = key;
        f.minus = strand == "-";
        if (f.minus) {  // subseq.go:340-352
            if (only) { if (up > 0) { f.s = en + 1; f.e = en + up; } else { f.s = st - down; f.e = st - 1; } }
            else { f.s = st - down; f.e = en + up; }
        } else {        // subseq.go:359-371
            if (only) { if (up > 0) { f.s = st - up; f.e = st - 1; } else { f.s = en + 1; f.e = en + down; } }
            else { f.s = st - up; f.e = en + down; }
        }
        f.suffix = "_" + std::to_string(st) + "-" + std::to_string(en) + ":" + strand + flank + " " + label;
        c->features.push_back(f);
    }
    c->info(std::to_string(nloaded) + (gtf ? " GTF" : " BED") + " features loaded", true);  // subseq.go:127-129, 157-159
}

// ---------------------------------------------------------------------------
// subseq by region  (SubseqTransform, bigseqkit-lib/subseq.go:36-165, 314-317)
// ---------------------------------------------------------------------------
void validate_subseq_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    if (o.b("OnlyFlank")) {
        if (o.i("UpStream") > 0 && o.i("DownStream") > 0)
            throw OptError("when flag -f (--only-flank) given, only one of flags -u (--up-stream) and -d (--down-stream) is allowed");
        else if (o.i("UpStream") == 0 && o.i("DownStream") == 0)
            throw OptError("when flag -f (--only-flank) given, one of flags -u (--up-stream) and -d (--down-stream) should be given");
    }
    if (!o.s("Region").empty()) {
        if (o.i("UpStream") > 0 || o.i("DownStream") > 0 || o.b("OnlyFlank"))
            throw OptError("when flag -r (--region) given, any of flags -u (--up-stream), -d (--down-stream) and -f (--only-flank) is not allowed");
        c->region_on = true;
        parse_region_opt(o.s("Region"), "subseq", &c->region_start, &c->region_end);
    } else if (!o.s("Gtf").empty() || !o.s("Bed").empty()) {
        load_features(c);
    } else {
        throw OptError("one of the options needed: -r/--region, --bed, --gtf");
    }
}

static int upload_features(bsk_ctx* c, hipStream_t st) {
    const size_t nf = c->features.size();
    uint64_t slots = 16;
    while (slots < 2 * nf) slots <<= 1;
    std::vector<uint64_t> keys(slots, 0);
    std::vector<uint32_t> idx(slots, 0), name_off{0}, suf_off{0};
    std::vector<int64_t> fs(nf), fe(nf);
    std::vector<uint8_t> minus(nf), names, sufs;
    for (size_t k = 0; k < nf; ++k) {
        const auto& f = c->features[k];
        uint64_t h = 1469598103934665603ull;
        for (unsigned char ch : f.name_lower) h = (h ^ ch) * 1099511628211ull;
        if (!h) h = 1;
        uint64_t s = h & (slots - 1);
        while (keys[s]) s = (s + 1) & (slots - 1);
        keys[s] = h;
        idx[s] = (uint32_t)k;
        names.insert(names.end(), f.name_lower.begin(), f.name_lower.end());
        name_off.push_back((uint32_t)names.size());
        sufs.insert(sufs.end(), f.suffix.begin(), f.suffix.end());
        suf_off.push_back((uint32_t)sufs.size());
        fs[k] = f.s; fe[k] = f.e; minus[k] = f.minus;
    }
    // one allocation, every array 16-byte aligned
    const void* src[8] = {keys.data(), idx.data(), name_off.data(), suf_off.data(), fs.data(), fe.data(), minus.data(), nullptr};
    const uint64_t bytes[8] = {slots * 8, slots * 4, name_off.size() * 4, suf_off.size() * 4, nf * 8, nf * 8, nf, 0};
    uint64_t off = 0;
    for (int a = 0; a < 7; ++a) { c->feat_off[a] = off; off += (bytes[a] + 15) & ~15ull; }
    const uint64_t names_at = off;
    off += (names.size() + 15) & ~15ull;
    const uint64_t sufs_at = off;
    off += (sufs.size() + 15) & ~15ull;
    c->feat_off[7] = names_at;
    int rc = grow(c, &c->d_feat, &c->feat_cap, off + 16);
    if (rc != BSK_OK) return rc;
    for (int a = 0; a < 7; ++a)
        if (bytes[a]) HIP_TRYX(c, hipMemcpyAsync(c->d_feat + c->feat_off[a], src[a], bytes[a], hipMemcpyHostToDevice, st));
    if (!names.empty()) HIP_TRYX(c, hipMemcpyAsync(c->d_feat + names_at, names.data(), names.size(), hipMemcpyHostToDevice, st));
    if (!sufs.empty()) HIP_TRYX(c, hipMemcpyAsync(c->d_feat + sufs_at, sufs.data(), sufs.size(), hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    c->feat_slots = slots;
    c->feat_off[8] = sufs_at;
    c->features_uploaded = true;
    return BSK_OK;
}

// the feature set of the context (subseq --gtf / --bed, faidx region queries) on the device, bound to P
int bind_features(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, SeqParams* P) {
    int rc = BSK_OK;
    if (!c->features_uploaded) {
        rc = upload_features(c, st);
        if (rc != BSK_OK) return rc;
    }
    Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);
    if (rc != BSK_OK) return rc;
    uint8_t comp[256];
    complement_table(ab, comp);
    if (!c->d_lut) HIP_TRYX(c, hipMalloc((void**)&c->d_lut, 256));
    HIP_TRYX(c, hipMemcpyAsync(c->d_lut, comp, 256, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));  // comp lives on the host stack
    const uint8_t* base = c->d_feat;
    P->feat_on = 1;
    P->fset_keys = (const uint64_t*)(base + c->feat_off[0]);
    P->fset_idx = (const uint32_t*)(base + c->feat_off[1]);
    P->fset_mask = c->feat_slots - 1;
    P->fname_off = (const uint32_t*)(base + c->feat_off[2]);
    P->fsuffix_off = (const uint32_t*)(base + c->feat_off[3]);
    P->f_s = (const int64_t*)(base + c->feat_off[4]);
    P->f_e = (const int64_t*)(base + c->feat_off[5]);
    P->f_minus = base + c->feat_off[6];
    P->fname = base + c->feat_off[7];
    P->fsuffix = base + c->feat_off[8];
    P->comp = c->d_lut;
    return BSK_OK;
}

// `subseq -r a:b` on FASTQ: the records leave from the streaming pass itself (stream_subseq.hip) -- per-range slices sized
// from the shard head, one scan over the ranges, one gather.  BSK_ERR_FILTER_FALLBACK: not this path's input (long lines,
// a slice too small); the caller takes the record-table path.
static int subseq_stream_run(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st, bsk_out* out) {
    const int blocks = std::max(1, c->num_cus * subseq_stream_max_blocks_per_cu(c->use_dpp));
    uint32_t nranges = 0;
    uint64_t chunk = 0;
    int rc = prep_ranges(c, d_buf, n, /*fastq=*/true, blocks, st, &nranges, &chunk);
    if (rc != BSK_OK) return rc;
    const size_t hb = std::min<size_t>(n, 256 * 1024);
    std::vector<uint8_t> head(hb);
    HIP_TRYX(c, hipMemcpyAsync(head.data(), d_buf, hb, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (!c->norm_active && fastq_head_multiline(head.data(), hb)) return BSK_ERR_MULTILINE_FASTQ;
    // output bytes per input byte over the complete records of the sample
    uint64_t in_b = 0, out_b = 0, line = 0, line_start = 0, rec_out = 0, max_line = 0;
    for (size_t i = 0; i < hb; ++i) {
        if (head[i] != '\n') continue;
        const uint64_t ll = i - line_start;
        max_line = std::max(max_line, ll);
        const uint32_t role = (uint32_t)(line & 3);
        if (role == 0) rec_out = ll + 1;
        else if (role == 2) rec_out += 2;
        else {
            uint32_t b, e;
            sub_location((uint32_t)ll, c->region_start, c->region_end, &b, &e);
            rec_out += (uint64_t)(e - b) + 1;
        }
        if (role == 3) { out_b += rec_out; in_b = i + 1; }
        ++line;
        line_start = i + 1;
    }
    max_line = std::max<uint64_t>(max_line, hb - line_start);
    // a lane copies its piece alone: lines of kilobytes (long reads) stay with the record-table kernels
    if (max_line > 2048 || in_b == 0) return BSK_ERR_FILTER_FALLBACK;
    double ratio = (double)(out_b + 64) / (double)in_b;
    if (const char* sc = c->tune.get("subseq_scale")) ratio *= atof(sc);  // tests: force the overflow -> fallback route
    uint64_t slice_cap = (uint64_t)((double)chunk * ratio * 1.25) + (c->tune.get("subseq_scale") ? 16 : 4096);
    slice_cap = (slice_cap + 15) & ~(uint64_t)15;
    if (slice_cap >= (1ull << 32) || slice_cap * nranges > 2 * (uint64_t)n + (64ull << 20)) return BSK_ERR_FILTER_FALLBACK;
    rc = grow(c, &c->d_slices, &c->slices_cap, slice_cap * nranges, 256);
    if (rc != BSK_OK) return rc;
    rc = grow(c, &c->d_names_aux, &c->names_aux_cap, 2 * ((uint64_t)nranges + 2), 16);
    if (rc != BSK_OK) return rc;
    SubseqDev D;
    D.slices = c->d_slices;
    D.slice_cap = slice_cap;
    D.range_bytes = c->d_names_aux;
    D.range_count = c->d_range_count;
    D.status = c->d_status;
    D.region_start = c->region_start;
    D.region_end = c->region_end;
    uint64_t* d_count_base = c->d_names_aux + nranges + 2;
    {
        Timed t(c, "k_subseq_stream", st);
        HIP_TRYX(c, launch_subseq_stream(c->use_dpp, blocks, d_buf, n, c->d_anchors, nranges,
                                         reinterpret_cast<uint32_t*>(c->d_anchors + (size_t)nranges + 1), D, st));
    }
    HIP_TRYX(c, launch_scan_small(D.range_bytes, c->d_range_base, nranges, st));
    HIP_TRYX(c, launch_scan_small(D.range_count, d_count_base, nranges, st));
    uint64_t total = 0, records = 0, status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&total, c->d_range_base + nranges, sizeof total, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&records, d_count_base + nranges, sizeof records, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status & ERR_CAPACITY) {
        status &= ~(uint64_t)ERR_CAPACITY;
        HIP_TRYX(c, hipMemcpyAsync(c->d_status, &status, sizeof status, hipMemcpyHostToDevice, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (status == 0) return BSK_ERR_FILTER_FALLBACK;
    }
    if (status) return kernel_error_to_status(c, status);
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    if (total) {
        Timed t(c, "k_subseq_compact", st);
        NamesDev G;
        memset(&G, 0, sizeof G);
        G.slices = D.slices;
        G.slice_cap = D.slice_cap;
        G.range_bytes = D.range_bytes;
        HIP_TRYX(c, launch_names_compact(G, c->d_range_base, nranges, c->d_out, st));
    }
    c->table.n = 0;  // no record table was built for this shard
    out->d_data = c->d_out;
    out->len = total;
    out->records = records;
    return BSK_OK;
}

int subseq_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const bool fastq = format == BSK_FORMAT_FASTQ;
    if (fastq && n > 0 && c->region_on && !c->tune.is("subseq", "table")) {
        const int rcs = subseq_stream_run(c, d_buf, n, st, out);
        if (rcs != BSK_ERR_FILTER_FALLBACK) return rcs;
    }
    int rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    SeqParams P = format_params(c, fastq);
    P.buf_end = d_buf + n;
    if (c->region_on) {
        P.region_on = 1;
        P.region_start = c->region_start;
        P.region_end = c->region_end;
    } else {
        if (c->features.empty()) return empty_result(c, out);  // no record can have a feature
        rc = bind_features(c, d_buf, n, format, st, &P);
        if (rc != BSK_OK) return rc;
    }
    {   // wrapped FASTA: random access through the text view instead of the sequential per-record walk
        TextTableH tt{nullptr, nullptr, nullptr};
        rc = prepare_text(c, d_buf, format, st, &tt);
        if (rc != BSK_OK) return rc;
        P.text_w = tt.text_w; P.lin_off = tt.lin_off; P.lin = tt.lin;
    }
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

// ---------------------------------------------------------------------------
// FASTA text view: classify every record, linearise the irregularly wrapped ones
// ---------------------------------------------------------------------------
int prepare_text(bsk_ctx* c, const uint8_t* d_buf, int format, hipStream_t st, TextTableH* tt, bool flatten, bool keep_out_len, uint64_t buf_n) {
    tt->text_w = nullptr;
    tt->lin_off = nullptr;
    tt->lin = nullptr;
    tt->lin_n = 0;
    c->flat_long_count = 0;
    c->flat_long_thresh = 0;
    if (format == BSK_FORMAT_FASTQ || c->table.n == 0) return BSK_OK;
    const uint64_t n = c->table.n;
    if (n + 1 > c->text_cap || !c->d_lin_off) {
        if (c->d_text_w) HIP_TRYX(c, hipFree(c->d_text_w));
        if (c->d_lin_off) HIP_TRYX(c, hipFree(c->d_lin_off));
        c->d_text_w = nullptr; c->d_lin_off = nullptr;
        const uint64_t cap = n + n / 8 + 16;
        HIP_TRYX(c, hipMalloc((void**)&c->d_text_w, cap * sizeof(uint32_t)));
        HIP_TRYX(c, hipMalloc((void**)&c->d_lin_off, (cap + 1) * sizeof(uint64_t)));
        c->text_cap = cap;
    }
    int rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    // The line layout of every record comes out of the index pass (RecordTable::text_w); BSK_TEXT=classify keeps the
    // separate pass over the line ends (tests cross-check the two).
    const char* mode = c->tune.get("text");
    const uint32_t* text_w = c->table.text_w;
    if (keep_out_len) {
        // the views once more AFTER the sizes of the output were computed (they sit in d_out_len): the lengths of the
        // linear copies go through d_text_w, which the views do not use
        HIP_TRYX(c, launch_lin_len(c->table, c->d_text_w, st));
        HIP_TRYX(c, launch_scan_u32(c->d_text_w, c->d_lin_off, n, c->d_scan_tmp, st));
        uint64_t total = 0;
        HIP_TRYX(c, hipMemcpyAsync(&total, c->d_lin_off + n, sizeof total, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (total) {
            rc = grow(c, &c->d_lin, &c->lin_cap, total, total / 8 + 64);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, launch_text_linearise(d_buf, c->table, text_w, c->d_lin_off, c->d_lin, st));
        }
        tt->lin_n = total;
        tt->text_w = text_w;
        tt->lin_off = c->d_lin_off;
        tt->lin = c->d_lin;
        return BSK_OK;
    }
    if (flatten && !(mode && strcmp(mode, "view") == 0)) {
        // every wrapped record gets a linear copy; the kernels then read contiguous text (BSK_TEXT=view keeps the views)
        HIP_TRYX(c, launch_lin_len_all(c->table, c->d_out_len, c->d_text_w, st));
        text_w = c->d_text_w;
    } else if (mode && strcmp(mode, "classify") == 0) {
        HIP_TRYX(c, launch_text_classify(d_buf, c->table, c->d_text_w, c->d_out_len, st));
        text_w = c->d_text_w;
    } else {
        HIP_TRYX(c, launch_lin_len(c->table, c->d_out_len, st));
    }
    HIP_TRYX(c, launch_scan_u32(c->d_out_len, c->d_lin_off, n, c->d_scan_tmp, st));
    uint64_t total = 0;
    uint64_t lc[2] = {0, 0};
    const bool flat = text_w == c->d_text_w && flatten;
    const char* lenv = c->tune.get("long_bytes");
    const uint32_t long_thresh = lenv && atoll(lenv) > 0 ? (uint32_t)atoll(lenv) : SEQ_LONG_THRESH;
    if (flat) {  // chromosome-sized records are copied by whole blocks: list them (read back with the total, one wait)
        rc = grow(c, &c->d_long_list, &c->long_list_cap, n, n / 8 + 16);
        if (rc != BSK_OK) return rc;
        if (!c->d_counter) HIP_TRYX(c, hipMalloc((void**)&c->d_counter, 4 * sizeof(uint64_t)));
        HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
        HIP_TRYX(c, launch_find_long(c->table.l_seq, n, long_thresh, c->d_long_list, c->d_counter + 2, st));
        HIP_TRYX(c, hipMemcpyAsync(lc, c->d_counter + 2, sizeof lc, hipMemcpyDeviceToHost, st));
    }
    HIP_TRYX(c, hipMemcpyAsync(&total, c->d_lin_off + n, sizeof total, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    c->flat_long_count = flat ? lc[0] : 0;
    c->flat_long_thresh = flat ? long_thresh : 0u;
    if (total) {
        rc = grow(c, &c->d_lin, &c->lin_cap, total, total / 8 + 64);
        if (rc != BSK_OK) return rc;
        if (flat) HIP_TRYX(c, launch_text_flatten(d_buf, buf_n, c->table, c->d_lin_off, c->d_lin, st, c->d_long_list, lc[0], lc[1], long_thresh));
        else HIP_TRYX(c, launch_text_linearise(d_buf, c->table, text_w, c->d_lin_off, c->d_lin, st));
    }
    tt->lin_n = total;
    tt->text_w = text_w;
    tt->lin_off = c->d_lin_off;
    tt->lin = c->d_lin;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// translate  (Translate.Before, bigseqkit-lib/translate.go:33-64)
// ---------------------------------------------------------------------------
#include "genetic_codes.inc"

// names of the tables as the reference lists them (bigseqkit-cli/translate.go:55-78); `translate -l 0` prints "ID\tName"
static const struct { int id; const char* name; } kCodeNames[] = {
    {1, "The Standard Code"},
    {2, "The Vertebrate Mitochondrial Code"},
    {3, "The Yeast Mitochondrial Code"},
    {4, "The Mold, Protozoan, and Coelenterate Mitochondrial Code and the Mycoplasma/Spiroplasma Code"},
    {5, "The Invertebrate Mitochondrial Code"},
    {6, "The Ciliate, Dasycladacean and Hexamita Nuclear Code"},
    {9, "The Echinoderm and Flatworm Mitochondrial Code"},
    {10, "The Euplotid Nuclear Code"},
    {11, "The Bacterial, Archaeal and Plant Plastid Code"},
    {12, "The Alternative Yeast Nuclear Code"},
    {13, "The Ascidian Mitochondrial Code"},
    {14, "The Alternative Flatworm Mitochondrial Code"},
    {16, "Chlorophycean Mitochondrial Code"},
    {21, "Trematode Mitochondrial Code"},
    {22, "Scenedesmus obliquus Mitochondrial Code"},
    {23, "Thraustochytrium Mitochondrial Code"},
    {24, "Pterobranchia Mitochondrial Code"},
    {25, "Candidate Division SR1 and Gracilibacteria Code"},
    {26, "Pachysolen tannophilus Nuclear Code"},
    {27, "Karyorelict Nuclear"},
    {28, "Condylostoma Nuclear"},
    {29, "Mesodinium Nuclear"},
    {30, "Peritrich Nuclear"},
    {31, "Blastocrithidia Nuclear"},
};

static const GeneticCode* find_code(int id) {
    for (auto& g : kGeneticCodes)
        if (g.id == id) return &g;
    return nullptr;
}

void validate_translate_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    if (!find_code((int)o.i("TranslTable"))) throw OptError("invalid translate table: " + std::to_string(o.i("TranslTable")));
    c->frames.clear();
    for (auto& f : o.sl("Frame")) {
        char* endp = nullptr;
        const long v = strtol(f.c_str(), &endp, 10);
        if (f.empty() || *endp)
            throw OptError("invalid frame(s): " + f + ". available: 1, 2, 3, -1, -2, -3, and 6 for all. multiple frames should be separated by comma");
        if (!(v == 1 || v == 2 || v == 3 || v == -1 || v == -2 || v == -3 || v == 6))
            throw OptError("invalid frame: " + std::to_string(v) + ". available: 1, 2, 3, -1, -2, -3, and 6 for all");
        if (v == 6) { c->frames = {1, 2, 3, -1, -2, -3}; break; }
        c->frames.push_back((int)v);
    }
    if (c->frames.size() > 6) throw OptError("libbsk: at most 6 frames per call");
    // translate.go:75-101: -l 0 / -L 0 list the tables; -l N / -L N print bio's CodonTable.String() /
    // StringWithAmbiguousCodons(), whose layout lives in shenwei356/bio (not in tree) -- refused, see PARITY.md
    if (o.i("ListTranslTable") > 0 || o.i("ListTranslTableWithAmbCodons") > 0)
        throw OptError("libbsk: translate -l N / -L N (the codon listing of one table) is not provided; -l 0 lists the tables");
}

// 4096-entry tables over 4-bit IUPAC codes (A=1 C=2 G=4 T=8): amino acid common to all
// expansions of the codon ('X' when they disagree), and the exact start codons
static void build_codon_tables(const GeneticCode& g, uint8_t* aa, uint8_t* start) {
    static const int tcag[4] = {8, 2, 1, 4};  // code of T, C, A, G
    auto idx64 = [&](int b1, int b2, int b3) {
        int i[3] = {b1, b2, b3}, r = 0;
        for (int k = 0; k < 3; ++k) {
            int j = 0;
            while (tcag[j] != i[k]) ++j;
            r = r * 4 + j;
        }
        return r;
    };
    memset(aa, 0, 4096);
    memset(start, 0, 4096);
    for (int c1 = 1; c1 < 16; ++c1)
        for (int c2 = 1; c2 < 16; ++c2)
            for (int c3 = 1; c3 < 16; ++c3) {
                char r = 0;
                for (int b1 = 1; b1 <= 8; b1 <<= 1) {
                    if (!(c1 & b1)) continue;
                    for (int b2 = 1; b2 <= 8; b2 <<= 1) {
                        if (!(c2 & b2)) continue;
                        for (int b3 = 1; b3 <= 8; b3 <<= 1) {
                            if (!(c3 & b3)) continue;
                            const char a = g.aa[idx64(b1, b2, b3)];
                            if (r == 0) r = a;
                            else if (r != a) r = 'X';
                        }
                    }
                }
                aa[(c1 << 8) | (c2 << 4) | c3] = (uint8_t)r;
            }
    for (int b1 = 1; b1 <= 8; b1 <<= 1)
        for (int b2 = 1; b2 <= 8; b2 <<= 1)
            for (int b3 = 1; b3 <= 8; b3 <<= 1)
                if (g.starts[idx64(b1, b2, b3)] == 'M') start[(b1 << 8) | (b2 << 4) | b3] = 1;
}

// translate.go:78-89: with -l 0 or -L 0 every Call returns the list of tables ("ID\tName", ascending ids) and reads no record
static int translate_list_tables(bsk_ctx* c, bsk_out* out) {
    std::string txt;
    uint64_t rows = 0;
    for (auto& e : kCodeNames) { txt += std::to_string(e.id) + "\t" + e.name + "\n"; ++rows; }
    int rc = ensure_out(c, txt.size());
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, hipMemcpy(c->d_out, txt.data(), txt.size(), hipMemcpyHostToDevice));
    out->d_data = c->d_out;
    out->len = txt.size();
    out->records = rows;
    return BSK_OK;
}

int translate_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    if (o.i("ListTranslTable") == 0 || o.i("ListTranslTableWithAmbCodons") == 0) return translate_list_tables(c, out);
    // FASTA: first with the record table from the '>' bytes alone (stream_fasta_light.hip) -- k_translate_wide validates the
    // whole text against the layout that table assumes; whatever does not fit (a record flagged by the wide kernel, a
    // chromosome-sized one, ...) sends the call through the full index pass below, and the context remembers it
    bool light = format == BSK_FORMAT_FASTA && c->translate_light_ok && !c->tune.is("translate_index", "full") &&
                 !c->tune.get("translate") && !o.b("InitCodonAsM");
    int rc = BSK_ERR_FILTER_FALLBACK;
    if (light) {
        rc = build_index_light(c, d_buf, n, st);
        if (rc != BSK_OK && rc != BSK_ERR_FILTER_FALLBACK) return rc;
    }
    if (rc == BSK_ERR_FILTER_FALLBACK) {
        light = false;
        rc = build_index(c, d_buf, n, format, st);
    }
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);
    if (rc != BSK_OK) return rc;
    if (!(ab == AB_DNA || ab == AB_DNAredundant || ab == AB_RNA || ab == AB_RNAredundant)) {  // translate.go:116-122
        c->set_error("command 'seqkit translate' only apply to DNA/RNA sequences");
        return BSK_ERR_FORMAT;
    }
    TextTableH tt;
    rc = prepare_text(c, d_buf, format, st, &tt);
    if (rc != BSK_OK) return rc;
    TranslateParams P;
    memset(&P, 0, sizeof P);
    P.fastq = format == BSK_FORMAT_FASTQ;
    P.nframes = (int)c->frames.size();
    for (int k = 0; k < P.nframes; ++k) P.frames[k] = c->frames[k];
    P.trim = o.b("Trim"); P.clean = o.b("Clean"); P.allow_unknown = o.b("AllowUnknownCodon");
    P.init_m = o.b("InitCodonAsM"); P.append_frame = o.b("AppendFrame");
    P.line_width = (int)o.ci("LineWidth");
    P.id_mode = id_mode_of(c);
    if (!c->d_codon) HIP_TRYX(c, hipMalloc((void**)&c->d_codon, 6 * 4096 + 256 + 16384));
    {
        std::vector<uint8_t> tab(6 * 4096 + 256 + 16384);
        uint8_t *fw = tab.data(), *stt = fw + 4096, *rcw = fw + 8192, *rcs = fw + 12288, *iu = fw + 16384;
        build_codon_tables(*find_code((int)o.i("TranslTable")), fw, stt);
        auto comp = [](int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); };
        for (int c0 = 0; c0 < 16; ++c0)
            for (int c1 = 0; c1 < 16; ++c1)
                for (int c2 = 0; c2 < 16; ++c2) {
                    const int i = (c0 << 8) | (c1 << 4) | c2, r = (comp(c2) << 8) | (comp(c1) << 4) | comp(c0);
                    rcw[i] = fw[r];
                    rcs[i] = stt[r];
                }
        memset(iu, 0, 256);
        const char* letters = "acgturyswkmbdhvn";
        const int codes[] = {1, 2, 4, 8, 8, 5, 10, 6, 9, 12, 3, 14, 13, 11, 7, 15};
        for (int k = 0; letters[k]; ++k) { iu[(uint8_t)letters[k]] = (uint8_t)codes[k]; iu[(uint8_t)(letters[k] - 32)] = (uint8_t)codes[k]; }
        for (int i = 0; i < 8192; ++i) {  // tables as the frames kernel wants them: -x and --clean folded in
            uint8_t a = i < 4096 ? fw[i] : rcw[i - 4096];
            if (P.allow_unknown && a == 0) a = 'X';
            if (P.clean && a == '*') a = 'X';
            tab[16384 + 256 + i] = a;
        }
        {   // pairs of plain-letter codons for k_translate_wide (TranslateParams::pair)
            const uint8_t* baked = tab.data() + 16384 + 256;
            const int iu4[4] = {1, 2, 8, 4};  // IUPAC code of the 2-bit codes A C T G
            auto full = [&](int j) { return (iu4[j & 3] << 8) | (iu4[(j >> 2) & 3] << 4) | iu4[(j >> 4) & 3]; };
            uint8_t* pair = tab.data() + 6 * 4096 + 256;
            for (int i = 0; i < 4096; ++i) {
                const int lo = i & 63, hi = i >> 6;
                pair[2 * i] = baked[full(lo)];
                pair[2 * i + 1] = baked[full(hi)];
                pair[8192 + 2 * i] = baked[4096 + full(hi)];
                pair[8192 + 2 * i + 1] = baked[4096 + full(lo)];
            }
        }
        HIP_TRYX(c, hipMemcpyAsync(c->d_codon, tab.data(), tab.size(), hipMemcpyHostToDevice, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
    }
    P.pair = c->d_codon + 6 * 4096 + 256;
    P.baked = c->d_codon + 16384 + 256;
    P.codon = c->d_codon;
    P.start = c->d_codon + 4096;
    P.codon_rc = c->d_codon + 8192;
    P.start_rc = c->d_codon + 12288;
    P.iupac = c->d_codon + 16384;
    // per-element scratch: nframes elements per record
    const uint64_t ne = c->table.n * (uint64_t)P.nframes;
    const uint64_t saved_n = c->table.n;
    c->table.n = ne;  // size the scratch for elements
    rc = ensure_record_scratch(c);
    c->table.n = saved_n;
    if (rc != BSK_OK) return rc;
    {
        Timed t(c, "k_translate_size+scan", st);
        HIP_TRYX(c, launch_translate_size(d_buf, c->table, tt, P, c->d_out_len, c->d_status, st));
        HIP_TRYX(c, launch_scan_u32(c->d_out_len, c->d_out_off, ne, c->d_scan_tmp, st));
    }
    uint64_t total = 0;
    HIP_TRYX(c, hipMemcpyAsync(&total, c->d_out_off + ne, sizeof total, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    // chromosome-sized records leave the per-record kernels (one wave would translate 10^8 bases alone)
    uint64_t long_max = 0;
    {
        const char* e = c->tune.get("long_bytes");
        const uint32_t thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
        const char* mode = c->tune.get("translate");
        if (!(mode && strcmp(mode, "legacy") == 0)) {
            rc = grow(c, &c->d_long_list, &c->long_list_cap, c->table.n, c->table.n / 8 + 16);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
            HIP_TRYX(c, launch_find_long(c->table.l_seq, c->table.n, thresh, c->d_long_list, c->d_counter + 2, st));
            uint64_t lc[2] = {0, 0};
            HIP_TRYX(c, hipMemcpyAsync(lc, c->d_counter + 2, sizeof lc, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            if (light && lc[0]) {  // chromosome-sized records are translated from positions, nothing validates their layout
                c->translate_light_ok = false;
                return translate_run_device(c, d_buf, n, format, st, out);
            }
            if (lc[0] && lc[0] * (uint64_t)P.nframes <= 65535) {  // (grid.y; more long records than that stay per record)
                P.long_list = c->d_long_list;
                P.long_count = lc[0];
                P.long_thresh = thresh;
                long_max = lc[1];
            }
        }
    }
    {
        // wave per record for long sequences, 16 lanes per record for reads; BSK_TRANSLATE=legacy keeps
        // the per-(record, frame) kernel (used by tests to cross-check the two implementations)
        const char* mode = c->tune.get("translate");
        if (mode && strcmp(mode, "legacy") == 0) {
            HIP_TRYX(c, launch_translate_emit(d_buf, c->table, tt, P, c->d_out_len, c->d_out_off, c->d_out, c->d_status, st));
        } else {
            const uint64_t avg = n / std::max<uint64_t>(1, c->table.n);
            const int forced = (int)c->tune.num("tr_lanes");  // measurement knob
            // one flag byte per record for the records k_translate_wide leaves to k_translate_frames4
            rc = grow(c, &c->d_redo, &c->redo_cap, c->table.n, c->table.n / 8 + 64);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_redo, 0, c->table.n, st));
            if (!c->d_counter) HIP_TRYX(c, hipMalloc((void**)&c->d_counter, 4 * sizeof(uint64_t)));
            HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, sizeof(uint64_t), st));
            // wide kernel: a wave per record from ~3 k bases (a step of 64 lanes covers 3072), 16 lanes per record below
            // (reads: a step of 4 lanes covers 192 bases; 16 lanes per 150-base read left 12 of them idle)
            const int wide_lanes = forced == 4 || forced == 16 || forced == 64 ? forced : (avg >= 3000 ? 64 : (avg < 500 ? 4 : 16));
            uint64_t redo_left = 0;
            {
                Timed t(c, "k_translate", st);
                HIP_TRYX(c, launch_translate_frames(avg >= 1024 ? 64 : 16, d_buf, c->table, tt, P, c->d_out_len, c->d_out_off,
                                                    c->d_out, c->d_status, st, n, c->d_redo, wide_lanes, c->d_counter,
                                                    c->tune.is("translate", "v3") ? 1 : (c->tune.is("translate", "frames4") ? 2 : 0),
                                                    light ? &redo_left : nullptr));
            }
            if (light && redo_left) {
                // a record did not fit the layout the light table assumed (or holds letters beyond ACGT): its l_seq cannot
                // be trusted -- the whole call again with the full index pass; this context stays with it
                c->translate_light_ok = false;
                return translate_run_device(c, d_buf, n, format, st, out);
            }
            HIP_TRYX(c, launch_translate_long(d_buf, c->table, tt, P, c->d_out_len, c->d_out_off, c->d_out, c->d_status,
                                              long_max, st));
        }
    }
    uint64_t status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status & ERR_UNKNOWN_CODON) {
        c->set_error("seq: unknown codon (use flag -x/--allow-unknown-codon to translate it to 'X')");
        return BSK_ERR_FORMAT;
    }
    rc = kernel_error_to_status(c, status);
    if (rc != BSK_OK) return rc;
    out->d_data = c->d_out;
    out->len = total;
    out->records = ne;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// rmdup  (bigseqkit/rmdup.go:70-108 + bigseqkit-lib/rmdup.go)
// ---------------------------------------------------------------------------
void validate_rmdup_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    if (o.b("BySeq") && o.b("ByName"))  // bigseqkit/rmdup.go:79-81
        throw OptError("only one/none of the flags -s (--by-seq) and -n (--by-name) is allowed");
    if (o.b("OnlyPositiveStrand") && !o.b("BySeq"))  // :83-85
        throw OptError("flag -s (--by-seq) needed when using -P (--only-positive-strand)");
}

// RmDupCheck.After (rmdup.go:244-279) with the swapped directory names of Q9 put right: the removed records go to
// <DupSeqsFile>/<executor id>, the duplicate-number lines to <DupNumFile>/<executor id>; nothing is written when no
// record was removed.  The executor id is the device index of the context.
int rmdup_finish(bsk_ctx* c) {
    if (c->side_written) return BSK_OK;
    c->side_written = true;
    if (c->removed == 0) return BSK_OK;
    const Options& o = c->opts;
    auto write = [&](const std::string& dir, const std::string& text) -> int {
        if (dir.empty()) return BSK_OK;
        std::string acc;
        for (size_t i = 0; i <= dir.size(); ++i) {  // os.MkdirAll
            if (i == dir.size() || dir[i] == '/') {
                if (!acc.empty() && mkdir(acc.c_str(), 0777) != 0 && errno != EEXIST) {
                    c->set_error("mkdir " + acc + ": " + strerror(errno));
                    return BSK_ERR_INVALID_ARG;
                }
            }
            if (i < dir.size()) acc.push_back(dir[i]);
        }
        const std::string path = dir + "/" + std::to_string(c->device < 0 ? 0 : c->device);
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) { c->set_error("open " + path + ": " + strerror(errno)); return BSK_ERR_INVALID_ARG; }
        const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
        fclose(f);
        if (!ok) { c->set_error("write " + path + " failed"); return BSK_ERR_INVALID_ARG; }
        return BSK_OK;
    };
    int rc = write(o.s("DupSeqsFile"), c->dup_seqs);
    if (rc != BSK_OK) return rc;
    return write(o.s("DupNumFile"), c->dup_nums);
}

// The records dedupe left in the overflow list (ops_rmdup.hip: same XXH64 key as an earlier record, another second key):
// groups of equal (k1, k2) among them keep their lowest record, exactly as the map of RmDupCheck.Call would
// (rmdup.go:150-199) -- a few records per 10^4 shards, settled on the host.  BSK_ERR_FILTER_FALLBACK: the list did not fit.
static int rmdup_settle_overflow(bsk_ctx* c, uint32_t* d_first, hipStream_t st) {
    uint32_t m = 0;
    HIP_TRYX(c, hipMemcpyAsync(&m, c->d_ovf, sizeof m, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (m == 0) return BSK_OK;
    if ((uint64_t)m + 1 > c->ovf_cap) return BSK_ERR_FILTER_FALLBACK;
    uint64_t* d_kk = nullptr;
    uint32_t* d_patch = nullptr;
    HIP_TRYX(c, hipMalloc((void**)&d_kk, (size_t)m * 16));
    std::vector<uint32_t> idx(m);
    std::vector<uint64_t> kk(2 * (size_t)m);
    int rc = BSK_OK;
    do {
        if (launch_gather_keys(c->d_ovf + 1, m, c->d_keys, c->d_keys2, d_kk, st) != hipSuccess ||
            hipMemcpyAsync(idx.data(), c->d_ovf + 1, (size_t)m * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(kk.data(), d_kk, (size_t)m * 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = BSK_ERR_HIP; break; }
        std::vector<uint32_t> order(m);
        for (uint32_t j = 0; j < m; ++j) order[j] = j;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            if (kk[2 * a] != kk[2 * b]) return kk[2 * a] < kk[2 * b];
            if (kk[2 * a + 1] != kk[2 * b + 1]) return kk[2 * a + 1] < kk[2 * b + 1];
            return idx[a] < idx[b];
        });
        std::vector<uint32_t> pi, pv;  // first[pi] := pv
        for (uint32_t j = 0; j < m;) {
            uint32_t e = j + 1;
            while (e < m && kk[2 * order[e]] == kk[2 * order[j]] && kk[2 * order[e] + 1] == kk[2 * order[j] + 1]) ++e;
            for (uint32_t q = j + 1; q < e; ++q) { pi.push_back(idx[order[q]]); pv.push_back(idx[order[j]]); }
            j = e;
        }
        if (pi.empty()) break;
        const size_t pm = pi.size();
        if (hipMalloc((void**)&d_patch, pm * 8) != hipSuccess ||
            hipMemcpyAsync(d_patch, pi.data(), pm * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_patch + pm, pv.data(), pm * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
            launch_scatter_u32(d_patch, d_patch + pm, (uint32_t)pm, d_first, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = BSK_ERR_HIP; break; }
    } while (false);
    if (d_kk) hipFree(d_kk);
    if (d_patch) hipFree(d_patch);
    if (rc != BSK_OK) c->set_error("libbsk: rmdup: settling the overflow list failed on the device");
    return rc;
}

int rmdup_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    // `-s` on FASTQ: the index pass also hashes (stream_rmdup.hip) and the two keys decide (hash_dev.hpp); everything else
    // (names, IDs, FASTA), BSK_RMDUP=table and BSK_RMDUP_KEYS=off take the separate hash kernel and compare the bytes
    bool by_keys = fastq && o.b("BySeq");
    bool verify_bytes = false;
    uint32_t k1_bits = 64;
    {
        const char* e = c->tune.get("rmdup");
        if (e && strcmp(e, "table") == 0) by_keys = false;
        e = c->tune.get("rmdup_keys");
        if (e && strcmp(e, "off") == 0) by_keys = false;
        if (e && strcmp(e, "verify") == 0) verify_bytes = true;  // keys decide, the bytes of every duplicate are compared on top
        e = c->tune.get("rmdup_k1_bits");                         // tests: keep only the low bits of k1 (forces the overflow list)
        if (e && atoi(e) >= 16 && atoi(e) < 64) k1_bits = (uint32_t)atoi(e);
    }
    int rc;
    if (by_keys) {
        const HashReq hq{o.b("IgnoreCase")};
        rc = build_index_ex(c, d_buf, n, format, st, nullptr, &hq);
    } else {
        rc = build_index(c, d_buf, n, format, st);
    }
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    TextTableH tt;
    rc = prepare_text(c, d_buf, format, st, &tt, /*flatten=*/!fastq && o.b("BySeq"), false, n);  // (see grep: hashed and compared as linear text)
    if (rc != BSK_OK) return rc;
    RmDupParams P;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    P.by_seq = o.b("BySeq");
    P.by_name = o.b("ByName");
    P.ignore_case = o.b("IgnoreCase");
    P.id_mode = id_mode_of(c);
    P.line_width = fastq ? 0 : (int)o.ci("LineWidth");
    P.buf_end = d_buf + n;
    const uint64_t N = c->table.n;
    uint64_t cap = 0;
    uint64_t* tk = nullptr;
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    if (!by_keys) {
        rc = grow(c, &c->d_keys, &c->keys_cap, N, N / 8 + 16);
        if (rc != BSK_OK) return rc;
        if (!fastq && P.by_seq && c->flat_long_count) P.hash_long_min = c->flat_long_thresh;  // (listed by prepare_text just above)
        Timed t(c, "k_rmdup_hash", st);
        HIP_TRYX(c, launch_rmdup_hash(d_buf, n, c->table, tt, P, c->d_keys, nullptr, st));
        HIP_TRYX(c, launch_rmdup_hash_long(d_buf, n, c->table, tt, P, c->d_keys, nullptr, c->d_long_list, c->flat_long_count, st));
    } else if (k1_bits < 64) {
        HIP_TRYX(c, launch_mask_keys(c->d_keys, N, (1ull << k1_bits) - 1ull, st));
    }
    // grouping: radix buckets + one LDS table per bucket (ops_rmdup.hip); BSK_RMDUP=table (and any shard on which a bucket
    // overflows, or with 2^32 records) keeps the one big table in HBM
    uint32_t* d_first = nullptr;
    bool by_buckets = N < (1ull << 32);
    {
        const char* e = c->tune.get("rmdup");
        if (e && strcmp(e, "table") == 0) by_buckets = false;
    }
    if (by_buckets) {
        size_t tmp_bytes = 0;
        HIP_TRYX(c, sort_pairs_bits_temp_bytes(N, 0, 16, &tmp_bytes));
        Arena A;
        const uint64_t o_sk = A.take(N * 8), o_vi = A.take(N * 4), o_vo = A.take(N * 4), o_first = A.take(N * 4),
                       o_bs = A.take((65536 + 2) * 4), o_hist = A.take(65536 * 4), o_tmp = A.take(tmp_bytes + 256);
        rc = arena_reserve(c, &A);
        if (rc != BSK_OK) return rc;
        uint64_t* d_sk = A.at<uint64_t>(o_sk);
        uint32_t* d_vi = A.at<uint32_t>(o_vi);
        uint32_t* d_vo = A.at<uint32_t>(o_vo);
        d_first = A.at<uint32_t>(o_first);
        uint32_t ovf_cap = 0;
        if (by_keys) {
            const uint64_t want = std::max<uint64_t>(4096, N / 16) + 1;
            rc = grow(c, &c->d_ovf, &c->ovf_cap, want, 16);
            if (rc != BSK_OK) return rc;
            ovf_cap = (uint32_t)std::min<uint64_t>(c->ovf_cap - 1, 0xFFFFFFFFull);
            HIP_TRYX(c, hipMemsetAsync(c->d_ovf, 0, sizeof(uint32_t), st));
        }
        {
            Timed t(c, "rmdup_group(sort+dedupe)", st);
            if (!c->tune.is("rmdup_buckets", "hand")) {  // the device radix sort of the pairs (two 8-bit digit passes: 1.5 ms per 79 M pairs)
                HIP_TRYX(c, launch_sort_iota(d_vi, N, st));
                HIP_TRYX(c, launch_sort_iota(d_first, N, st));
                HIP_TRYX(c, launch_sort_pairs_bits(A.at<uint8_t>(o_tmp), tmp_bytes, c->d_keys, d_sk, d_vi, d_vo, N, 0, 16, st));
                HIP_TRYX(c, launch_bucket_dedupe(d_sk, d_vo, N, A.at<uint32_t>(o_bs), d_first, c->d_status, st,
                                                 by_keys ? c->d_keys2 : nullptr, by_keys ? c->d_ovf : nullptr, ovf_cap));
            } else {  // one 16-bit histogram + scatter by hand (ops_rmdup.hip): 6.8 ms -- kept for the comparison
                HIP_TRYX(c, launch_bucket_pass(c->d_keys, N, A.at<uint32_t>(o_hist), A.at<uint32_t>(o_bs), d_first, d_sk, d_vo, st));
                HIP_TRYX(c, launch_bucket_dedupe(d_sk, d_vo, N, A.at<uint32_t>(o_bs), d_first, c->d_status, st,
                                                 by_keys ? c->d_keys2 : nullptr, by_keys ? c->d_ovf : nullptr, ovf_cap, true));
            }
        }
        uint64_t status = 0;
        HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (status & ERR_BUCKET_OVERFLOW) {
            status &= ~(uint64_t)ERR_BUCKET_OVERFLOW;
            HIP_TRYX(c, hipMemcpyAsync(c->d_status, &status, sizeof status, hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            by_buckets = false;
        } else if (by_keys) {
            rc = rmdup_settle_overflow(c, d_first, st);
            if (rc == BSK_ERR_FILTER_FALLBACK) by_buckets = false;  // (the list did not fit: the table path compares bytes)
            else if (rc != BSK_OK) return rc;
            else if (verify_bytes) {
                Timed t(c, "k_rmdup_resolve", st);
                HIP_TRYX(c, launch_rmdup_resolve_first(d_buf, c->table, tt, P, d_first, nullptr, c->d_out_len, c->d_status, nullptr, st));
            } else {
                Timed t(c, "k_rmdup_sizes", st);
                HIP_TRYX(c, launch_rmdup_sizes(c->table, P, d_first, c->d_out_len, st));
            }
        } else {
            Timed t(c, "k_rmdup_resolve", st);
            HIP_TRYX(c, launch_rmdup_resolve_first(d_buf, c->table, tt, P, d_first, nullptr, c->d_out_len, c->d_status, nullptr, st));
        }
    }
    if (!by_buckets) {
        if (by_keys && k1_bits < 64) {
            c->set_error("libbsk: BSK_RMDUP_K1_BITS is a test switch of the key path; the table path needs whole keys");
            return BSK_ERR_INVALID_ARG;
        }
        rc = key_table(c, N, &cap, &tk, st);
        if (rc != BSK_OK) return rc;
        HIP_TRYX(c, launch_rmdup_insert(c->d_keys, N, 0, tk, cap, st));
        HIP_TRYX(c, launch_rmdup_resolve(d_buf, c->table, tt, P, c->d_keys, tk, cap, c->d_out_len, c->d_status, st));
    }
    uint64_t total = 0, kept = 0;
    rc = finish_sizes(c, st, &total, &kept);
    if (rc == BSK_OK) {
        uint64_t status = 0;
        HIP_TRYX(c, hipMemcpy(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost));
        if (status & ERR_HASH_COLLISION) {
            c->set_error("libbsk: two distinct subjects share one 64-bit XXH64 key; refusing to guess (rerun on the CPU path)");
            return BSK_ERR_UNSUPPORTED;
        }
    }
    if (rc != BSK_OK) return rc;
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    SeqParams F = format_params(c, fastq);
    if (!fastq && tt.text_w == c->d_text_w) {  // back to the views for the emit (and the side files)
        rc = prepare_text(c, d_buf, format, st, &tt, false, /*keep_out_len=*/true);
        if (rc != BSK_OK) return rc;
    }
    F.text_w = tt.text_w; F.lin_off = tt.lin_off; F.lin = tt.lin;
    apply_long(c, &F);
    { const int rce = emit_records(c, d_buf, n, F, total, kept, st); if (rce != BSK_OK) return rce; }
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    if (!o.s("DupSeqsFile").empty() || !o.s("DupNumFile").empty()) {
        // side outputs, after the main emit on the same stream (d_out_len / d_out_off are free again)
        c->removed += N - kept;
        c->side_written = false;
        uint8_t* d_has = nullptr;
        uint32_t* d_row_len = nullptr;
        uint64_t* d_row_off = nullptr;
        uint8_t* d_side = nullptr;
        auto cleanup = [&]() {
            for (void* p : {(void*)d_has, (void*)d_row_len, (void*)d_row_off, (void*)d_side}) if (p) hipFree(p);
        };
        int src = BSK_OK;
        do {
            if (hipMalloc((void**)&d_has, N) != hipSuccess || hipMalloc((void**)&d_row_len, N * 4) != hipSuccess ||
                hipMalloc((void**)&d_row_off, (N + 1) * 8) != hipSuccess) { src = BSK_ERR_HIP; break; }
            if (hipMemsetAsync(d_has, 0, N, st) != hipSuccess) { src = BSK_ERR_HIP; break; }
            if (by_buckets) {  // keys[i] := survivor of record i, has_dup[survivor] := 1, from first[] (d_out_len is scratch here)
                if (launch_rmdup_resolve_first(d_buf, c->table, tt, P, d_first, c->d_keys, c->d_out_len, c->d_status, d_has, st) != hipSuccess) { src = BSK_ERR_HIP; break; }
            } else if (launch_rmdup_group(N, c->d_keys, tk, cap, d_has, st) != hipSuccess) { src = BSK_ERR_HIP; break; }
            if (launch_rmdup_side_sizes(d_buf, c->table, P, c->d_keys, d_has, c->d_out_len, d_row_len, st) != hipSuccess) { src = BSK_ERR_HIP; break; }
            if (launch_scan_u32(c->d_out_len, c->d_out_off, N, c->d_scan_tmp, st) != hipSuccess) { src = BSK_ERR_HIP; break; }
            if (launch_scan_u32(d_row_len, d_row_off, N, c->d_scan_tmp, st) != hipSuccess) { src = BSK_ERR_HIP; break; }
            uint64_t dup_total = 0, row_total = 0;
            hipMemcpyAsync(&dup_total, c->d_out_off + N, 8, hipMemcpyDeviceToHost, st);
            hipMemcpyAsync(&row_total, d_row_off + N, 8, hipMemcpyDeviceToHost, st);
            if (hipStreamSynchronize(st) != hipSuccess) { src = BSK_ERR_HIP; break; }
            if (hipMalloc((void**)&d_side, std::max<uint64_t>(1, std::max(dup_total, row_total))) != hipSuccess) { src = BSK_ERR_HIP; break; }
            if (!o.s("DupSeqsFile").empty() && dup_total) {
                SeqParams F2 = F;  // other sizes than the main output: every record goes through the per-record kernel
                F2.long_list = nullptr; F2.long_count = 0;
                if (launch_seq_emit(d_buf, c->table, F2, c->d_out_len, c->d_out_off, d_side, st) != hipSuccess) { src = BSK_ERR_HIP; break; }
                const size_t at = c->dup_seqs.size();
                c->dup_seqs.resize(at + dup_total);
                if (hipMemcpyAsync(&c->dup_seqs[at], d_side, dup_total, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess) { src = BSK_ERR_HIP; break; }
            }
            if (!o.s("DupNumFile").empty() && row_total) {
                if (launch_rmdup_rows(d_buf, c->table, P, c->d_keys, d_row_len, d_row_off, d_side, st) != hipSuccess) { src = BSK_ERR_HIP; break; }
                std::string rows(row_total, '\0');
                if (hipMemcpyAsync(&rows[0], d_side, row_total, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess) { src = BSK_ERR_HIP; break; }
                // rows are in file order: group them by survivor, groups in the order of their survivor
                std::vector<std::pair<uint64_t, std::string>> groups;  // survivor -> "id, id, ..."
                std::unordered_map<uint64_t, size_t> where;
                std::vector<uint32_t> count;
                for (size_t i = 0; i < rows.size();) {
                    const size_t e = rows.find('\n', i);
                    const uint64_t g = strtoull(rows.substr(i, 20).c_str(), nullptr, 10);
                    const std::string id = rows.substr(i + 21, e - i - 21);
                    auto it = where.find(g);
                    if (it == where.end()) { where[g] = groups.size(); groups.emplace_back(g, id); count.push_back(1); }
                    else { groups[it->second].second += ", " + id; ++count[it->second]; }
                    i = e + 1;
                }
                std::vector<size_t> order(groups.size());
                for (size_t k = 0; k < order.size(); ++k) order[k] = k;
                std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return groups[a].first < groups[b].first; });
                for (size_t k : order) c->dup_nums += std::to_string(count[k]) + "\t" + groups[k].second + "\n";
            }
        } while (false);
        cleanup();
        if (src != BSK_OK) { c->set_error("libbsk: rmdup side outputs (-d / -D) failed on the device"); return src; }
    }
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// rmdup across ranks.  One call sequence per rank, the caller runs the collectives in between:
//   keys  -> [all-gather of record counts]      -> pack -> [all-to-all of tuples]
//   resolve (owner side)                        -> [all-to-all of keep bytes, reversed]
//   emit
// The context keeps the record table of the shard between keys and emit.
// ---------------------------------------------------------------------------
static RmDupParams rmdup_params(bsk_ctx* c, bool fastq) {
    const Options& o = c->opts;
    RmDupParams P;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    P.by_seq = o.b("BySeq");
    P.by_name = o.b("ByName");
    P.ignore_case = o.b("IgnoreCase");
    P.id_mode = id_mode_of(c);
    P.line_width = fastq ? 0 : (int)o.ci("LineWidth");
    P.buf_end = c->dist_buf ? c->dist_buf + c->dist_n : nullptr;
    return P;
}

int rmdup_dist_keys(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, uint64_t* n_records) {
    if (!c->opts.s("DupSeqsFile").empty() || !c->opts.s("DupNumFile").empty()) {
        c->set_error("libbsk: -d / -D side files are not available on the multi-GPU rmdup path");
        return BSK_ERR_UNSUPPORTED;
    }
    // `-s` on FASTQ: both keys come out of the index pass (stream_rmdup.hip); every rank computes the same two functions
    // whichever kernel it takes (hash_dev.hpp)
    const bool fused = format == BSK_FORMAT_FASTQ && c->opts.b("BySeq") && !(c->tune.get("rmdup_keys") && strcmp(c->tune.get("rmdup_keys"), "off") == 0);
    const HashReq hq{c->opts.b("IgnoreCase")};
    int rc = fused ? build_index_ex(c, d_buf, n, format, st, nullptr, &hq) : build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    c->dist_buf = d_buf;
    c->dist_n = n;
    c->dist_format = format;
    const uint64_t N = c->table.n;
    *n_records = N;
    uint64_t status = 0;
    if (N == 0) {
        HIP_TRYX(c, hipMemcpy(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost));
        return kernel_error_to_status(c, status);
    }
    TextTableH tt;
    rc = prepare_text(c, d_buf, format, st, &tt);
    if (rc != BSK_OK) return rc;
    rc = grow(c, &c->d_keys, &c->keys_cap, N, N / 8 + 16);
    if (rc != BSK_OK) return rc;
    rc = grow(c, &c->d_keys2, &c->keys2_cap, N, N / 8 + 16);
    if (rc != BSK_OK) return rc;
    if (!fused) HIP_TRYX(c, launch_rmdup_hash(d_buf, n, c->table, tt, rmdup_params(c, format == BSK_FORMAT_FASTQ), c->d_keys, c->d_keys2, st));
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    return kernel_error_to_status(c, status);
}

int rmdup_dist_pack(bsk_ctx* c, uint64_t base, int world, uint64_t* d_send, uint64_t* counts, hipStream_t st) {
    const uint64_t N = c->table.n;
    for (int r = 0; r < world; ++r) counts[r] = 0;
    if (N == 0) return BSK_OK;
    int rc = grow(c, &c->d_scan_tmp, &c->scan_tmp_cap, 64 + 16, 16);  // [0..63] counts, then cursors
    if (rc != BSK_OK) return rc;
    uint64_t* d_counts = c->d_scan_tmp;
    HIP_TRYX(c, hipMemsetAsync(d_counts, 0, 64 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_rmdup_count_owner(c->d_keys, N, (uint32_t)world, d_counts, st));
    HIP_TRYX(c, hipMemcpyAsync(counts, d_counts, world * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    uint64_t cursor[64], acc = 0;
    for (int r = 0; r < world; ++r) { cursor[r] = acc; acc += counts[r]; }
    HIP_TRYX(c, hipMemcpyAsync(d_counts, cursor, world * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HIP_TRYX(c, launch_rmdup_pack(c->d_keys, c->d_keys2, N, base, (uint32_t)world, d_counts, d_send, st));
    HIP_TRYX(c, hipStreamSynchronize(st));  // cursor lives on the host stack
    return BSK_OK;
}

int rmdup_dist_resolve(bsk_ctx* c, const uint64_t* d_tuples, uint64_t m, uint8_t* d_keep, hipStream_t st) {
    if (m == 0) return BSK_OK;
    uint64_t cap = 1024;
    while (cap < 2 * m) cap <<= 1;
    int rc = grow(c, &c->d_own, &c->own_cap, 3 * cap);
    if (rc != BSK_OK) return rc;
    uint64_t *tk = c->d_own, *tf = c->d_own + cap, *t2 = c->d_own + 2 * cap;
    HIP_TRYX(c, hipMemsetAsync(tk, 0, cap * sizeof(uint64_t), st));
    HIP_TRYX(c, hipMemsetAsync(tf, 0xFF, cap * sizeof(uint64_t), st));
    HIP_TRYX(c, hipMemsetAsync(t2, 0, cap * sizeof(uint64_t), st));
    HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 2 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_rmdup_own(d_tuples, m, tk, tf, t2, cap, d_keep, c->d_status, st));
    uint64_t status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status & ERR_HASH_COLLISION) {
        c->set_error("libbsk: two distinct subjects share one 64-bit XXH64 key; refusing to guess");
        return BSK_ERR_UNSUPPORTED;
    }
    return BSK_OK;
}

int rmdup_dist_emit(bsk_ctx* c, const uint64_t* d_send, const uint8_t* d_reply, uint64_t base, hipStream_t st, bsk_out* out) {
    const uint64_t N = c->table.n;
    if (N == 0) return empty_result(c, out);
    const bool fastq = c->dist_format == BSK_FORMAT_FASTQ;
    int rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 2 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_rmdup_apply(c->table, rmdup_params(c, fastq), d_send, d_reply, base, c->d_out_len, st));
    uint64_t total = 0, kept = 0;
    rc = finish_sizes(c, st, &total, &kept);
    if (rc != BSK_OK) return rc;
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    SeqParams F = format_params(c, fastq);
    if (!fastq) { F.text_w = c->table.text_w; F.lin_off = c->d_lin_off; F.lin = c->d_lin; }  // prepared by the keys phase
    apply_long(c, &F);
    HIP_TRYX(c, launch_seq_emit(c->dist_buf, c->table, F, c->d_out_len, c->d_out_off, c->d_out, st, total, kept));
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// multi-line FASTQ (SeqParser.Read accepts it, helper.go:252-269; the streaming kernels read strict 4-line records)
// ---------------------------------------------------------------------------
// true when a COMPLETE record in the sample has its sequence or its quality on more than one line (FASTQ grammar of
// PARITY.md SPLIT-FQ); a sample of strict 4-line records, or one that cannot be judged, gives false
bool fastq_head_multiline(const uint8_t* h, size_t hb) {
    size_t p = 0;
    auto line_end = [&](size_t s) { while (s < hb && h[s] != '\n') ++s; return s; };
    while (p < hb && h[p] == '@') {
        size_t e = line_end(p);
        if (e >= hb) return false;
        size_t cur = e + 1, seqlen = 0, quallen = 0, seqlines = 0, quallines = 0;
        bool isq = false;
        for (;;) {
            if (cur >= hb) return false;  // the sample ends inside this record
            const size_t le = line_end(cur);
            if (le >= hb) return false;
            const size_t k = le - cur;
            if (!isq) {
                if (k > 0 && h[cur] == '+') isq = true;
                else { seqlen += k; ++seqlines; }
            } else {
                quallen += k;
                ++quallines;
            }
            cur = le + 1;
            if (isq && quallines && quallen >= seqlen) break;
            if (isq && quallines && cur < hb && h[cur] == '@') break;
        }
        if (quallen != seqlen) return false;  // malformed either way: the strict path reports it
        if (seqlines != 1 || quallines != 1) return true;
        p = cur;
    }
    return false;
}

namespace {
struct DevFree {
    std::vector<void*> p;
    ~DevFree() { for (void* q : p) if (q) hipFree(q); }
    template <class T> hipError_t alloc(T** out, uint64_t count) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, std::max<uint64_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) p.push_back(q);
        *out = (T*)q;
        return e;
    }
};
}  // namespace

// the shard rewritten as strict 4-line FASTQ into c->d_norm (ops_mlfq.hip).  A rare path: scratch is allocated and
// freed per call (about 0.6 bytes per input byte for text wrapped at 60 columns).
// d_out == null: no text is written -- c->table gets the records of the text AS THEY STAND (start[] only: what the
// operators that print a record's text need, range / head / duplicate) and *n_out the byte behind the last record.
int normalize_multiline_fastq(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st, const uint8_t** d_out, size_t* n_out) {
    const bool starts_only = d_out == nullptr;
    if (d_out) *d_out = d_buf;
    *n_out = n;
    if (starts_only) c->table.n = 0;
    if (n == 0) return BSK_OK;
    DevFree F;
    const uint64_t nb = mlfq_blocks(n);
    uint32_t* d_cnt = nullptr;
    uint64_t *d_base = nullptr, *d_tmp = nullptr;
    HIP_TRYX(c, F.alloc(&d_cnt, nb));
    HIP_TRYX(c, F.alloc(&d_base, nb + 1));
    HIP_TRYX(c, F.alloc(&d_tmp, 2 * ((nb + 2047) / 2048) + 4));
    HIP_TRYX(c, launch_nl_count(d_buf, n, d_cnt, st));
    HIP_TRYX(c, launch_scan_u32(d_cnt, d_base, nb, d_tmp, st));
    uint64_t nnl = 0;
    uint8_t last = 0;
    HIP_TRYX(c, hipMemcpyAsync(&nnl, d_base + nb, sizeof nnl, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&last, d_buf + n - 1, 1, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    const bool ends_nl = last == '\n';
    const uint64_t L64 = nnl + (ends_nl ? 0 : 1);
    if (L64 >= 0xFFFFFFF0ull) { c->set_error("libbsk: multi-line FASTQ with 2^32 or more lines in one shard"); return BSK_ERR_UNSUPPORTED; }
    const uint32_t L = (uint32_t)L64;
    uint64_t* d_ls = nullptr;
    HIP_TRYX(c, F.alloc(&d_ls, L64 + 2));
    HIP_TRYX(c, hipMemsetAsync(d_ls, 0, sizeof(uint64_t), st));
    HIP_TRYX(c, launch_nl_write(d_buf, n, d_base, d_ls, st));
    if (!ends_nl) {
        const uint64_t v = n + 1;
        HIP_TRYX(c, hipMemcpyAsync(d_ls + L, &v, sizeof v, hipMemcpyHostToDevice, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
    }
    MlfqScratch S;
    const uint32_t nch = mlfq_chunks(L);
    uint32_t* d_tb = nullptr;
    HIP_TRYX(c, F.alloc(&S.next, L64));
    HIP_TRYX(c, F.alloc(&S.plus, L64));
    HIP_TRYX(c, F.alloc(&S.slen, L64));
    HIP_TRYX(c, F.alloc(&S.exitp, L64));
    HIP_TRYX(c, F.alloc(&S.is_start, L64));
    HIP_TRYX(c, F.alloc(&S.entry, (uint64_t)nch));
    HIP_TRYX(c, F.alloc(&S.status, 2));
    HIP_TRYX(c, F.alloc(&d_tb, 1));
    HIP_TRYX(c, hipMemsetAsync(S.is_start, 0, L64 * sizeof(uint32_t), st));
    HIP_TRYX(c, hipMemsetAsync(S.entry, 0xFF, (uint64_t)nch * sizeof(uint32_t), st));
    HIP_TRYX(c, hipMemsetAsync(S.status, 0, 2 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_trailing_blank(d_ls, L, d_tb, st));
    uint32_t tb = 0;
    HIP_TRYX(c, hipMemcpyAsync(&tb, d_tb, sizeof tb, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    HIP_TRYX(c, launch_mlfq_resolve(d_buf, d_ls, L, tb, S, st));
    uint64_t *d_rank = nullptr, *d_tmp2 = nullptr;
    HIP_TRYX(c, F.alloc(&d_rank, L64 + 1));
    HIP_TRYX(c, F.alloc(&d_tmp2, 2 * ((L64 + 2047) / 2048) + 4));
    HIP_TRYX(c, launch_scan_u32(S.is_start, d_rank, L64, d_tmp2, st));
    uint64_t nrec = 0, status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&nrec, d_rank + L64, sizeof nrec, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&status, S.status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    auto format_error = [&](uint64_t f) {
        if (f & 1) c->set_error("invalid FASTQ: record does not start with '@' (multi-line FASTQ reader)");
        else if (f & 2) c->set_error("invalid FASTQ: a record has no '+' line or unmatched length of sequence and quality (multi-line FASTQ reader)");
        else c->set_error("libbsk: a FASTQ record of 2^32 bytes or more");
        return (f & 4) ? BSK_ERR_UNSUPPORTED : BSK_ERR_FORMAT;
    };
    if (status) return format_error(status);
    uint32_t *d_rec = nullptr, *d_len = nullptr;
    uint64_t *d_off = nullptr, *d_tmp3 = nullptr;
    HIP_TRYX(c, F.alloc(&d_rec, nrec));
    HIP_TRYX(c, F.alloc(&d_len, nrec));
    HIP_TRYX(c, F.alloc(&d_off, nrec + 1));
    HIP_TRYX(c, F.alloc(&d_tmp3, 2 * ((nrec + 2047) / 2048) + 4));
    HIP_TRYX(c, launch_mlfq_list(d_ls, L, S, d_rank, d_rec, d_len, st));
    if (starts_only) {
        HIP_TRYX(c, hipMemcpyAsync(&status, S.status, sizeof status, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (status) return format_error(status);
        RecordTable& t = c->table;
        if (nrec + 1 > t.cap || !t.start) {
            for (void* q : {(void*)t.start, (void*)t.l_head, (void*)t.l_seq, (void*)t.aux, (void*)t.text_w})
                if (q) HIP_TRYX(c, hipFree(q));
            t = RecordTable();
            const uint64_t cap = nrec + nrec / 8 + 16;
            HIP_TRYX(c, hipMalloc((void**)&t.start, (cap + 1) * sizeof(uint64_t)));
            HIP_TRYX(c, hipMalloc((void**)&t.l_head, cap * sizeof(uint32_t)));
            HIP_TRYX(c, hipMalloc((void**)&t.l_seq, cap * sizeof(uint32_t)));
            HIP_TRYX(c, hipMalloc((void**)&t.aux, cap * sizeof(uint32_t)));
            HIP_TRYX(c, hipMalloc((void**)&t.text_w, cap * sizeof(uint32_t)));
            t.cap = cap;
        }
        t.n = nrec;
        uint64_t end = n;
        if (nrec) {
            HIP_TRYX(c, launch_mlfq_starts(d_ls, S, d_rec, nrec, n, t.start, st));
            HIP_TRYX(c, hipMemcpyAsync(&end, t.start + nrec, sizeof end, hipMemcpyDeviceToHost, st));
        }
        HIP_TRYX(c, hipStreamSynchronize(st));  // the scratch is freed on return
        *n_out = end;
        return BSK_OK;
    }
    uint64_t total = 0;
    if (nrec) {
        HIP_TRYX(c, launch_scan_u32(d_len, d_off, nrec, d_tmp3, st));
        HIP_TRYX(c, hipMemcpyAsync(&total, d_off + nrec, sizeof total, hipMemcpyDeviceToHost, st));
    }
    HIP_TRYX(c, hipMemcpyAsync(&status, S.status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status) return format_error(status);
    int rc = grow(c, &c->d_norm, &c->norm_cap, total + 16, total / 16 + 256);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, launch_mlfq_emit(d_buf, d_ls, S, d_rec, d_off, nrec, c->d_norm, st));
    HIP_TRYX(c, hipStreamSynchronize(st));  // the scratch is freed on return
    *d_out = c->d_norm;
    *n_out = total;
    return BSK_OK;
}

// `seq -n` / `seq -n -i` on FASTQ: the names leave from the streaming pass itself (stream_names.hip) -- per-range slices
// sized from the header density of the shard head, one scan over the ranges, one gather.  BSK_ERR_FILTER_FALLBACK: a
// slice was too small (or the estimate does not fit); the caller takes the record-table path.
static int seq_names_run(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    const int blocks = std::max(1, c->num_cus * names_max_blocks_per_cu(c->use_dpp));
    uint32_t nranges = 0;
    uint64_t chunk = 0;
    int rc = prep_ranges(c, d_buf, n, /*fastq=*/true, blocks, st, &nranges, &chunk);
    if (rc != BSK_OK) return rc;
    const size_t hb = std::min<size_t>(n, 256 * 1024);
    std::vector<uint8_t> head(hb);
    HIP_TRYX(c, hipMemcpyAsync(head.data(), d_buf, hb, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (!c->norm_active && fastq_head_multiline(head.data(), hb)) return BSK_ERR_MULTILINE_FASTQ;
    uint64_t hdr = 0, line = 0, line_start = 0;
    for (size_t i = 0; i < hb; ++i)
        if (head[i] == '\n') { if ((line & 3) == 0) hdr += i - line_start; ++line; line_start = i + 1; }
    if ((line & 3) == 0) hdr += hb - line_start;  // a header cut by the end of the sample
    double ratio = (double)(hdr + 64) / (double)hb;
    if (const char* sc = c->tune.get("names_scale")) ratio *= atof(sc);  // tests: force the overflow -> fallback route
    uint64_t slice_cap = (uint64_t)((double)chunk * ratio * 1.25) + (c->tune.get("names_scale") ? 16 : 4096);
    slice_cap = (slice_cap + 15) & ~(uint64_t)15;
    if (slice_cap >= (1ull << 32) || slice_cap * nranges > (uint64_t)n + (64ull << 20)) return BSK_ERR_FILTER_FALLBACK;
    rc = grow(c, &c->d_slices, &c->slices_cap, slice_cap * nranges, 256);
    if (rc != BSK_OK) return rc;
    rc = grow(c, &c->d_names_aux, &c->names_aux_cap, 2 * ((uint64_t)nranges + 2), 16);
    if (rc != BSK_OK) return rc;
    NamesDev D;
    D.slices = c->d_slices;
    D.slice_cap = slice_cap;
    D.range_bytes = c->d_names_aux;
    D.range_count = c->d_range_count;
    D.status = c->d_status;
    D.only_id = o.b("OnlyId") ? 1 : 0;
    D.id_mode = id_mode_of(c);
    uint64_t* d_count_base = c->d_names_aux + nranges + 2;
    {
        Timed t(c, "k_names", st);
        HIP_TRYX(c, launch_names(c->use_dpp, blocks, d_buf, n, c->d_anchors,
                                 nranges, reinterpret_cast<uint32_t*>(c->d_anchors + (size_t)nranges + 1), D, st));
    }
    HIP_TRYX(c, launch_scan_small(D.range_bytes, c->d_range_base, nranges, st));
    HIP_TRYX(c, launch_scan_small(D.range_count, d_count_base, nranges, st));
    uint64_t total = 0, records = 0, status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&total, c->d_range_base + nranges, sizeof total, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&records, d_count_base + nranges, sizeof records, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status & ERR_CAPACITY) {
        status &= ~(uint64_t)ERR_CAPACITY;
        HIP_TRYX(c, hipMemcpyAsync(c->d_status, &status, sizeof status, hipMemcpyHostToDevice, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (status == 0) return BSK_ERR_FILTER_FALLBACK;
    }
    if (status) return kernel_error_to_status(c, status);
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    if (total) {
        Timed t(c, "k_names_compact", st);
        HIP_TRYX(c, launch_names_compact(D, c->d_range_base, nranges, c->d_out, st));
    }
    c->table.n = 0;  // no record table was built for this shard
    out->d_data = c->d_out;
    out->len = total;
    out->records = records;
    return BSK_OK;
}

int seq_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    out->d_data = nullptr;
    out->len = 0;
    out->records = 0;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    {
        // names only, nothing that needs the sequence (length / quality filters, gap removal, letter validation): the
        // streaming pass writes them (BSK_NAMES=off keeps the record-table path)
        const char* nm = c->tune.get("names");
        const bool explicit_alphabet = !(c->alphabet == AB_NONE || c->alphabet == AB_UNLIMIT);
        if (fastq && n > 0 && o.b("Name") && !o.b("Seq") && !o.b("RemoveGaps") && o.i("MinLen") <= 0 && o.i("MaxLen") <= 0 &&
            !(o.f("MinQual") > 0) && !(o.f("MaxQual") > 0) && !o.b("ValidateSeq") && !explicit_alphabet &&
            (!o.b("OnlyId") || id_mode_of(c) != 2) && !(nm && strcmp(nm, "off") == 0)) {
            const int rcn = seq_names_run(c, d_buf, n, st, out);
            if (rcn != BSK_ERR_FILTER_FALLBACK) return rcn;
        }
    }
    int rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    // ---- per-partition decisions of SeqTransform.Call (seq.go:94-125)
    Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);  // parser.t after the first record
    if (rc != BSK_OK) return rc;
    if (ab == AB_NONE) ab = AB_UNLIMIT;
    SeqParams P;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    bool printName = true, printSeq = true, printQual = fastq;
    if (o.b("Name") && o.b("Seq")) { /* both on; printQual as is */ }
    else if (o.b("Name")) { printSeq = false; printQual = false; }
    else if (o.b("Seq")) { printName = false; printQual = false; }
    else if (o.b("Qual")) {
        if (!fastq && c->table.n > 0) {
            c->set_error("FASTA format has no quality. So do not just use flag -q (--qual)");
            return BSK_ERR_FORMAT;
        }
        printName = false; printSeq = false; printQual = true;
    }
    P.print_name = printName; P.print_seq = printSeq; P.print_qual = printQual;
    P.qual_only = o.b("Qual");
    P.only_id = o.b("OnlyId");
    P.buf_end = d_buf + n;
    P.id_mode = id_mode_of(c);
    P.reverse = o.b("Reverse");
    P.remove_gaps = o.b("RemoveGaps");
    set_bits(P.gap_set, o.s("GapLetters"));
    P.gap_lt64 = 1;
    for (char ch : o.s("GapLetters")) if ((uint8_t)ch >= 64) P.gap_lt64 = 0;
    P.line_width = (fastq || o.b("Seq") || o.b("Qual")) ? 0 : (int)o.ci("LineWidth");
    P.min_len = (int)o.i("MinLen"); P.max_len = (int)o.i("MaxLen");
    P.min_qual = o.f("MinQual"); P.max_qual = o.f("MaxQual");
    P.qual_base = (int)o.i("QualAsciiBase");
    bool validate = o.b("ValidateSeq");
    if (!validate && !(c->alphabet == AB_NONE || c->alphabet == AB_UNLIMIT)) validate = true;  // seq.go:66-72
    const char* letters = alphabet_letters(ab);
    P.validate = validate && letters != nullptr;
    P.validate_len = (int)o.i("ValidateSeqLength");
    if (letters) set_bits(P.valid_set, letters);
    // one byte map for complement -> dna2rna -> rna2dna -> case (seq.go:191-239)
    uint8_t lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)i;
    bool use_lut = false;
    auto apply = [&](const char* from, const char* to) {
        uint8_t m[256];
        for (int i = 0; i < 256; ++i) m[i] = (uint8_t)i;
        for (size_t k = 0; from[k]; ++k) m[(uint8_t)from[k]] = (uint8_t)to[k];
        for (int i = 0; i < 256; ++i) lut[i] = m[lut[i]];
        use_lut = true;
    };
    if (o.b("Complement")) {
        if (ab == AB_DNA || ab == AB_DNAredundant) apply("acgtryswkmbdhvACGTRYSWKMBDHV", "tgcayrswmkvhdbTGCAYRSWMKVHDB");
        else if (ab == AB_RNA || ab == AB_RNAredundant) apply("acguryswkmbdhvACGURYSWKMBDHV", "ugcayrswmkvhdbUGCAYRSWMKVHDB");
    }
    if (o.b("Dna2rna") && !(ab == AB_RNA || ab == AB_RNAredundant)) apply("tT", "uU");
    if (o.b("Rna2dna") && !(ab == AB_DNA || ab == AB_DNAredundant)) apply("uU", "tT");
    if (o.b("LowerCase")) apply("ABCDEFGHIJKLMNOPQRSTUVWXYZ", "abcdefghijklmnopqrstuvwxyz");
    else if (o.b("UpperCase")) apply("abcdefghijklmnopqrstuvwxyz", "ABCDEFGHIJKLMNOPQRSTUVWXYZ");
    P.use_lut = use_lut;
    if (!c->d_lut) HIP_TRYX(c, hipMalloc((void**)&c->d_lut, 256));
    if (!c->d_qual_err) HIP_TRYX(c, hipMalloc((void**)&c->d_qual_err, 256 * sizeof(double)));
    HIP_TRYX(c, hipMemcpyAsync(c->d_lut, lut, 256, hipMemcpyHostToDevice, st));
    double qe[256];
    for (int q = 0; q < 256; ++q) qe[q] = std::pow(10.0, (double)(q - P.qual_base) / -10.0);
    HIP_TRYX(c, hipMemcpyAsync(c->d_qual_err, qe, sizeof qe, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));  // lut / qe live on the host stack
    P.lut = c->d_lut;
    P.qual_err = c->d_qual_err;

    if (c->table.n == 0) {
        uint64_t status = 0;
        HIP_TRYX(c, hipMemcpy(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost));
        return kernel_error_to_status(c, status);
    }
    {  // wrapped FASTA: random access through the text view (with gap removal: for the records without a gap letter)
        TextTableH tt{nullptr, nullptr, nullptr};
        rc = prepare_text(c, d_buf, format, st, &tt);
        if (rc != BSK_OK) return rc;
        P.text_w = tt.text_w; P.lin_off = tt.lin_off; P.lin = tt.lin;
    }
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
