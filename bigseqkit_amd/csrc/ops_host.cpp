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
        HIP_TRYX(c, sort_pairs_bits_temp_bytes(N, 0, (int)RMDUP_BUCKET_BITS, &tmp_bytes));
        Arena A;  // (used for its offset arithmetic only: the memory is c->d_group)
        const uint64_t o_sk = A.take(N * 8), o_vi = A.take(N * 4), o_vo = A.take(N * 4), o_first = A.take(N * 4),
                       o_bs = A.take(((1u << RMDUP_BUCKET_BITS) + 2) * 4), o_hist = A.take((1u << RMDUP_BUCKET_BITS) * 4), o_tmp = A.take(tmp_bytes + 256);
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
            HIP_TRYX(c, launch_sort_pairs_bits(A.at<uint8_t>(o_tmp), tmp_bytes, c->d_keys, d_sk, d_vi, d_vo, N, 0, (int)RMDUP_BUCKET_BITS, st));
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
int prep_ranges(bsk_ctx* c, const uint8_t* d_buf, size_t n, bool fastq, int blocks, hipStream_t st, uint32_t* nranges_out,
                       uint64_t* chunk_out, uint64_t force_chunk) {
    const uint64_t waves = (uint64_t)blocks * 4;
    uint64_t nr = pick_nranges(n, waves, c->min_range_bytes, (int)c->tune.num("ranges_per_wave"));
    if (force_chunk) nr = std::min<uint64_t>(std::max<uint64_t>(1, (n + force_chunk - 1) / force_chunk), 0x7FFFFFF0ull);
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
    {
        Timed t(c, "k_prep", st);
        HIP_TRYX(c, launch_prep(fastq, d_buf, n, chunk, nranges, c->d_anchors, queue, st, /*line_mode=*/!fastq,
                                /*raw=*/fastq ? nullptr : c->d_anchors + (size_t)nranges + 2));
    }
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
        if (hash) return launch_rmdup_stream(c->use_dpp, hash->fold, hash->mode(), blocks, d_buf, n, anchors, nranges, queue, D, HD, st);
        return launch_index(fastq, c->use_dpp, blocks, d_buf, n, anchors, nranges, queue, D, st, fastq ? 0 : chunk);
    };
    const int per_cu = F ? filter_max_blocks_per_cu(c->use_dpp)
                         : (hash ? rmdup_stream_max_blocks_per_cu(c->use_dpp, hash->fold, hash->mode()) : index_max_blocks_per_cu(fastq, c->use_dpp));
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
        {   // (the call's one head sample: pinned memory, shared with the alphabet guess)
            const int rch = sample_head(c, d_buf, n, st);
            if (rch != BSK_OK) return rch;
        }
        const size_t hb = c->head_len;
        const uint8_t* head = c->h_head;
        if (fastq && !c->norm_active && fastq_head_multiline(head, hb)) return BSK_ERR_MULTILINE_FASTQ;
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
                HD.k2 = hash->k2 ? c->d_keys_sparse + c->sparse.cap : nullptr;
            }
            {
                Timed t(c, F ? "k_filter" : (hash ? "k_rmdup_stream" : "k_index"), st);
                HIP_TRYX(c, launch_pass(blocks, anchors, nranges, queue, D));
            }
            {
                Timed t(c, "k_range_scan", st);
                HIP_TRYX(c, launch_scan_small(c->d_range_count, c->d_range_base, nranges, st, c->d_fin + bsk_ctx::FIN_TABLE_N));
            }
            {   // number of records + status word: one read-back
                const int rcr = ctl_readback(c, st);
                if (rcr != BSK_OK) return rcr;
            }
            total = c->fin(bsk_ctx::FIN_TABLE_N);
            uint64_t status = c->status_word();
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
                    if (rc3 == BSK_OK && hash->k2) rc3 = grow(c, &c->d_keys2, &c->keys2_cap, total, total / 8 + 16);
                    if (rc3 != BSK_OK) return rc3;
                    Timed t(c, "k_rmdup_compact", st);
                    HIP_TRYX(c, launch_rmdup_compact(c->sparse, sparse_cap, c->d_range_count, c->d_range_base, nranges, c->table, HD,
                                                     HashDev{c->d_keys, hash->k2 ? c->d_keys2 : nullptr}, st));
                } else if (total) {
                    Timed t(c, "k_index_compact", st);
                    HIP_TRYX(c, launch_index_compact(c->sparse, sparse_cap, c->d_range_count, c->d_range_base, nranges,
                                                     c->table, st));
                }
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
            if (rc4 == BSK_OK && hash->k2) rc4 = grow(c, &c->d_keys2, &c->keys2_cap, total, total / 8 + 16);
            if (rc4 != BSK_OK) return rc4;
            HD.k1 = c->d_keys;
            HD.k2 = hash->k2 ? c->d_keys2 : nullptr;
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

const char* alphabet_letters(Alphabet a) {
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
bool has_unquoted_comma(const std::string& p) {
    const size_t open = p.rfind('{');
    if (open != std::string::npos && p.find('}', open) == std::string::npos) return true;
    const size_t close = p.find('}');
    return close != std::string::npos && p.find('{') > close;
}
const char* const HELP_UNQUOTED_COMMA =
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

// ---- the control block and the head sample (ctx.hpp): what a call learns from the device, in as few round trips as the
// data dependencies allow.  A stream synchronisation costs 15 - 30 us of idle device, a copy of eight bytes as much as a
// copy of 256: round 3's calls made twenty of them (0.6 ms of a 4.4 ms grep), this round's four.
int ctl_readback(bsk_ctx* c, hipStream_t st) {
    HIP_TRYX(c, hipMemcpyAsync(c->h_ctl, c->d_ctl, bsk_ctx::CTL_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    return BSK_OK;
}

// head (<= 256 KiB) and tail (<= 4 KiB) of the shard of the running call in c->h_head; a second request of the same call
// for the same shard costs nothing
int sample_head(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st) {
    if (c->head_gen == c->call_gen && c->head_of == d_buf && c->head_n == n) return BSK_OK;
    c->head_len = std::min<size_t>(n, bsk_ctx::HEAD_BYTES);
    c->tail_len = std::min<size_t>(n, bsk_ctx::TAIL_BYTES);
    if (c->head_len) HIP_TRYX(c, hipMemcpyAsync(c->h_head, d_buf, c->head_len, hipMemcpyDeviceToHost, st));
    if (c->tail_len) HIP_TRYX(c, hipMemcpyAsync(c->h_head + bsk_ctx::HEAD_BYTES, d_buf + n - c->tail_len, c->tail_len, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    c->head_of = d_buf;
    c->head_n = n;
    c->head_gen = c->call_gen;
    return BSK_OK;
}

Alphabet partition_alphabet(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, int* rc) {
    *rc = BSK_OK;
    if (c->alphabet != AB_NONE) return c->alphabet;
    if (c->alpha_gen == c->call_gen && c->alpha_of == d_buf && c->alpha_format == format) return (Alphabet)c->alpha_value;
    const int64_t thr = c->opts.ci("AlphabetGuessSeqLength");
    size_t want = (size_t)std::max<int64_t>(thr, 10000) * 2 + 65536;
    want = std::min(want, n);
    std::vector<uint8_t> h;
    if (want <= bsk_ctx::HEAD_BYTES) {  // the call's head sample holds it
        *rc = sample_head(c, d_buf, n, st);
        if (*rc != BSK_OK) return AB_UNLIMIT;
        h.assign(c->h_head, c->h_head + want);
    } else {
        h.resize(want);
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
    const Alphabet ab = guess_alphabet_less_conservatively(s.data(), s.size(), thr);
    c->alpha_gen = c->call_gen; c->alpha_of = d_buf; c->alpha_format = format; c->alpha_value = (int)ab;
    return ab;
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
    const uint64_t need = 3 * ((n + 2047) / 2048) + 6;  // block sums, their scan, non-zero counts (launch_scan_u32_fin)
    rc = grow(c, &c->d_scan_tmp, &c->scan_tmp_cap, need, 16);
    if (rc != BSK_OK) return rc;
    return BSK_OK;  // (c->d_counter: part of the control block, init_device)
}

// size array -> scan -> total / kept / kernel status; then the caller emits.  Three launches and ONE read-back (round 3:
// seven launches, four copies): the scan kernels count the records with output and list the records with a very large
// output (written by whole blocks, k_seq_emit<.., LONG>) on the way, and leave every scalar in the control block.
int finish_sizes(bsk_ctx* c, hipStream_t st, uint64_t* total, uint64_t* kept) {
    int rc = grow(c, &c->d_long_list, &c->long_list_cap, c->table.n, c->table.n / 8 + 16);
    if (rc != BSK_OK) return rc;
    {
        const char* e = c->tune.get("long_bytes");
        c->long_thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
    }
    {
        Timed t(c, "k_sizes_scan", st);
        HIP_TRYX(c, launch_scan_u32_fin(c->d_out_len, c->d_out_off, c->table.n, c->d_scan_tmp, c->long_thresh, c->d_long_list, c->d_fin, st));
    }
    rc = ctl_readback(c, st);
    if (rc != BSK_OK) return rc;
    *total = c->fin(bsk_ctx::FIN_TOTAL);
    *kept = c->fin(bsk_ctx::FIN_KEPT);
    c->long_count = c->fin(bsk_ctx::FIN_LONG_COUNT);
    c->long_max = c->fin(bsk_ctx::FIN_LONG_MAX);
    return kernel_error_to_status(c, c->status_word());
}

// records that leave exactly as they stand in the shard (4-line FASTQ, everything printed, nothing rewritten)
static bool records_verbatim(const SeqParams& P) {
    return P.fastq && !P.fasta_out && P.print_name && P.print_seq && P.print_qual && !P.qual_only && !P.only_id && !P.reverse && !P.use_lut &&
           !P.region_on && !P.feat_on && !P.remove_gaps;  // (rename: per record, k_seg_build)
}

// Round 6 (results as ordered slices, include/bsk.h bsk_out.d_seg_*): the records an operator KEEPS, verbatim, are segments of
// the shard -- `seq` with its length / quality filters, `grep`, whatever selects whole FASTQ records.  With out=slices the
// segment list that k_seg_copy would be driven by IS the result (a record that was dropped is a segment of no bytes) and the
// output block is neither allocated nor written.  1: `out` describes the result; 0: not this call (the caller emits as
// before); < 0: error.  Sizes and offsets are the call's d_out_len / d_out_off (finish_sizes).
int try_records_as_slices(bsk_ctx* c, const uint8_t* d_buf, size_t n, const SeqParams& P, uint64_t total, uint64_t kept, hipStream_t st,
                          bsk_out* out) {
    const RecordTable& t = c->table;
    if (!slices_wanted(c) || !records_verbatim(P) || t.n == 0 || total == 0 || c->tune.is("segcopy", "off")) return 0;
    int rc = grow(c, &c->d_seg_src, &c->seg_src_cap, t.n + 1, t.n / 8 + 16);
    if (rc != BSK_OK) return rc > 0 ? -rc : rc;
    rc = grow(c, &c->d_seg_first, &c->seg_first_cap, seg_tiles(total) + 1, 64);
    if (rc != BSK_OK) return rc > 0 ? -rc : rc;
    uint64_t* d_other = c->d_fin + bsk_ctx::FIN_OTHER;
    if (hipMemsetAsync(d_other, 0, sizeof(uint64_t), st) != hipSuccess) { c->set_error("hipMemsetAsync failed"); return -BSK_ERR_HIP; }
    {
        Timed tm(c, "k_seg_prep", st);
        if (launch_seg_build_fastq(d_buf, n, t, c->d_out_len, c->d_seg_src, d_other, st, P.ren_ord) != hipSuccess ||
            launch_seg_first(c->d_out_off, t.n, c->d_seg_first, st) != hipSuccess) { c->set_error("k_seg_prep launch failed"); return -BSK_ERR_HIP; }
    }
    rc = ctl_readback(c, st);
    if (rc != BSK_OK) return rc > 0 ? -rc : rc;
    if (c->fin(bsk_ctx::FIN_OTHER) != 0) return 0;  // (records that are not four plain lines: the emit kernel writes those)
    out_as_segments(c, out, c->d_seg_src, c->d_out_off, t.n, c->d_seg_first, d_buf, d_buf + n, total, kept);
    return 1;
}

// tell the emit kernel which records it must leave to the block-per-chunk launch
int emit_records_at(bsk_ctx* c, const uint8_t* d_buf, size_t n, const SeqParams& Pin, const uint32_t* d_len, const uint64_t* d_off,
                    uint8_t* d_out, uint64_t total, uint64_t kept, hipStream_t st) {
    SeqParams P = Pin;
    P.seg_src = nullptr;
    const RecordTable& t = c->table;
    const char* env = c->tune.get("segcopy");  // off: never; force: whenever the records qualify (tests)
    const bool verbatim = records_verbatim(P);
    bool seg = verbatim && t.n > 0 && total > 0 && ((uintptr_t)d_out & 15u) == 0 && !(env && strcmp(env, "off") == 0);
    if (seg && !(env && strcmp(env, "force") == 0)) seg = kept * 2 >= t.n && total >= (4u << 20);
    if (seg) {
        int rc = grow(c, &c->d_seg_src, &c->seg_src_cap, t.n + 1, t.n / 8 + 16);
        if (rc != BSK_OK) return rc;
        rc = grow(c, &c->d_seg_first, &c->seg_first_cap, seg_tiles(total) + 1, 64);
        if (rc != BSK_OK) return rc;
        uint64_t* d_other = c->d_fin + bsk_ctx::FIN_OTHER;  // (in the control block: comes back with the final read-back)
        HIP_TRYX(c, hipMemsetAsync(d_other, 0, sizeof(uint64_t), st));
        {
            // (round 4 tried to derive the sources inside k_seg_copy from the table instead of building seg_src: the build
            // pass went away, 0.37 ms per 79 M records, and the copy grew by 0.7 ms -- two dependent loads more at the head
            // of every tile; profiles/r04d_*.  The source array stays.)
            Timed tm(c, "k_seg_prep", st);
            HIP_TRYX(c, launch_seg_build_fastq(d_buf, n, t, d_len, c->d_seg_src, d_other, st, P.ren_ord));
            HIP_TRYX(c, launch_seg_first(d_off, t.n, c->d_seg_first, st));
        }
        {
            Timed tm(c, "k_seg_copy", st);
            HIP_TRYX(c, launch_seg_copy(c->d_seg_src, d_off, t.n, c->d_seg_first, d_out, total, d_buf, d_buf + n, st));
        }
        rc = ctl_readback(c, st);
        if (rc != BSK_OK) return rc;
        if (c->fin(bsk_ctx::FIN_OTHER) == 0) return BSK_OK;
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

void out_as_segments(bsk_ctx* c, bsk_out* out, const uint64_t* seg_src, const uint64_t* seg_off, uint64_t nseg, const uint32_t* first4k,
                     const uint8_t* lo, const uint8_t* hi, uint64_t total, uint64_t records) {
    bsk_ctx::PendingOut& P = c->pend_out;
    P = bsk_ctx::PendingOut();
    P.kind = 1;
    P.total = total; P.records = records; P.nseg = nseg;
    P.seg_src = seg_src; P.seg_off = seg_off; P.first4k = first4k;
    P.lo = lo; P.hi = hi;
    P.gen = c->call_gen;
    out->d_data = nullptr;
    out->len = total;
    out->records = records;
    out->d_seg_src = seg_src;
    out->d_seg_off = seg_off;
    out->n_segments = nseg;
}

int out_as_slices(bsk_ctx* c, bsk_out* out, const uint8_t* slices, uint64_t slice_cap, const uint64_t* range_base, uint32_t nranges,
                  uint64_t total, uint64_t records, hipStream_t st) {
    int rc = grow(c, &c->d_slice_src, &c->slice_src_cap, (uint64_t)nranges + 1, 64);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, launch_slice_srcs(slices, slice_cap, nranges, c->d_slice_src, st));
    bsk_ctx::PendingOut& P = c->pend_out;
    P = bsk_ctx::PendingOut();
    P.kind = 2;
    P.total = total; P.records = records; P.nseg = nranges;
    P.seg_src = c->d_slice_src; P.seg_off = range_base; P.first4k = nullptr;
    P.lo = slices; P.hi = slices + slice_cap * (uint64_t)nranges;
    P.slice_cap = slice_cap;
    P.gen = c->call_gen;
    out->d_data = nullptr;
    out->len = total;
    out->records = records;
    out->d_seg_src = P.seg_src;
    out->d_seg_off = P.seg_off;
    out->n_segments = nranges;
    return BSK_OK;
}

int pending_first4k(bsk_ctx* c, hipStream_t st) {
    bsk_ctx::PendingOut& P = c->pend_out;
    if (P.first4k || P.total == 0) return BSK_OK;
    int rc = grow(c, &c->d_seg_first, &c->seg_first_cap, seg_tiles(P.total) + 1, 64);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, launch_seg_first(P.seg_off, P.nseg, c->d_seg_first, st));
    P.first4k = c->d_seg_first;
    return BSK_OK;
}

int materialize_out(bsk_ctx* c, bsk_out* out, hipStream_t st) {
    if (out->n_segments == 0) return BSK_OK;
    bsk_ctx::PendingOut& P = c->pend_out;
    if (P.kind == 0 || out->d_seg_src != P.seg_src || out->d_seg_off != P.seg_off || out->n_segments != P.nseg || out->len != P.total) {
        c->set_error("libbsk: this result is not the one the context holds as slices (a later run replaced it)");
        return BSK_ERR_INVALID_ARG;
    }
    int rc = ensure_out(c, P.total);
    if (rc != BSK_OK) return rc;
    if (P.total) {
        rc = pending_first4k(c, st);
        if (rc != BSK_OK) return rc;
        Timed tm(c, P.kind == 1 ? "k_seg_copy" : "k_slices_compact", st);
        HIP_TRYX(c, launch_seg_copy(P.seg_src, P.seg_off, P.nseg, P.first4k, c->d_out, P.total, P.lo, P.hi, st));
    }
    out->d_data = c->d_out;
    out->d_seg_src = nullptr;
    out->d_seg_off = nullptr;
    out->n_segments = 0;
    P.kind = 0;
    return BSK_OK;
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

void parse_region_opt(const std::string& region, const char* cmd, int* start, int* end) {
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
// multi-line FASTQ (SeqParser.Read accepts it, helper.go:252-269; the streaming kernels read strict 4-line records)
// ---------------------------------------------------------------------------
// true when a COMPLETE record in the sample has its sequence or its quality on more than one line (FASTQ grammar of
// PARITY.md SPLIT-FQ); a sample of strict 4-line records, or one that cannot be judged, gives false
bool fastq_head_multiline(const uint8_t* h, size_t hb) {
    size_t p = 0;
    // (memchr: this walks the whole 256 KiB sample of every call -- byte by byte it cost 0.18 ms of host time per call)
    auto line_end = [&](size_t s) {
        if (s >= hb) return s;
        const void* q = memchr(h + s, '\n', hb - s);
        return q ? (size_t)((const uint8_t*)q - h) : hb;
    };
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

}  // namespace bsk
