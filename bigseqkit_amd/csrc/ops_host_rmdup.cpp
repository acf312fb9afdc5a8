// Host side of `rmdup` (RmDupPrepare / RmDupCheck, /root/reference/bigseqkit-lib/rmdup.go) and of its multi-GPU phases.
// (split off ops_host.cpp in round 3; shared helpers: ops_host_internal.hpp)  C-ABI in include/bsk.h.
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
#include "ops_rmdup_xcheck.hpp"
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

static_assert(SEG_TILE == 4096, "k_rmdup_place (ops_rmdup.hip, SEG_TILE_BYTES) writes first4k[] for tiles of this size");

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
static int rmdup_settle_overflow(bsk_ctx* c, uint32_t* d_first, hipStream_t st, uint64_t m64, const uint64_t* d_k1 = nullptr,
                                 const uint64_t* d_k2 = nullptr) {
    if (!d_k1) { d_k1 = c->d_keys; d_k2 = c->d_keys2; }
    const uint32_t m = (uint32_t)std::min<uint64_t>(m64, 0xFFFFFFFFull);  // (the list's length: status word [3] of the read-back)
    if (m == 0) return BSK_OK;
    if ((uint64_t)m + 1 > c->ovf_cap) return BSK_ERR_FILTER_FALLBACK;
    uint64_t* d_kk = nullptr;
    uint32_t* d_patch = nullptr;
    HIP_TRYX(c, hipMalloc((void**)&d_kk, (size_t)m * 16));
    std::vector<uint32_t> idx(m);
    std::vector<uint64_t> kk(2 * (size_t)m);
    int rc = BSK_OK;
    do {
        if (launch_gather_keys(c->d_ovf + 1, m, d_k1, d_k2, d_kk, st) != hipSuccess ||
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

static int rmdup_settle_first(bsk_ctx* c, const uint8_t* d_buf, const TextTableH& tt, const RmDupParams& P, uint32_t* d_first, hipStream_t st);

int rmdup_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    // `-s` on FASTQ: the index pass also hashes (stream_rmdup.hip) and the two keys decide (hash_dev.hpp); everything else
    // (names, IDs, FASTA), BSK_RMDUP=table and BSK_RMDUP_KEYS=off take the separate hash kernel and compare the bytes
    // rmdup_keys: (default) "verify" -- the keys group, and the bytes of every duplicate are compared with its survivor's
    // (RmDupCheck's test, rmdup.go:193-199; a difference fails the call: ERR_HASH_COLLISION); "two-key" -- equal (k1, k2)
    // decide alone, 128 bits and no second look at the text (PARITY.md KEYS: round 3's default, ~1.5 ms faster at C5);
    // "off" -- the separate hash kernel and the byte-comparing table path
    bool by_keys = fastq && o.b("BySeq");
    bool verify_bytes = true;
    uint32_t k1_bits = 64, k2_bits = 64;
    {
        const char* e = c->tune.get("rmdup");
        if (e && strcmp(e, "table") == 0) by_keys = false;
        e = c->tune.get("rmdup_keys");
        if (e && strcmp(e, "off") == 0) by_keys = false;
        if (e && (strcmp(e, "two-key") == 0 || strcmp(e, "keys") == 0)) verify_bytes = false;
        e = c->tune.get("rmdup_k1_bits");                         // tests: keep only the low bits of k1 (forces the overflow list)
        if (e && atoi(e) >= 16 && atoi(e) < 64) k1_bits = (uint32_t)atoi(e);
        e = c->tune.get("rmdup_k2_bits");                         // tests: ... and of k2 (different sequences under one PAIR of keys: the exact settlement)
        if (e && atoi(e) >= 1 && atoi(e) < 64) k2_bits = (uint32_t)atoi(e);
    }
    int rc;
    TextTableH tt;
    RmDupParams P;
    uint64_t N = 0, cap = 0, total = 0, kept = 0;
    uint64_t* tk = nullptr;
    uint32_t* d_first = nullptr;
    bool by_buckets = false;
    bool placed = false;  // sizes, comparison, offsets and segment list came from the one pass of k_rmdup_place
    // verify (default): k1 alone groups and the bytes decide -- the pass computes one key, nothing gathers second keys; only a
    // shard on which the comparison finds two different sequences under one k1 goes round again with both keys
    bool with_k2 = !verify_bytes;
    SeqParams F;
    for (;;) {
    if (by_keys) {
        // (k1 alone groups and the bytes decide: the key's VALUE is seen by nothing but the grouping, so it is the chain-free
        // grouping key of hash_dev.hpp, not XXH64 -- `rmdup_hash=xxh64` keeps the reference's function there too)
        HashReq hq{o.b("IgnoreCase"), with_k2};
        hq.group = !with_k2 && !c->tune.is("rmdup_hash", "xxh64");
        rc = build_index_ex(c, d_buf, n, format, st, nullptr, &hq);
    } else {
        rc = build_index(c, d_buf, n, format, st);
    }
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    rc = prepare_text(c, d_buf, format, st, &tt, /*flatten=*/!fastq && o.b("BySeq"), false, n);  // (see grep: hashed and compared as linear text)
    if (rc != BSK_OK) return rc;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    P.by_seq = o.b("BySeq");
    P.by_name = o.b("ByName");
    P.ignore_case = o.b("IgnoreCase");
    P.id_mode = id_mode_of(c);
    P.line_width = fastq ? 0 : (int)o.ci("LineWidth");
    P.buf_end = d_buf + n;
    N = c->table.n;
    cap = 0;
    tk = nullptr;
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    if (!by_keys) {
        rc = grow(c, &c->d_keys, &c->keys_cap, N, N / 8 + 16);
        if (rc != BSK_OK) return rc;
        if (!fastq && P.by_seq && c->flat_long_count) P.hash_long_min = c->flat_long_thresh;  // (listed by prepare_text just above)
        Timed t(c, "k_rmdup_hash", st);
        HIP_TRYX(c, launch_rmdup_hash(d_buf, n, c->table, tt, P, c->d_keys, nullptr, st));
        HIP_TRYX(c, launch_rmdup_hash_long(d_buf, n, c->table, tt, P, c->d_keys, nullptr, c->d_long_list, c->flat_long_count, st));
        if (k1_bits < 64) HIP_TRYX(c, launch_mask_keys(c->d_keys, N, (1ull << k1_bits) - 1ull, st));
    } else {
        if (k1_bits < 64) HIP_TRYX(c, launch_mask_keys(c->d_keys, N, (1ull << k1_bits) - 1ull, st));
        if (with_k2 && k2_bits < 64) HIP_TRYX(c, launch_mask_keys(c->d_keys2, N, (1ull << k2_bits) - 1ull, st));
    }
    // grouping: radix buckets + one LDS table per bucket (ops_rmdup.hip); BSK_RMDUP=table (and any shard on which a bucket
    // overflows, or with 2^32 records) keeps the one big table in HBM
    d_first = nullptr;
    by_buckets = N < (1ull << 32);
    {
        const char* e = c->tune.get("rmdup");
        if (e && strcmp(e, "table") == 0) by_buckets = false;
    }
    if (by_buckets) {
        size_t tmp_bytes = 0;
        HIP_TRYX(c, sort_pairs_bits_iota_temp_bytes(N, 0, (int)RMDUP_BUCKET_BITS, &tmp_bytes));
        Arena A;
        const uint64_t o_sk = A.take(N * 8), o_vo = A.take(N * 4), o_first = A.take(N * 4),
                       o_bs = A.take(((1u << RMDUP_BUCKET_BITS) + 2) * 4), o_hist = A.take((1u << RMDUP_BUCKET_BITS) * 4), o_tmp = A.take(tmp_bytes + 256);
        rc = arena_reserve(c, &A);
        if (rc != BSK_OK) return rc;
        uint64_t* d_sk = A.at<uint64_t>(o_sk);
        uint32_t* d_vo = A.at<uint32_t>(o_vo);
        d_first = A.at<uint32_t>(o_first);
        uint32_t ovf_cap = 0;
        if (by_keys && with_k2) {
            const uint64_t want = std::max<uint64_t>(4096, N / 16) + 1;
            rc = grow(c, &c->d_ovf, &c->ovf_cap, want, 16);
            if (rc != BSK_OK) return rc;
            ovf_cap = (uint32_t)std::min<uint64_t>(c->ovf_cap - 1, 0xFFFFFFFFull);
            HIP_TRYX(c, hipMemsetAsync(c->d_ovf, 0, sizeof(uint32_t), st));
        }
        {
            Timed t(c, "rmdup_group(sort+dedupe)", st);
            if (!c->tune.is("rmdup_buckets", "hand")) {  // the device radix sort of the pairs (two 8-bit digit passes: 1.5 ms per 79 M pairs)
                HIP_TRYX(c, launch_sort_iota(d_first, N, st));
                // (the record numbers travel as a counting iterator: no second iota pass, no array to read)
                HIP_TRYX(c, launch_sort_pairs_bits_iota(A.at<uint8_t>(o_tmp), tmp_bytes, c->d_keys, d_sk, d_vo, N, 0, (int)RMDUP_BUCKET_BITS, st));
                HIP_TRYX(c, launch_bucket_dedupe(d_sk, d_vo, N, A.at<uint32_t>(o_bs), d_first, c->d_status, st,
                                                 by_keys && with_k2 ? c->d_keys2 : nullptr, by_keys && with_k2 ? c->d_ovf : nullptr, ovf_cap));
            } else {  // one 16-bit histogram + scatter by hand (ops_rmdup.hip): 6.8 ms -- kept for the comparison
                HIP_TRYX(c, launch_bucket_pass(c->d_keys, N, A.at<uint32_t>(o_hist), A.at<uint32_t>(o_bs), d_first, d_sk, d_vo, st));
                HIP_TRYX(c, launch_bucket_dedupe(d_sk, d_vo, N, A.at<uint32_t>(o_bs), d_first, c->d_status, st,
                                                 by_keys && with_k2 ? c->d_keys2 : nullptr, by_keys && with_k2 ? c->d_ovf : nullptr, ovf_cap, true));
            }
        }
        rc = ctl_readback(c, st);  // status word + the length of the overflow list: one copy
        if (rc != BSK_OK) return rc;
        uint64_t status = c->status_word();
        if (status & ERR_BUCKET_OVERFLOW) {
            status &= ~(uint64_t)ERR_BUCKET_OVERFLOW;
            HIP_TRYX(c, hipMemcpyAsync(c->d_status, &status, sizeof status, hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            by_buckets = false;
        } else if (by_keys) {
            rc = with_k2 ? rmdup_settle_overflow(c, d_first, st, c->h_ctl[3]) : BSK_OK;  // (k1 alone: no list, the bytes tell)
            if (rc == BSK_ERR_FILTER_FALLBACK) by_buckets = false;  // (the list did not fit: the table path compares bytes)
            else if (rc != BSK_OK) return rc;
            else if (verify_bytes && !c->tune.is("segcopy", "off") && !c->tune.is("rmdup_place", "off")) {
                // round 5: the comparison, the output offsets and the segment list of the copy in ONE pass over the table
                // (k_rmdup_place: decoupled look-back) instead of verify + three scan launches + segment build + first-of-tile
                rc = grow(c, &c->d_seg_src, &c->seg_src_cap, N + 1, N / 8 + 16);
                if (rc != BSK_OK) return rc;
                rc = grow(c, &c->d_seg_first, &c->seg_first_cap, seg_tiles((uint64_t)n + 1) + 1, 64);  // (survivors are copies: no more output than input)
                if (rc != BSK_OK) return rc;
                {
                    const char* e = c->tune.get("long_bytes");
                    c->long_thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
                }
                const uint64_t nb = rmdup_place_blocks(N);
                uint64_t* chain = c->d_scan_tmp;  // [nb] chain words + the ticket (ensure_record_scratch: 3 nb + 6 words)
                HIP_TRYX(c, hipMemsetAsync(chain, 0, (nb + 1) * sizeof(uint64_t), st));
                HIP_TRYX(c, hipMemsetAsync(c->d_fin, 0, 5 * sizeof(uint64_t), st));  // FIN_TOTAL .. FIN_OTHER
                Timed t(c, "k_rmdup_place", st);
                HIP_TRYX(c, launch_rmdup_place(d_buf, n, c->table, P, d_first, chain, reinterpret_cast<uint32_t*>(chain + nb), c->d_out_off,
                                               c->d_seg_src, c->d_seg_first, c->d_fin, c->long_thresh, c->d_status, st));
                placed = true;
            } else if (verify_bytes) {
                Timed t(c, "k_rmdup_verify", st);
                HIP_TRYX(c, launch_rmdup_verify_fastq(d_buf, c->table, P, d_first, c->d_out_len, c->d_status, st));
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
        if (k1_bits < 64) {
            c->set_error("libbsk: BSK_RMDUP_K1_BITS is a test switch of the key path; the table path needs whole keys");
            return BSK_ERR_INVALID_ARG;
        }
        rc = key_table(c, N, &cap, &tk, st);
        if (rc != BSK_OK) return rc;
        HIP_TRYX(c, launch_rmdup_insert(c->d_keys, N, 0, tk, cap, st));
        HIP_TRYX(c, launch_rmdup_resolve(d_buf, c->table, tt, P, c->d_keys, tk, cap, c->d_out_len, c->d_status, st));
    }
    total = kept = 0;
    if (placed && by_buckets) {
        rc = ctl_readback(c, st);
        if (rc != BSK_OK) return rc;
        total = c->fin(bsk_ctx::FIN_TOTAL);
        kept = c->fin(bsk_ctx::FIN_KEPT);
        c->long_count = 0;
        c->long_max = 0;
        rc = kernel_error_to_status(c, c->status_word());
        if (rc == BSK_OK && c->fin(bsk_ctx::FIN_LONG_COUNT) > 0) {
            // a record of a MiB or more: its copy is a block-per-chunk launch that wants the list of such records -- the
            // size pass of the general path writes it (the comparison is done: sizes from first[] alone)
            placed = false;
            HIP_TRYX(c, launch_rmdup_sizes(c->table, P, d_first, c->d_out_len, st));
            rc = finish_sizes(c, st, &total, &kept);
        }
    } else {
        placed = false;
        rc = finish_sizes(c, st, &total, &kept);  // (ERR_HASH_COLLISION comes back as BSK_ERR_UNSUPPORTED: kernel_error_to_status)
    }
    if (rc == BSK_ERR_UNSUPPORTED && by_keys && by_buckets && !with_k2 && (c->last_kernel_flags & ERR_HASH_COLLISION) &&
        !(c->last_kernel_flags & ~(uint64_t)ERR_HASH_COLLISION)) {
        // two different sequences under one XXH64 value (about N^2 / 2^65 per shard): once more with the second key, whose
        // overflow list settles exactly such records; the byte comparison then runs again on what the two keys grouped
        uint64_t zero = 0;
        HIP_TRYX(c, hipMemcpyAsync(c->d_status, &zero, sizeof zero, hipMemcpyHostToDevice, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        c->set_error("");
        with_k2 = true;
        placed = false;
        continue;
    }
    if (rc == BSK_ERR_UNSUPPORTED && by_buckets && (with_k2 || !by_keys) && (c->last_kernel_flags & ERR_HASH_COLLISION) &&
        !(c->last_kernel_flags & ~(uint64_t)ERR_HASH_COLLISION)) {
        // Round 6: two different subjects under one key group even so (both keys equal: about N^2 / 2^129 per shard; the
        // tests mask the keys).  RmDupCheck keys its map by the subject TEXT (rmdup.go:193-211): both survive.  The
        // records that differ from the record their group names are regrouped by text on the host, first[] is put right,
        // and the sizes follow from it without another look at the text.
        uint64_t zero = 0;
        HIP_TRYX(c, hipMemcpyAsync(c->d_status, &zero, sizeof zero, hipMemcpyHostToDevice, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        c->set_error("");
        rc = rmdup_settle_first(c, d_buf, tt, P, d_first, st);
        if (rc != BSK_OK) return rc;
        placed = false;
        HIP_TRYX(c, launch_rmdup_sizes(c->table, P, d_first, c->d_out_len, st));
        rc = finish_sizes(c, st, &total, &kept);
    }
    if (rc != BSK_OK) return rc;
    // Round 6: the survivors ARE text of the shard, in output order -- with the switch "out" = "slices" the result is the
    // list of them (segment i = record i, empty for a duplicate) and nothing is copied: RmDupCheck's result elements are the
    // strings it was handed (rmdup.go:200-222).  Not with side files (-d / -D read the scratch arrays again) and not when
    // some record must be re-formatted ('+' lines that repeat the name: FIN_OTHER).
    if (placed && total && slices_wanted(c) && c->fin(bsk_ctx::FIN_OTHER) == 0 && o.s("DupSeqsFile").empty() && o.s("DupNumFile").empty()) {
        out_as_segments(c, out, c->d_seg_src, c->d_out_off, N, c->d_seg_first, d_buf, d_buf + n, total, kept);
        return BSK_OK;
    }
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    F = format_params(c, fastq);
    if (!fastq && tt.text_w == c->d_text_w) {  // back to the views for the emit (and the side files)
        rc = prepare_text(c, d_buf, format, st, &tt, false, /*keep_out_len=*/true);
        if (rc != BSK_OK) return rc;
    }
    F.text_w = tt.text_w; F.lin_off = tt.lin_off; F.lin = tt.lin;
    apply_long(c, &F);
    if (placed) {
        if (total) {
            Timed tm(c, "k_seg_copy", st);
            HIP_TRYX(c, launch_seg_copy(c->d_seg_src, c->d_out_off, N, c->d_seg_first, c->d_out, total, d_buf, d_buf + n, st));
        }
        if (c->fin(bsk_ctx::FIN_OTHER) != 0) {
            // the few records the copy left out ('+' lines that repeat the name, a last record without '\n'): the record-wise
            // emit writes them, from sizes recomputed for it
            HIP_TRYX(c, launch_rmdup_sizes(c->table, P, d_first, c->d_out_len, st));
            F.seg_src = c->d_seg_src;
            HIP_TRYX(c, launch_seq_emit(d_buf, c->table, F, c->d_out_len, c->d_out_off, c->d_out, st, total, kept));
            F.seg_src = nullptr;
        }
    } else {
        const int rce = emit_records(c, d_buf, n, F, total, kept, st);
        if (rce != BSK_OK) return rce;
    }
    break;
    }
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
    if (const char* e = c->tune.get("rmdup_k1_bits"))  // tests: only the low bits of k1 (different subjects under one k1 at the owner)
        if (atoi(e) >= 16 && atoi(e) < 64) HIP_TRYX(c, launch_mask_keys(c->d_keys, N, (1ull << atoi(e)) - 1ull, st));
    if (const char* e = c->tune.get("rmdup_k2_bits"))  // tests: ... and of k2 (different subjects under one PAIR of keys: the text comparison across ranks)
        if (atoi(e) >= 1 && atoi(e) < 64) HIP_TRYX(c, launch_mask_keys(c->d_keys2, N, (1ull << atoi(e)) - 1ull, st));
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

int rmdup_dist_resolve(bsk_ctx* c, const uint64_t* d_tuples, uint64_t m, uint8_t* d_keep, hipStream_t st, uint64_t* d_surv) {
    if (m == 0) return BSK_OK;
    // Round 4: the received tuples are grouped the way a shard groups its own records -- radix sort by the low key bits, one
    // LDS table per bucket, k2 compared inside, the few k1 collisions settled exactly through the overflow list -- instead
    // of one big table in HBM (3 x 2^28 words zeroed and hit at random: 18.4 ms for the 79 M tuples of a C5 rank; now
    // profiles/r04j_*).  `rmdup=table`, 2^32 tuples or a bucket that overflows keep the table.
    bool by_buckets = m < (1ull << 32) && !c->tune.is("rmdup", "table");
    if (by_buckets) {
        int rc = grow(c, &c->d_own, &c->own_cap, 2 * m);
        if (rc != BSK_OK) return rc;
        uint64_t *k1 = c->d_own, *k2 = c->d_own + m;
        size_t tmp_bytes = 0;
        HIP_TRYX(c, sort_pairs_bits_iota_temp_bytes(m, 0, (int)RMDUP_BUCKET_BITS, &tmp_bytes));
        Arena A;
        const uint64_t o_sk = A.take(m * 8), o_vo = A.take(m * 4), o_first = A.take(m * 4),
                       o_bs = A.take(((1u << RMDUP_BUCKET_BITS) + 2) * 4), o_tmp = A.take(tmp_bytes + 256);
        rc = arena_reserve(c, &A);
        if (rc != BSK_OK) return rc;
        uint32_t* d_first = A.at<uint32_t>(o_first);
        const uint64_t want = std::max<uint64_t>(4096, m / 16) + 1;
        rc = grow(c, &c->d_ovf, &c->ovf_cap, want, 16);
        if (rc != BSK_OK) return rc;
        const uint32_t ovf_cap = (uint32_t)std::min<uint64_t>(c->ovf_cap - 1, 0xFFFFFFFFull);
        HIP_TRYX(c, hipMemsetAsync(c->d_ovf, 0, sizeof(uint32_t), st));
        HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 4 * sizeof(uint64_t), st));
        HIP_TRYX(c, launch_split_tuples(d_tuples, m, k1, k2, st));
        HIP_TRYX(c, launch_sort_iota(d_first, m, st));
        HIP_TRYX(c, launch_sort_pairs_bits_iota(A.at<uint8_t>(o_tmp), tmp_bytes, k1, A.at<uint64_t>(o_sk), A.at<uint32_t>(o_vo), m, 0,
                                                (int)RMDUP_BUCKET_BITS, st));
        HIP_TRYX(c, launch_bucket_dedupe(A.at<uint64_t>(o_sk), A.at<uint32_t>(o_vo), m, A.at<uint32_t>(o_bs), d_first, c->d_status, st, k2,
                                         c->d_ovf, ovf_cap));
        rc = ctl_readback(c, st);
        if (rc != BSK_OK) return rc;
        uint64_t status = c->status_word();
        if (status & ERR_BUCKET_OVERFLOW) {
            by_buckets = false;
        } else {
            rc = rmdup_settle_overflow(c, d_first, st, c->h_ctl[3], k1, k2);
            if (rc == BSK_ERR_FILTER_FALLBACK) by_buckets = false;
            else if (rc != BSK_OK) return rc;
        }
        if (by_buckets) {
            // (the sorted keys are done with: their array is the scratch of the per-group minimum)
            HIP_TRYX(c, launch_keep_lowest(d_tuples, d_first, m, A.at<uint64_t>(o_sk), d_keep, st, d_surv));
            HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 4 * sizeof(uint64_t), st));  // (word 3 counted the list)
            HIP_TRYX(c, hipStreamSynchronize(st));
            return BSK_OK;
        }
        HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 4 * sizeof(uint64_t), st));
    }
    uint64_t cap = 1024;
    while (cap < 2 * m) cap <<= 1;
    int rc = grow(c, &c->d_own, &c->own_cap, 3 * cap);
    if (rc != BSK_OK) return rc;
    uint64_t *tk = c->d_own, *tf = c->d_own + cap, *t2 = c->d_own + 2 * cap;
    HIP_TRYX(c, hipMemsetAsync(tk, 0, cap * sizeof(uint64_t), st));
    HIP_TRYX(c, hipMemsetAsync(tf, 0xFF, cap * sizeof(uint64_t), st));
    HIP_TRYX(c, hipMemsetAsync(t2, 0, cap * sizeof(uint64_t), st));
    HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 2 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_rmdup_own(d_tuples, m, tk, tf, t2, cap, d_keep, c->d_status, st, d_surv));
    uint64_t status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status & ERR_HASH_COLLISION) {
        c->set_error("libbsk: two distinct subjects share one 64-bit XXH64 key; refusing to guess");
        return BSK_ERR_UNSUPPORTED;
    }
    return BSK_OK;
}

int rmdup_dist_emit(bsk_ctx* c, const uint64_t* d_send, const uint8_t* d_reply, uint64_t base, hipStream_t st, bsk_out* out,
                    const uint64_t* d_surv_reply) {
    const uint64_t N = c->table.n;
    if (N == 0) return empty_result(c, out);
    const bool fastq = c->dist_format == BSK_FORMAT_FASTQ;
    int rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 2 * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_rmdup_apply(c->table, rmdup_params(c, fastq), d_send, d_reply, base, c->d_out_len, st));
    const bool xchecked = c->dist_xchecked;  // (bsk_rmdup_dist_xapply compared every pair, local and across ranks)
    c->dist_xchecked = false;
    if (xchecked) {
        // the flagged records that are the first of their TEXT over all ranks survive after all (rmdup_dist_flagged_settle)
        if (c->xres_n) HIP_TRYX(c, launch_x_resurrect(c->table, rmdup_params(c, fastq), c->d_xres, c->xres_n, c->d_out_len, st));
        c->xres_n = 0;
    } else if (d_surv_reply && fastq && c->opts.b("BySeq") && N < (1ull << 32)) {
        // Round 5: a caller that does NOT run the cross-rank check (bsk_rmdup_dist_x*) still
        // has the duplicates whose survivor lives in THIS shard held to RmDupCheck's own test (rmdup.go:193-199): the
        // owner's reply names the survivor's global index, the pairs inside the shard go through the byte comparison of
        // the single-GPU call; a difference fails the call (ERR_HASH_COLLISION).  Pairs that cross ranks then rest on the
        // two keys (PARITY.md KEYS).
        Arena A;
        const uint64_t o_first = A.take(N * 4);
        rc = arena_reserve(c, &A);
        if (rc != BSK_OK) return rc;
        uint64_t* d_nloc = c->d_fin + bsk_ctx::FIN_AUX0;
        HIP_TRYX(c, hipMemsetAsync(d_nloc, 0, sizeof(uint64_t), st));
        HIP_TRYX(c, launch_dist_first(d_send, d_reply, d_surv_reply, N, base, A.at<uint32_t>(o_first), d_nloc, st));
        Timed t(c, "k_rmdup_verify", st);
        HIP_TRYX(c, launch_rmdup_verify_fastq(c->dist_buf, c->table, rmdup_params(c, fastq), A.at<uint32_t>(o_first), nullptr, c->d_status, st));
    }
    uint64_t total = 0, kept = 0;
    rc = finish_sizes(c, st, &total, &kept);
    if (rc != BSK_OK) return rc;
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    SeqParams F = format_params(c, fastq);
    if (!fastq) { F.text_w = c->table.text_w; F.lin_off = c->d_lin_off; F.lin = c->d_lin; }  // prepared by the keys phase
    apply_long(c, &F);
    // (emit_records: the verbatim survivors of a FASTQ shard leave through the segment copy, like the single-GPU call's)
    { const int rce = emit_records(c, c->dist_buf, c->dist_n, F, total, kept, st); if (rce != BSK_OK) return rce; }
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    if (!xchecked) c->dist_local_pairs = d_surv_reply ? c->fin(bsk_ctx::FIN_AUX0) : 0;   // (came back with the size pass's read-back)
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// Round 6: RmDupCheck's text comparison (bigseqkit-lib/rmdup.go:193-211) for EVERY duplicate of the multi-GPU path, also the
// ones whose survivor lives on another rank (7 of 8 at C5 on eight GPUs; until round 5 they rested on the two keys).
//   xpack    : per destination rank the requests of this shard's cross-rank duplicates and their subject text
//              [requests all-to-all, text all-to-all]
//   xcompare : survivor side, one verdict byte per request                [verdicts back the same routes]
//   xapply   : the requests that came back "differs" and the local pairs that differ -> the flagged list
//              [sum of the flagged counts; only when it is not zero: the flagged texts to every rank]
//   flagged_get / flagged_settle : the exact settlement of flagged records on the host -- grouped by TEXT over all ranks,
//              the lowest global index of every text survives: what RmDupCheck's map keyed by the subject decides
// ---------------------------------------------------------------------------
static TextTableH dist_text(bsk_ctx* c, bool fastq) {
    TextTableH tt;
    tt.text_w = fastq ? nullptr : c->table.text_w;
    tt.lin_off = fastq ? nullptr : c->d_lin_off;
    tt.lin = fastq ? nullptr : c->d_lin;
    tt.lin_n = 0;
    return tt;
}

constexpr uint64_t XFLAG_MAX = 1u << 20;  // more flagged records than this: the call fails (keys that collide a million times are no keys)

int rmdup_dist_xpack(bsk_ctx* c, const uint64_t* d_send, const uint8_t* d_reply, const uint64_t* d_surv, uint64_t base,
                     const uint64_t* rank_base, int world, uint64_t* req_cnt, uint64_t* byte_cnt, void** d_req, void** d_text,
                     hipStream_t st) {
    const uint64_t N = c->table.n;
    for (int r = 0; r < world; ++r) req_cnt[r] = byte_cnt[r] = 0;
    c->x_send = d_send; c->x_reply = d_reply; c->x_surv = d_surv; c->x_base = base;
    c->x_m_req = 0;
    c->dist_cross_pairs = 0;
    c->xres_n = 0;
    c->x_flag_host.clear();
    c->x_flag_blob.clear();
    *d_req = nullptr;
    *d_text = nullptr;
    if (N == 0) return BSK_OK;
    XRanks R;
    memset(&R, 0, sizeof R);
    R.world = (uint32_t)world;
    int rank = -1;
    for (int r = 0; r <= world; ++r) R.base[r] = rank_base[r];
    for (int r = 0; r < world; ++r)
        if (rank_base[r] == base && rank_base[r + 1] == base + N) { rank = r; break; }
    if (rank < 0) { c->set_error("libbsk: rmdup: this shard's records are not a range of rank_base[]"); return BSK_ERR_INVALID_ARG; }
    R.rank = (uint32_t)rank;
    const bool fastq = c->dist_format == BSK_FORMAT_FASTQ;
    const TextTableH tt = dist_text(c, fastq);
    const RmDupParams P = rmdup_params(c, fastq);
    int rc = grow(c, &c->d_scan_tmp, &c->scan_tmp_cap, 4 * XCHECK_MAX_WORLD + 16, 16);
    if (rc != BSK_OK) return rc;
    uint64_t* d_cnt = c->d_scan_tmp;              // [64] requests per destination, [64] bytes, then the two cursor arrays
    uint64_t* d_bytes = d_cnt + XCHECK_MAX_WORLD;
    HIP_TRYX(c, hipMemsetAsync(d_cnt, 0, 4 * XCHECK_MAX_WORLD * sizeof(uint64_t), st));
    HIP_TRYX(c, launch_x_count(c->dist_buf, c->table, tt, P, d_send, d_reply, d_surv, N, R, d_cnt, d_bytes, st));
    uint64_t h[2 * XCHECK_MAX_WORLD];
    HIP_TRYX(c, hipMemcpyAsync(h, d_cnt, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    uint64_t m = 0, nb = 0;
    XFrom seg;
    memset(&seg, 0, sizeof seg);
    seg.world = (uint32_t)world;
    uint64_t cur[2 * XCHECK_MAX_WORLD];
    memset(cur, 0, sizeof cur);
    for (int r = 0; r < world; ++r) {
        req_cnt[r] = h[r];
        byte_cnt[r] = h[XCHECK_MAX_WORLD + r];
        cur[r] = m;                 // requests of destination r begin here ...
        seg.req_start[r] = m;
        seg.byte_start[r] = nb;     // ... and their text here
        m += h[r];
        nb += h[XCHECK_MAX_WORLD + r];
    }
    seg.req_start[world] = m;
    seg.byte_start[world] = nb;
    c->x_m_req = m;
    c->dist_cross_pairs = m;
    if (m == 0) return BSK_OK;
    rc = grow(c, &c->d_xreq, &c->xreq_cap, XREQ_WORDS * m, m / 4 + 64);
    if (rc != BSK_OK) return rc;
    rc = grow(c, &c->d_xtext, &c->xtext_cap, nb + 16, nb / 8 + 64);
    if (rc != BSK_OK) return rc;
    uint64_t* d_cur = d_cnt + 2 * XCHECK_MAX_WORLD;
    HIP_TRYX(c, hipMemcpyAsync(d_cur, cur, sizeof cur, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, launch_x_place(c->dist_buf, c->table, tt, P, d_send, d_reply, d_surv, N, R, d_cur, d_cur + XCHECK_MAX_WORLD, c->d_xreq, st));
    HIP_TRYX(c, launch_x_copy(c->dist_buf, c->table, tt, P, c->d_xreq, m, R, seg, c->d_xtext, st));
    HIP_TRYX(c, hipStreamSynchronize(st));  // (cur lives on this stack; the caller hands the buffers to a collective next)
    *d_req = c->d_xreq;
    *d_text = c->d_xtext;
    return BSK_OK;
}

int rmdup_dist_xcompare(bsk_ctx* c, const uint64_t* d_req_in, const uint64_t* req_from, const uint8_t* d_text_in, const uint64_t* bytes_from,
                        int world, uint8_t* d_verdict, hipStream_t st) {
    XFrom from;
    memset(&from, 0, sizeof from);
    from.world = (uint32_t)world;
    uint64_t m = 0, nb = 0;
    for (int r = 0; r < world; ++r) {
        from.req_start[r] = m;
        from.byte_start[r] = nb;
        m += req_from[r];
        nb += bytes_from[r];
    }
    from.req_start[world] = m;
    from.byte_start[world] = nb;
    if (m == 0) return BSK_OK;
    const bool fastq = c->dist_format == BSK_FORMAT_FASTQ;
    HIP_TRYX(c, launch_x_compare(c->dist_buf, c->table, dist_text(c, fastq), rmdup_params(c, fastq), d_req_in, m, from, d_text_in, c->x_base,
                                 d_verdict, c->d_status, st));
    uint64_t status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status & ERR_HASH_COLLISION) {  // (k_x_compare: a request that names no record of this shard, or text outside its segment)
        HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, sizeof(uint64_t), st));
        c->set_error("libbsk: rmdup: a request of the cross-rank comparison names no record of this shard");
        return BSK_ERR_INVALID_ARG;
    }
    return kernel_error_to_status(c, status);
}

// the flagged list (count word + entries) of the running exchange, emptied
static int xflag_reset(bsk_ctx* c, hipStream_t st) {
    int rc = grow(c, &c->d_xflag, &c->xflag_cap, XFLAG_MAX + 1);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, hipMemsetAsync(c->d_xflag, 0, sizeof(uint32_t), st));
    return BSK_OK;
}
// ... brought to the host, ascending (c->x_flag_host)
static int xflag_fetch(bsk_ctx* c, hipStream_t st) {
    uint32_t cnt = 0;
    HIP_TRYX(c, hipMemcpyAsync(&cnt, c->d_xflag, sizeof cnt, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    c->x_flag_host.clear();
    if (cnt == 0) return BSK_OK;
    if (cnt > XFLAG_MAX) {
        c->set_error("libbsk: rmdup: more than 2^20 records differ from the survivor of their key group; refusing (keys that collide that often are no keys)");
        return BSK_ERR_UNSUPPORTED;
    }
    c->x_flag_host.resize(cnt);
    HIP_TRYX(c, hipMemcpy(c->x_flag_host.data(), c->d_xflag + 1, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    std::sort(c->x_flag_host.begin(), c->x_flag_host.end());
    return BSK_OK;
}

int rmdup_dist_xapply(bsk_ctx* c, const uint8_t* d_verdict_back, uint64_t* n_flagged, uint64_t* pairs_compared, hipStream_t st) {
    const uint64_t N = c->table.n;
    *n_flagged = 0;
    if (pairs_compared) *pairs_compared = 0;
    c->dist_xchecked = true;
    c->dist_local_pairs = 0;
    c->xres_n = 0;
    if (N == 0) return BSK_OK;
    const bool fastq = c->dist_format == BSK_FORMAT_FASTQ;
    const TextTableH tt = dist_text(c, fastq);
    const RmDupParams P = rmdup_params(c, fastq);
    int rc = xflag_reset(c, st);
    if (rc != BSK_OK) return rc;
    if (c->x_m_req) HIP_TRYX(c, launch_x_apply(c->d_xreq, d_verdict_back, c->x_m_req, c->d_xflag, (uint32_t)XFLAG_MAX, st));
    // the pairs inside the shard.  `-s` on FASTQ: the comparison kernel of the single-GPU call (4 lanes per duplicate, raises
    // a flag); only when it has found a difference does the listing kernel walk the pairs again.  Every other subject
    // (names, IDs, wrapped FASTA) goes to the listing kernel at once.
    uint64_t* d_nloc = c->d_fin + bsk_ctx::FIN_AUX0;
    HIP_TRYX(c, hipMemsetAsync(d_nloc, 0, sizeof(uint64_t), st));
    bool listed = false;
    if (fastq && c->opts.b("BySeq") && N < (1ull << 32) && !c->tune.is("rmdup_xlocal", "list")) {
        Arena A;
        const uint64_t o_first = A.take(N * 4);
        rc = arena_reserve(c, &A);
        if (rc != BSK_OK) return rc;
        HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, sizeof(uint64_t), st));
        HIP_TRYX(c, launch_dist_first(c->x_send, c->x_reply, c->x_surv, N, c->x_base, A.at<uint32_t>(o_first), d_nloc, st));
        {
            Timed t(c, "k_rmdup_verify", st);
            HIP_TRYX(c, launch_rmdup_verify_fastq(c->dist_buf, c->table, P, A.at<uint32_t>(o_first), nullptr, c->d_status, st));
        }
        rc = ctl_readback(c, st);
        if (rc != BSK_OK) return rc;
        c->dist_local_pairs = c->fin(bsk_ctx::FIN_AUX0);
        const uint64_t status = c->status_word();
        if (status & ERR_HASH_COLLISION) {
            uint64_t rest = status & ~(uint64_t)ERR_HASH_COLLISION;
            HIP_TRYX(c, hipMemcpyAsync(c->d_status, &rest, sizeof rest, hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            if (rest) return kernel_error_to_status(c, rest);
        } else {
            rc = kernel_error_to_status(c, status);
            if (rc != BSK_OK) return rc;
            listed = true;  // (nothing to list)
        }
    }
    if (!listed) {
        HIP_TRYX(c, hipMemsetAsync(d_nloc, 0, sizeof(uint64_t), st));
        HIP_TRYX(c, launch_x_local_list(c->dist_buf, c->table, tt, P, c->x_send, c->x_reply, c->x_surv, N, c->x_base, c->d_xflag,
                                        (uint32_t)XFLAG_MAX, d_nloc, st));
        rc = ctl_readback(c, st);
        if (rc != BSK_OK) return rc;
        c->dist_local_pairs = c->fin(bsk_ctx::FIN_AUX0);
    }
    rc = xflag_fetch(c, st);
    if (rc != BSK_OK) return rc;
    *n_flagged = c->x_flag_host.size();
    if (pairs_compared) *pairs_compared = c->dist_local_pairs + c->x_m_req;
    return BSK_OK;
}

// the flagged records of c->x_flag_host with their subject text: entries {u64 global index, u64 length, bytes padded to 8}
static int xflag_serialise(bsk_ctx* c, const uint8_t* d_buf, const TextTableH& tt, const RmDupParams& P, uint64_t base, hipStream_t st,
                           std::string* blob) {
    blob->clear();
    const uint32_t m = (uint32_t)c->x_flag_host.size();
    if (m == 0) return BSK_OK;
    uint32_t *d_list = nullptr, *d_len = nullptr;
    uint64_t* d_off = nullptr;
    uint8_t* d_txt = nullptr;
    auto cleanup = [&] { for (void* p : {(void*)d_list, (void*)d_len, (void*)d_off, (void*)d_txt}) if (p) hipFree(p); };
    int rc = BSK_OK;
    do {
        if (hipMalloc((void**)&d_list, (size_t)m * 4) != hipSuccess || hipMalloc((void**)&d_len, (size_t)m * 4) != hipSuccess ||
            hipMalloc((void**)&d_off, (size_t)m * 8) != hipSuccess) { rc = BSK_ERR_HIP; break; }
        std::vector<uint32_t> len(m);
        std::vector<uint64_t> off(m);
        if (hipMemcpyAsync(d_list, c->x_flag_host.data(), (size_t)m * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
            launch_x_subject_len(d_buf, c->table, tt, P, d_list, m, d_len, st) != hipSuccess ||
            hipMemcpyAsync(len.data(), d_len, (size_t)m * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = BSK_ERR_HIP; break; }
        uint64_t total = 0;
        for (uint32_t j = 0; j < m; ++j) { off[j] = total; total += len[j]; }
        std::string txt(total, '\0');
        if (total) {
            if (hipMalloc((void**)&d_txt, total) != hipSuccess ||
                hipMemcpyAsync(d_off, off.data(), (size_t)m * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
                launch_x_subject_copy(d_buf, c->table, tt, P, d_list, d_off, m, d_txt, st) != hipSuccess ||
                hipMemcpyAsync(&txt[0], d_txt, total, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { rc = BSK_ERR_HIP; break; }
        }
        blob->reserve(16 * (size_t)m + total + 8 * (size_t)m);
        for (uint32_t j = 0; j < m; ++j) {
            const uint64_t g = base + c->x_flag_host[j], l = len[j];
            blob->append((const char*)&g, 8);
            blob->append((const char*)&l, 8);
            blob->append(txt, off[j], l);
            blob->append((8 - l % 8) % 8, '\0');
        }
    } while (false);
    cleanup();
    if (rc != BSK_OK) c->set_error("libbsk: rmdup: fetching the subjects of the flagged records failed on the device");
    return rc;
}

// groups the entries of `all` by text (-i: case-folded) and returns, per text, the lowest global index
static bool xflag_group(const bsk_ctx* c, const uint8_t* all, size_t n, std::unordered_map<std::string, uint64_t>* lowest,
                        std::vector<std::pair<uint64_t, std::string>>* entries) {
    const bool fold = c->opts.b("IgnoreCase");
    size_t p = 0;
    while (p < n) {
        if (p + 16 > n) return false;
        uint64_t g, l;
        memcpy(&g, all + p, 8);
        memcpy(&l, all + p + 8, 8);
        p += 16;
        const uint64_t padded = l + (8 - l % 8) % 8;
        if (l > n || p + padded > n) return false;
        std::string t((const char*)all + p, (size_t)l);
        p += (size_t)padded;
        if (fold)
            for (auto& ch : t) if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32);  // (bytes.ToLower on ASCII letters, as the kernels fold)
        auto it = lowest->find(t);
        if (it == lowest->end()) lowest->emplace(t, g);
        else if (g < it->second) it->second = g;
        entries->emplace_back(g, std::move(t));
    }
    return true;
}

int rmdup_dist_flagged_get(bsk_ctx* c, void* buf, size_t cap, size_t* need, hipStream_t st) {
    if (c->x_flag_blob.empty() && !c->x_flag_host.empty()) {
        const bool fastq = c->dist_format == BSK_FORMAT_FASTQ;
        const int rc = xflag_serialise(c, c->dist_buf, dist_text(c, fastq), rmdup_params(c, fastq), c->x_base, st, &c->x_flag_blob);
        if (rc != BSK_OK) return rc;
    }
    *need = c->x_flag_blob.size();
    if (buf && cap >= c->x_flag_blob.size() && !c->x_flag_blob.empty()) memcpy(buf, c->x_flag_blob.data(), c->x_flag_blob.size());
    return BSK_OK;
}

int rmdup_dist_flagged_settle(bsk_ctx* c, const void* all, size_t n, hipStream_t st) {
    std::unordered_map<std::string, uint64_t> lowest;
    std::vector<std::pair<uint64_t, std::string>> entries;
    if (!xflag_group(c, (const uint8_t*)all, n, &lowest, &entries)) {
        c->set_error("libbsk: rmdup: the flagged-record lists of the ranks are malformed");
        return BSK_ERR_INVALID_ARG;
    }
    std::vector<uint32_t> res;
    const uint64_t base = c->x_base, N = c->table.n;
    for (auto& e : entries)
        if (e.first >= base && e.first - base < N && lowest[e.second] == e.first) res.push_back((uint32_t)(e.first - base));
    std::sort(res.begin(), res.end());
    res.erase(std::unique(res.begin(), res.end()), res.end());
    c->xres_n = 0;
    if (res.empty()) return BSK_OK;
    int rc = grow(c, &c->d_xres, &c->xres_cap, res.size(), 64);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, hipMemcpyAsync(c->d_xres, res.data(), res.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    c->xres_n = (uint32_t)res.size();
    return BSK_OK;
}

// The single-GPU twin (rmdup_run_device): the key groups are in first[]; records whose text differs from the record their
// group names (two subjects under one pair of keys) are regrouped by text on the host and first[] is put right -- the
// first of every text names itself, the later ones name it.
static int rmdup_settle_first(bsk_ctx* c, const uint8_t* d_buf, const TextTableH& tt, const RmDupParams& P, uint32_t* d_first, hipStream_t st) {
    int rc = xflag_reset(c, st);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, launch_x_first_list(d_buf, c->table, tt, P, d_first, c->d_xflag, (uint32_t)XFLAG_MAX, st));
    rc = xflag_fetch(c, st);
    if (rc != BSK_OK) return rc;
    if (c->x_flag_host.empty()) return BSK_OK;
    std::string blob;
    rc = xflag_serialise(c, d_buf, tt, P, 0, st, &blob);
    if (rc != BSK_OK) return rc;
    std::unordered_map<std::string, uint64_t> lowest;
    std::vector<std::pair<uint64_t, std::string>> entries;
    if (!xflag_group(c, (const uint8_t*)blob.data(), blob.size(), &lowest, &entries)) { c->set_error("libbsk: rmdup: internal list malformed"); return BSK_ERR_INVALID_ARG; }
    std::vector<uint32_t> pi, pv;
    for (auto& e : entries) { pi.push_back((uint32_t)e.first); pv.push_back((uint32_t)lowest[e.second]); }
    uint32_t* d_patch = nullptr;
    const size_t pm = pi.size();
    HIP_TRYX(c, hipMalloc((void**)&d_patch, pm * 8));
    const bool ok = hipMemcpyAsync(d_patch, pi.data(), pm * 4, hipMemcpyHostToDevice, st) == hipSuccess &&
                    hipMemcpyAsync(d_patch + pm, pv.data(), pm * 4, hipMemcpyHostToDevice, st) == hipSuccess &&
                    launch_scatter_u32(d_patch, d_patch + pm, (uint32_t)pm, d_first, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    hipFree(d_patch);
    if (!ok) { c->set_error("libbsk: rmdup: settling the records that differ from their key group failed on the device"); return BSK_ERR_HIP; }
    c->x_flag_host.clear();
    return BSK_OK;
}


}  // namespace bsk
