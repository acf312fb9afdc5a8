// Internal to the host side of libbsk (ops_host.cpp, ops_host_next.cpp): scratch helpers and the pieces of the
// seq-style "size -> scan -> emit" flow that several operators share.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bsk.h"
#include "ctx.hpp"
#include "ops_host.hpp"
#include "ops_seq.hpp"
#include "ops_translate.hpp"  // TextTableH

namespace bsk {

#define HIP_TRYX(ctx, expr)                                                                  \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            (ctx)->set_error(std::string(#expr) + ": " + hipGetErrorString(e__));           \
            return BSK_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

template <class T>
inline int grow(bsk_ctx* c, T** p, uint64_t* cap, uint64_t need, uint64_t slack = 0) {
    if (need <= *cap && *p) return BSK_OK;
    if (*p) HIP_TRYX(c, hipFree(*p));
    *p = nullptr;
    const uint64_t n = need + slack;
    HIP_TRYX(c, hipMalloc((void**)p, std::max<uint64_t>(n, 1) * sizeof(T)));
    *cap = n;
    return BSK_OK;
}

// Scratch of the global operators (sort, rename, faidx): ONE grow-only allocation per context, carved per call.
// (hipMalloc / hipFree of gigabytes per call cost 110 of the 144 ms of `sort -l` on 25 GB.)
struct Arena {
    uint8_t* base = nullptr;
    uint64_t used = 0;
    uint64_t take(uint64_t bytes) { const uint64_t at = used; used = (used + bytes + 255) & ~255ull; return at; }
    template <class T> T* at(uint64_t off) const { return reinterpret_cast<T*>(base + off); }
};
inline int arena_reserve(bsk_ctx* c, Arena* a) {
    int rc = grow(c, &c->d_arena, &c->arena_cap, a->used, a->used / 8 + 256);
    a->base = c->d_arena;
    return rc;
}

// Open-addressing table of the key-grouping operators (rmdup, rename, pair, common, concat, grep --delete-matched):
// d_keys for N records and `cap` zeroed slots (a power of two >= 2 N) of 16 bytes {key, ~first record index}.
int key_table(bsk_ctx* c, uint64_t N, uint64_t* cap_out, uint64_t** table, hipStream_t st);
struct RmDupParams;
// c->d_keys: XXH64 keys -> first record of every record's group (+ d_has, c->d_out_len); see ops_host.cpp
int group_resolve(bsk_ctx* c, const uint8_t* d_buf, const TextTableH& tt, const RmDupParams& P, uint8_t* d_has, hipStream_t st);
// FASTA text view of the shard's records (text_dev.hpp); null pointers for FASTQ
int prepare_text(bsk_ctx* c, const uint8_t* d_buf, int format, hipStream_t st, TextTableH* tt, bool flatten = false,
                 bool keep_out_len = false, uint64_t buf_n = 0);  // buf_n: bytes in the shard (flatten: bounds its wide loads)
// the control block (status, counters, summary words) in c->h_ctl: one copy + one synchronisation
int ctl_readback(bsk_ctx* c, hipStream_t st);
// head / tail of the call's shard in c->h_head (one copy per call and shard)
int sample_head(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st);
// f(start, end, terminated) for every line of h[0, hb) -- end = position of its '\n', or hb for a last line without one.
// memchr per line: the samples are walked once per call, and a byte loop over 256 KiB costs a quarter of a millisecond
template <class F>
inline void for_lines(const uint8_t* h, size_t hb, F f) {
    size_t p = 0;
    while (p < hb) {
        const void* q = memchr(h + p, '\n', hb - p);
        const size_t e = q ? (size_t)((const uint8_t*)q - h) : hb;
        f(p, e, q != nullptr);
        p = e + 1;
    }
}
// size array -> scan -> total / kept / kernel status (also lists the records with a very large output)
int finish_sizes(bsk_ctx* c, hipStream_t st, uint64_t* total, uint64_t* kept);
void apply_long(const bsk_ctx* c, SeqParams* P);
// the emit step of size -> scan -> emit into c->d_out (sizes in c->d_out_len / c->d_out_off): FASTQ records that leave
// unchanged go through the segmented copy when most records have output, everything else through k_seq_emit
int emit_records(bsk_ctx* c, const uint8_t* d_buf, size_t n, const SeqParams& P, uint64_t total, uint64_t kept, hipStream_t st);
// the same with the caller's size / offset arrays (offsets in RECORD order) and output place
int emit_records_at(bsk_ctx* c, const uint8_t* d_buf, size_t n, const SeqParams& P, const uint32_t* d_len, const uint64_t* d_off,
                    uint8_t* d_out, uint64_t total, uint64_t kept, hipStream_t st);
// out=slices: kept records that leave verbatim as segments of the shard (1: `out` is the result, nothing to emit; 0: emit as
// usual; < 0: -status).  Call after finish_sizes and BEFORE ensure_out: the block is not needed then.
int try_records_as_slices(bsk_ctx* c, const uint8_t* d_buf, size_t n, const SeqParams& P, uint64_t total, uint64_t kept, hipStream_t st,
                          bsk_out* out);
int empty_result(bsk_ctx* c, bsk_out* out);
// Round 6, results as ordered slices (include/bsk.h bsk_out.d_seg_*): does the running call leave its text where it is?
inline bool slices_wanted(const bsk_ctx* c) { return c->out_slices && !c->force_contiguous; }
// ... the text as segments of the shard (kind 1: what launch_seg_copy would move) / as the per-range slices of a streaming
// pass (kind 2: what launch_names_compact would gather; range_base = [nranges + 1] scanned bytes)
void out_as_segments(bsk_ctx* c, bsk_out* out, const uint64_t* seg_src, const uint64_t* seg_off, uint64_t nseg, const uint32_t* first4k,
                     const uint8_t* lo, const uint8_t* hi, uint64_t total, uint64_t records);
int out_as_slices(bsk_ctx* c, bsk_out* out, const uint8_t* slices, uint64_t slice_cap, const uint64_t* range_base, uint32_t nranges,
                  uint64_t total, uint64_t records, hipStream_t st);
// one block in c->d_out after all (out->d_data set, slice fields cleared); no-op on a contiguous result
int materialize_out(bsk_ctx* c, bsk_out* out, hipStream_t st);
// first4k of the pending result (kind 2 builds it on first use)
int pending_first4k(bsk_ctx* c, hipStream_t st);
// SeqParams that print the whole record unchanged == fastx.Record.Format(lineWidth)
SeqParams format_params(bsk_ctx* c, bool fastq);
void set_bits(uint32_t* set, const std::string& letters);
void check_id_regexp(bsk_ctx* c);
int id_mode_of(const bsk_ctx* c);
int id_spans(bsk_ctx* c, const uint8_t* d_buf, hipStream_t st);
// the context's feature set (ctx.features) uploaded and bound to P: name lookup, regions, suffixes, complement map
int bind_features(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, SeqParams* P);
// helpers shared by the per-operator files (ops_host_search / subseq / translate / rmdup / seq .cpp)
bool has_unquoted_comma(const std::string& p);
extern const char* const HELP_UNQUOTED_COMMA;
void parse_region_opt(const std::string& region, const char* cmd, int* start, int* end);  // reRegion + the range checks of Before()
// ranges of the streaming kernels for a shard: anchors in c->d_anchors (k_prep), the work queue behind them
// force_chunk != 0: ranges of that nominal size (a multiple of 16) instead of the number pick_nranges chooses
int prep_ranges(bsk_ctx* c, const uint8_t* d_buf, size_t n, bool fastq, int blocks, hipStream_t st, uint32_t* nranges_out,
                uint64_t* chunk_out, uint64_t force_chunk = 0);
const char* alphabet_letters(Alphabet a);
void complement_table(Alphabet ab, uint8_t m[256]);
// lines of a text file ("\r\n" trimmed, empty lines skipped): pattern files, region files
std::vector<std::string> read_pattern_lines(const std::string& path);

}  // namespace bsk
