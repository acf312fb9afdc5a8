// ============================================================================
// ops_locate.hip -- Locate.Call, exact path (/root/reference/bigseqkit-lib/locate.go:575-767).
// 16 lanes per record test 16 start positions at a time; the '-' strand is searched as the
// reverse-complemented pattern on the forward text (no RevCom(seq) copy, locate.go:673) and
// visited in the order of RevCom coordinates, so rows leave in the reference's order:
// per record, per pattern (CLI order), '+' hits ascending, then '-' hits ascending in the
// reverse-complement frame.  Exact matching makes the "matched" column equal to the pattern.
// Two passes of the same kernel: row bytes per record -> scan -> rows written in place.
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ops_locate.hpp"
#include "pattern_match_dev.hpp"
#include "regex_nfa.hpp"
#include "regex_vm.hpp"
#include "text_dev.hpp"

namespace bsk {

namespace {


__device__ __forceinline__ uint8_t lower8(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

__device__ __forceinline__ uint32_t dec_len(uint64_t v) {
    if ((v >> 32) == 0) {  // a compare chain instead of divisions (64-bit division is emulated)
        const uint32_t x = (uint32_t)v;
        return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) +
               (x >= 10000000u) + (x >= 100000000u) + (x >= 1000000000u);
    }
    uint32_t n = 0;
    do { ++n; v /= 10; } while (v);
    return n;
}
__device__ __forceinline__ uint32_t put_dec(uint8_t* o, uint64_t v) {  // digits straight to their places (no scratch array)
    const uint32_t n = dec_len(v);
    if ((v >> 32) == 0) {
        uint32_t x = (uint32_t)v;
        for (uint32_t k = n; k-- > 0;) { o[k] = (uint8_t)('0' + x % 10u); x /= 10u; }
    } else {
        for (uint32_t k = n; k-- > 0;) { o[k] = (uint8_t)('0' + v % 10); v /= 10; }
    }
    return n;
}
// coordinates of the FM-index branch can be <= 0 on a circular '-' hit (locate.go:330-331 has no +l shift): %d
__device__ __forceinline__ uint32_t dec_len_s(int64_t v) { return v < 0 ? 1u + dec_len((uint64_t)(-v)) : dec_len((uint64_t)v); }
__device__ __forceinline__ uint32_t put_dec_s(uint8_t* o, int64_t v) {
    if (v < 0) { o[0] = '-'; return 1u + put_dec(o + 1, (uint64_t)(-v)); }
    return put_dec(o, (uint64_t)v);
}
// n bytes, any alignment: 16 / 8 / 4 / 2 / 1-byte pieces (a row is written by one lane; byte-wise copies of the ID, the
// name and the pattern made the emit pass 1.8 ms for 0.8 M rows)
__device__ __forceinline__ uint32_t put_bytes(uint8_t* o, const uint8_t* s, uint32_t n) {
    uint32_t k = 0;
    for (; k + 16u <= n; k += 16u) { uint4 v; __builtin_memcpy(&v, s + k, 16); __builtin_memcpy(o + k, &v, 16); }
    if (n & 8u) { uint2 v; __builtin_memcpy(&v, s + k, 8); __builtin_memcpy(o + k, &v, 8); k += 8u; }
    if (n & 4u) { uint32_t v; __builtin_memcpy(&v, s + k, 4); __builtin_memcpy(o + k, &v, 4); k += 4u; }
    if (n & 2u) { uint16_t v; __builtin_memcpy(&v, s + k, 2); __builtin_memcpy(o + k, &v, 2); k += 2u; }
    if (n & 1u) o[k] = s[k];
    return n;
}
__device__ __forceinline__ uint32_t put_str(uint8_t* o, const char* s) {
    uint32_t n = 0;
    while (s[n]) { o[n] = (uint8_t)s[n]; ++n; }
    return n;
}

struct RowCtx {
    const uint8_t* id; uint32_t id_len;
    const uint8_t* name; uint32_t name_len;
    const uint8_t* pat; uint32_t pat_len;
    const uint8_t* disp; uint32_t disp_len;  // pattern column (== pat unless the pattern is a regular expression)
    int format;
    // matched column taken from the text (class patterns): m letters from forward position f, reverse-complemented on '-'
    bool from_text, rc, lower;
    Text T; uint32_t l; uint64_t f;
    const uint8_t* comp;
};

__device__ uint32_t put_matched(uint8_t* o, const RowCtx& r) {
    if (!r.from_text) return put_bytes(o, r.pat, r.pat_len);
    const uint32_t m = r.pat_len;
    for (uint32_t q = 0; q < m; ++q) {
        uint64_t j = r.f + (r.rc ? m - 1 - q : q);
        while (r.l && j >= r.l) j -= r.l;  // (--circular: a match may wrap, and one longer than the sequence more than once)
        uint8_t c = r.T.at((uint32_t)j);
        if (r.rc) c = r.comp[c];
        if (r.lower) c = lower8(c);
        o[q] = c;
    }
    return m;
}

__device__ __forceinline__ uint32_t row_len(const RowCtx& r, int64_t begin, int64_t end) {
    switch (r.format) {
        case 2:  // "%s\tSeqKit\tlocation\t%d\t%d\t0\t%c\t.\tgene_id \"%s\"; \n"
            return r.id_len + 17 + dec_len_s(begin) + 1 + dec_len_s(end) + 3 + 2 + 2 + 9 + r.name_len + 3 + 1;
        case 3:  // "%s\t%d\t%d\t%s\t0\t%c\n"
            return r.id_len + 1 + dec_len_s(begin - 1) + 1 + dec_len_s(end) + 1 + r.name_len + 3 + 1 + 1;
        default: {
            uint32_t n = r.id_len + 1 + r.name_len + 1 + r.disp_len + 1 + 1 + 1 + dec_len_s(begin) + 1 + dec_len_s(end);
            if (r.format == 0) n += 1 + r.pat_len;
            return n + 1;
        }
    }
}

__device__ uint32_t row_put(uint8_t* o, const RowCtx& r, char strand, int64_t begin, int64_t end) {
    uint32_t n = 0;
    n += put_bytes(o + n, r.id, r.id_len);
    if (r.format == 2) {
        n += put_str(o + n, "\tSeqKit\tlocation\t");
        n += put_dec_s(o + n, begin); o[n++] = '\t';
        n += put_dec_s(o + n, end);
        n += put_str(o + n, "\t0\t");
        o[n++] = (uint8_t)strand;
        n += put_str(o + n, "\t.\tgene_id \"");
        n += put_bytes(o + n, r.name, r.name_len);
        n += put_str(o + n, "\"; \n");
    } else if (r.format == 3) {
        o[n++] = '\t';
        n += put_dec_s(o + n, begin - 1); o[n++] = '\t';
        n += put_dec_s(o + n, end); o[n++] = '\t';
        n += put_bytes(o + n, r.name, r.name_len);
        n += put_str(o + n, "\t0\t");
        o[n++] = (uint8_t)strand;
        o[n++] = '\n';
    } else {
        o[n++] = '\t';
        n += put_bytes(o + n, r.name, r.name_len); o[n++] = '\t';
        n += put_bytes(o + n, r.disp, r.disp_len); o[n++] = '\t';
        o[n++] = (uint8_t)strand; o[n++] = '\t';
        n += put_dec_s(o + n, begin); o[n++] = '\t';
        n += put_dec_s(o + n, end);
        if (r.format == 0) { o[n++] = '\t'; n += put_matched(o + n, r); }
        o[n++] = '\n';
    }
    return n;
}

// does pattern pp (length m) occur at forward position f of the (possibly circular) text?
__device__ __forceinline__ bool match_at(const Text& T, uint32_t l, bool fold, const uint8_t* pp, uint32_t m, uint64_t f) {
    for (uint32_t q = 0; q < m; ++q) {
        uint64_t j = f + q;
        if (j >= l) j -= l;  // second copy of a circular text
        uint8_t c = T.at((uint32_t)j);
        if (fold) c = lower8(c);
        if (c != pp[q]) return false;
    }
    return true;
}


__device__ __forceinline__ uint32_t fold_dword(uint32_t x) {  // ASCII lower-case, 4 bytes at once
    const uint32_t ge_a = (x & 0x7F7F7F7Fu) + 0x3F3F3F3Fu;
    const uint32_t ge_z1 = (x & 0x7F7F7F7Fu) + 0x25252525u;
    const uint32_t up = ge_a & ~ge_z1 & ~x & 0x80808080u;
    return x | (up >> 2);
}

// 16 consecutive start positions from a 32-byte window: first min(m,4) bytes as one dword
// (v_alignbyte at static shifts), survivors verified byte by byte -> 16-bit hit mask
__device__ __forceinline__ uint32_t window_hits(const uint8_t* src, const uint8_t* buf_end, bool fold, const uint8_t* pp,
                                                uint32_t m, uint32_t p32, uint32_t pmask, uint32_t valid = 0xFFFFu) {
    uint32_t dw[8];
    if (src + 32 <= buf_end) {
        uint4 a, b2;
        __builtin_memcpy(&a, src, 16);
        __builtin_memcpy(&b2, src + 16, 16);
        dw[0] = a.x; dw[1] = a.y; dw[2] = a.z; dw[3] = a.w;
        dw[4] = b2.x; dw[5] = b2.y; dw[6] = b2.z; dw[7] = b2.w;
    } else {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            uint32_t w = 0;
            for (int q = 0; q < 4; ++q)
                if (src + d * 4 + q < buf_end) w |= (uint32_t)src[d * 4 + q] << (8 * q);
            dw[d] = w;
        }
    }
    if (fold) {
#pragma unroll
        for (int d = 0; d < 8; ++d) dw[d] = fold_dword(dw[d]);
    }
    uint32_t miss = 0;  // mismatch bit per start position: alignbyte, xor, and, min(.,1), lshl_or (no compares)
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int d = b >> 2, sft = b & 3;
        const uint32_t w = sft == 0 ? dw[d] : __builtin_amdgcn_alignbyte(dw[d + 1], dw[d], sft);
        uint32_t t = (w ^ p32) & pmask;
        t = t < 1u ? t : 1u;
        miss |= t << b;
    }
    uint32_t cand = ~miss & valid;  // only start positions that exist are verified
    uint32_t hits = 0;
    while (cand) {
        const uint32_t b = (uint32_t)__ffs((int)cand) - 1u;
        cand &= cand - 1u;
        if (verify_from4(src + b, buf_end, fold, pp, m)) hits |= 1u << b;
    }
    return hits;
}

// GROUP lanes per record (4 for reads: the kernel is bound by dependent-load latency per wave, not by lanes; 16 for
// long sequences)
// LONG (GROUP == 64): one wave per cell (pattern, strand, chunk of start positions) of a chromosome-sized record, see
// LocateParams::long_list; the per-record launches leave those records alone.
template <bool EMIT, bool GEN, int GROUP, bool LONG>
__global__ __launch_bounds__(256) void k_locate(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t, TextTable tt,
                                                LocateParams P, uint32_t* __restrict__ out_len,
                                                const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
                                                uint64_t* __restrict__ rows) {
    const uint64_t slot = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;  // LONG: cell index inside the record
    const uint32_t gl = threadIdx.x % GROUP;
    const uint32_t gshift = (threadIdx.x & 63) / GROUP * GROUP;
    constexpr uint64_t GMASK = GROUP >= 64 ? ~0ull : ((1ull << (GROUP & 63)) - 1ull);
    const int nstr = P.both_strands ? 2 : 1;
    bool live;
    uint64_t g;
    int cell_k = 0, cell_strand = 0;
    uint64_t a_lo = 0, a_hi = ~0ull;  // start positions this group searches (strand frame)
    uint64_t cell_at = 0;             // LONG: index of this cell in cell_bytes / cell_off
    uint64_t cell_rec0 = 0;           // LONG: first cell of this cell's record
    if constexpr (LONG) {
        cell_at = slot;  // one flat grid over the cells of all long records
        live = cell_at < P.long_cells;
        // the record of this cell: last y with cellbase[y] <= cell_at   (wave-uniform)
        uint64_t ylo = 0, yhi = P.long_count;
        while (live && yhi - ylo > 1) {
            const uint64_t mid = (ylo + yhi) >> 1;
            if (P.cellbase[mid] <= cell_at) ylo = mid; else yhi = mid;
        }
        g = live ? P.long_list[ylo] : 0;
        cell_rec0 = live ? P.cellbase[ylo] : 0;
        const uint64_t ln = live ? (P.circular ? 2ull * t.l_seq[g] : (uint64_t)t.l_seq[g]) : 1;
        const uint32_t nch = (uint32_t)((ln + LOCATE_LONG_CH - 1) / LOCATE_LONG_CH);
        const uint64_t cell = live ? cell_at - cell_rec0 : 0;
        const uint32_t per = (uint32_t)(cell / nch), c = (uint32_t)(cell % nch);  // cells are numbered in row order
        if (P.fmi_order) { cell_strand = (int)(per / (uint32_t)P.npat); cell_k = (int)(per % (uint32_t)P.npat); }
        else { cell_k = (int)(per / (uint32_t)nstr); cell_strand = (int)(per % (uint32_t)nstr); }
        a_lo = (uint64_t)c * LOCATE_LONG_CH;
        a_hi = a_lo + LOCATE_LONG_CH;
    } else {
        live = EMIT ? slot < P.nhit : (P.cand ? slot < P.ncand : slot < t.n);
        g = EMIT ? (live ? (uint64_t)P.hit_list[slot] : 0) : (P.cand ? (live ? (uint64_t)P.cand[slot] : 0) : slot);
        if (live && P.long_thresh && t.l_seq[g] >= P.long_thresh) live = false;  // a cell launch handles this record
    }
    const uint64_t gi = live ? g : 0;
    const Text T = text_of(buf, t, tt, gi);
    const uint32_t l = live ? T.L : 0;
    const uint64_t n = P.circular ? 2ull * l : l;  // len(record.Seq.Seq) after the doubling
    // (emit pass: every group of the hit list has rows; groups past its end stay idle but reach the barriers)
    RowCtx R;
    R.id = nullptr;
    R.id_len = 0;
    R.format = P.format;
    R.from_text = GEN;
    R.lower = P.matched_lower != 0;
    R.T = T;
    R.l = l;
    R.comp = P.comp;
    R.rc = false;
    R.f = 0;
    bool have_id = false;  // the ID is only parsed once a record has a hit (group-uniform)
    auto need_id = [&]() {
        if (have_id) return;
        const uint32_t lh = t.l_head[gi];
        const uint8_t* h = buf + t.start[gi] + 1;
        uint32_t off;
        R.id_len = id_span_rec(t, gi, h, lh > 0 ? lh - 1 : 0, P.id_mode, &off, buf + buf_n);
        R.id = h + off;
        have_id = true;
    };
    uint64_t bytes = 0;  // running row bytes of this record (group-uniform)
    uint32_t nrows = 0;
    uint8_t* o = EMIT ? out + out_off[gi] : nullptr;
    if constexpr (LONG && EMIT) { if (live) o += P.cell_off[cell_at] - P.cell_off[cell_rec0]; }
    // reference loop order: per pattern both strands (locate.go:575-767); FM-index branch: per strand all patterns
    const int n_outer = P.fmi_order ? nstr : P.npat, n_inner = P.fmi_order ? P.npat : nstr;
    // (a cell is one iteration of this nest)
    const int lo_0 = LONG ? (P.fmi_order ? cell_strand : cell_k) : 0, li_0 = LONG ? (P.fmi_order ? cell_k : cell_strand) : 0;
    for (int lo_ = lo_0; lo_ < (LONG ? lo_0 + 1 : n_outer); ++lo_)
    for (int li_ = li_0; li_ < (LONG ? li_0 + 1 : n_inner); ++li_) {
        const int k = P.fmi_order ? li_ : lo_;
        const int strand = P.fmi_order ? lo_ : li_;
        R.name = P.name + P.name_off[k];
        R.name_len = P.name_off[k + 1] - P.name_off[k];
        R.pat = P.pat + P.pat_off[k];
        R.pat_len = P.pat_off[k + 1] - P.pat_off[k];
        R.disp = P.disp ? P.disp + P.disp_off[k] : R.pat;
        R.disp_len = P.disp ? P.disp_off[k + 1] - P.disp_off[k] : R.pat_len;
        R.rc = strand != 0;
        const uint32_t m = R.pat_len;
        {
            const uint8_t* pp = P.pat + P.pat_off[strand * P.npat + k];
            const uint32_t* cls = GEN ? P.cls + (uint64_t)P.pat_off[strand * P.npat + k] * 8u : nullptr;
            auto matches = [&](uint64_t f) -> bool {
                if (GEN) return class_match_at(T, l, cls, m, f, P.max_mm);
                return match_at(T, l, P.ignore_case, pp, m, f);
            };
            // '-' coordinates in the original frame (locate.go:698-703); the FM-index branch has no +l shift (:330-331)
            auto coords = [&](uint64_t a, int64_t* begin, int64_t* end) {
                if (strand == 0) { *begin = (int64_t)a + 1; *end = (int64_t)(a + m); }
                else {
                    *begin = (int64_t)l - (int64_t)a - (int64_t)m + 1; *end = (int64_t)l - (int64_t)a;
                    if (!P.fmi_order && a + m > l) { *begin += l; *end += l; }
                }
            };
            if (!live || m == 0 || m > n) continue;
            // candidate start positions a (in the strand's own frame): a + m <= n, and a < l when circular
            uint64_t npos = n - m + 1;
            if (P.circular && npos > l) npos = l;
            const char sc = strand ? '-' : '+';
            // contiguous text inside the shard, plain greedy search: 16 positions per lane and step
            const bool in_buf = T.p >= buf && T.p < buf + buf_n;
            const bool in_lin = tt.lin_n != 0 && T.p >= tt.lin && T.p < tt.lin + tt.lin_n;  // linear copy of a wrapped record
            const bool fastp = !GEN && !P.non_greedy && !P.circular && T.W == 0 && (in_buf || in_lin);
            if (fastp) {
                const uint8_t* const buf_end = in_buf ? buf + buf_n : tt.lin + tt.lin_n;
                const uint32_t npos32 = (uint32_t)npos, n32 = (uint32_t)n;  // not circular: n == l < 2^32
                uint32_t p32 = 0;  // first min(m, 4) pattern bytes, loaded once per (record, pattern, strand)
                for (uint32_t q = 0; q < m && q < 4; ++q) p32 |= (uint32_t)pp[q] << (8 * q);
                const uint32_t pmask = m >= 4 ? 0xFFFFFFFFu : ((1u << (8 * m)) - 1u);
                const uint32_t aend32 = a_hi < npos32 ? (uint32_t)a_hi : npos32;
                for (uint32_t a0 = a_lo < npos32 ? (uint32_t)a_lo : npos32; a0 < aend32; a0 += GROUP * 16) {
                    const uint32_t ib = a0 + gl * 16u;  // first position of this lane, strand frame
                    uint32_t hits = 0;                           // bit k <-> position ib + k (ascending)
                    if (ib < aend32) {
                        const uint32_t left0 = aend32 - ib;
                        const uint32_t valid = left0 < 16u ? (1u << left0) - 1u : 0xFFFFu;  // positions ib .. ib + 15 that exist
                        if (strand == 0) {
                            hits = window_hits(T.p + ib, buf_end, P.ignore_case, pp, m, p32, pmask, valid);
                        } else if (ib + 15u + m <= n32) {
                            // forward window [n-m-ib-15, n-m-ib]: bit b is position ib + 15 - b (all 16 exist here)
                            const uint32_t h = window_hits(T.p + (n32 - m - ib - 15u), buf_end, P.ignore_case, pp, m, p32, pmask);
                            hits = __brev(h) >> 16;
                        } else {
                            // the last positions of the '-' frame are the first D + 1 forward positions: bit b of the
                            // window at forward position 0 is position ib + D - b   (D = n - m - ib < 15)
                            const uint32_t D = n32 - m - ib;
                            const uint32_t h = window_hits(T.p, buf_end, P.ignore_case, pp, m, p32, pmask, (2u << D) - 1u);
                            hits = __brev(h) >> (31u - D);
                        }
                        const uint32_t left = aend32 - ib;
                        if (left < 16u) hits &= (1u << left) - 1u;
                    }
                    const uint64_t gmask = (__ballot(hits != 0) >> gshift) & GMASK;
                    if (gmask == 0) continue;
                    need_id();
                    uint32_t mine = 0, cnt = 0;
                    for (uint32_t hm = hits; hm; hm &= hm - 1u) {
                        const uint64_t a = ib + ((uint32_t)__ffs((int)hm) - 1u);
                        int64_t begin, end;
                        coords(a, &begin, &end);
                        mine += row_len(R, begin, end);
                        ++cnt;
                    }
                    uint32_t incl = mine, icnt = cnt;
#pragma unroll
                    for (int d = 1; d < GROUP; d <<= 1) {
                        const uint32_t v = (uint32_t)__shfl_up((int)incl, d, GROUP);
                        const uint32_t vc = (uint32_t)__shfl_up((int)icnt, d, GROUP);
                        if ((int)gl >= d) { incl += v; icnt += vc; }
                    }
                    const uint32_t tot = (uint32_t)__shfl((int)incl, GROUP - 1, GROUP);
                    const uint32_t totc = (uint32_t)__shfl((int)icnt, GROUP - 1, GROUP);
                    if (EMIT && hits) {
                        uint64_t at = bytes + (incl - mine);
                        for (uint32_t hm = hits; hm; hm &= hm - 1u) {
                            const uint64_t a = ib + ((uint32_t)__ffs((int)hm) - 1u);
                            int64_t begin, end;
                            coords(a, &begin, &end);
                            at += row_put(o + at, R, sc, begin, end);
                        }
                    }
                    bytes += tot;
                    nrows += totc;
                }
            } else if (!P.non_greedy) {
                const uint64_t aend = a_hi < npos ? a_hi : npos;
                for (uint64_t a0 = a_lo < npos ? a_lo : npos; a0 < aend; a0 += GROUP) {
                    const uint64_t a = a0 + gl;
                    bool hit = false;
                    int64_t begin = 0, end = 0;
                    if (a < aend) {
                        const uint64_t f = strand ? n - a - m : a;  // forward position of the occurrence
                        hit = matches(f);
                        coords(a, &begin, &end);
                        R.f = f;
                    }
                    const uint64_t mask = (__ballot(hit) >> gshift) & GMASK;
                    if (mask == 0) continue;
                    need_id();
                    uint32_t mine = hit ? row_len(R, begin, end) : 0u;
                    // exclusive prefix of row sizes inside the group
                    uint32_t incl = mine;
#pragma unroll
                    for (int d = 1; d < GROUP; d <<= 1) {
                        const uint32_t v = (uint32_t)__shfl_up((int)incl, d, GROUP);
                        if ((int)gl >= d) incl += v;
                    }
                    const uint32_t tot = (uint32_t)__shfl((int)incl, GROUP - 1, GROUP);
                    if (EMIT && hit) row_put(o + bytes + (incl - mine), R, sc, begin, end);
                    bytes += tot;
                    nrows += (uint32_t)__popcll(mask);
                }
            } else if (gl == 0) {
                need_id();
                // --non-greedy (locate.go:659-663): the search resumes one base AFTER the match end
                uint64_t a = 0;
                while (a < npos) {
                    const uint64_t f = strand ? n - a - m : a;
                    if (matches(f)) {
                        int64_t begin, end;
                        coords(a, &begin, &end);
                        R.f = f;
                        if (EMIT) row_put(o + bytes, R, sc, begin, end);
                        bytes += row_len(R, begin, end);
                        ++nrows;
                        a += (uint64_t)m + 1;
                    } else {
                        ++a;
                    }
                }
            }
        }
    }
    if (P.non_greedy) {  // lane 0 did the work
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)bytes, 0, GROUP);
        const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(bytes >> 32), 0, GROUP);
        bytes = ((uint64_t)hi << 32) | lo;
        nrows = (uint32_t)__shfl((int)nrows, 0, GROUP);
    }
    if (EMIT) {
        // rows of this block -> one global atomic per block
        __shared__ unsigned int s_rows;
        if (threadIdx.x == 0) s_rows = 0;
        __syncthreads();
        if (live && gl == 0 && nrows) atomicAdd(&s_rows, nrows);
        __syncthreads();
        if (threadIdx.x == 0 && s_rows) atomicAdd((unsigned long long*)rows, (unsigned long long)s_rows);
    }
    if constexpr (LONG && !EMIT) {
        if (live && gl == 0) P.cell_bytes[cell_at] = (uint32_t)bytes;
        return;
    }
    if (live && gl == 0) {
        if (!EMIT) {
            // no global atomics here: 2 % of 39 M records hitting one counter cost more than the whole search
            // (25 ms vs 12 ms at C3); the hit list is built by k_compact_hits, the rows are counted by the emit pass
            out_len[g] = (uint32_t)bytes;  // rows of one record beyond 4 GiB are not representable
        }
    }
}

// ---------------------------------------------------------------------------
// locate -r with matches of VARIABLE length (bigseqkit-lib/locate.go:583-667, 679-766): FindSubmatchIndex from a moving
// offset, per pattern and strand; a match that lies inside an earlier one of the same pattern and strand is dropped
// (:604-614).  One lane per record runs the Pike VM of regex_vm.hpp on the text view (the '-' strand is the
// complement read backwards, never materialised).  Within one (pattern, strand) the match starts only grow, so
// "inside an earlier match" is "does not end beyond the furthest end printed so far".  Count pass / emit pass as above.
// ---------------------------------------------------------------------------
template <bool EMIT>
__global__ __launch_bounds__(64) void k_locate_vm(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t, TextTable tt,
                                                  LocateParams P, const VmProgram* __restrict__ progs,
                                                  uint32_t* __restrict__ out_len, const uint64_t* __restrict__ out_off,
                                                  uint8_t* __restrict__ out, uint64_t* __restrict__ rows,
                                                  const uint32_t* __restrict__ list, uint64_t nlist) {
    // with a list (records the boolean automaton of grep -r let through / records with rows): one lane per ENTRY, so
    // that the lanes of a wave all have work -- one record in fifty matching left 63 lanes of 64 idle for the whole walk
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (list ? nlist : t.n)) return;
    const uint64_t g = list ? (uint64_t)list[idx] : idx;
    if (EMIT && out_len[g] == 0) return;
    const Text T = text_of(buf, t, tt, g);
    const uint32_t l = T.L;
    RowCtx R;
    {
        const uint32_t lh = t.l_head[g];
        const uint8_t* h = buf + t.start[g] + 1;
        uint32_t off;
        R.id_len = id_span_rec(t, g, h, lh > 0 ? lh - 1 : 0, P.id_mode, &off, buf + buf_n);
        R.id = h + off;
    }
    R.format = P.format;
    R.from_text = true;
    R.lower = false;
    R.T = T;
    R.l = l;
    R.comp = P.comp;
    uint8_t* o = EMIT ? out + out_off[g] : nullptr;
    uint64_t bytes = 0;
    uint32_t nrows = 0;
    const uint8_t* comp = P.comp;
    for (int k = 0; k < P.npat; ++k) {
        R.name = P.name + P.name_off[k];
        R.name_len = P.name_off[k + 1] - P.name_off[k];
        R.disp = P.disp + P.disp_off[k];
        R.disp_len = P.disp_off[k + 1] - P.disp_off[k];
        R.pat = nullptr;
        const VmProgram& prog = progs[k];
        // --circular (locate.go:231-234, 595-597, 694-703): the text is the sequence twice, a match must begin in the first
        // copy, and on the '-' strand a match that reaches into the second copy is reported l further on
        const uint32_t tl = P.circular ? 2u * l : l;   // length of the text that is searched
        auto fw = [&](uint32_t i) -> uint8_t { return T.at(i >= l ? i - l : i); };
        for (int strand = 0; strand < (P.both_strands ? 2 : 1); ++strand) {
            uint32_t offset = 0;
            uint32_t far = 0;     // furthest end (in the strand's own coordinates) of a printed match
            uint32_t far_x = 0;   // ... among the '-' matches that reach into the second copy (their rows are shifted by l)
            bool any = false, any_x = false;
            // An expression that consumes a byte per step (no ^ / $, not nullable) has all its matches end where the boolean
            // automaton of grep -r is in a final state: a strand on which it never is holds no match, and no match ends
            // beyond the last such position -- the matcher (the costly one) walks [offset, lim) instead of [offset, l), the
            // matches it reports are the same (those of a start are the ones that end by lim, in the same priority order).
            uint32_t lim = tl;
            if (P.pre_regex && !P.circular) {
                const RegexProgram& pr = P.pre_regex[k];
                if (!pr.nullable && pr.accept[RE_SYM_BEGIN] == 0 && pr.accept[RE_SYM_END] == 0) {
                    const uint32_t nchunk = (pr.npos + 7u) >> 3;
                    uint64_t S = 0;
                    uint32_t last_e = 0;
                    bool acc = false;
                    for (uint32_t x = 0; x < l; ++x) {
                        const uint32_t ch = strand == 0 ? T.at(x) : comp[T.at(l - 1u - x)];
                        uint64_t f = pr.first;
                        for (uint32_t q = 0; q < nchunk; ++q) f |= pr.follow[q][(S >> (8u * q)) & 255u];
                        S = f & pr.accept[ch];
                        if (S & pr.last) { acc = true; last_e = x + 1u; }
                    }
                    if (!acc) continue;
                    lim = last_e;
                }
            }
            for (;;) {
                if (offset > lim) break;
                uint32_t caps[4];
                bool found;
                if (!P.circular) {
                    if (strand == 0) found = vm_search_fn(prog, [&](uint32_t i) { return T.at(offset + i); }, lim - offset, 0u, caps);
                    else found = vm_search_fn(prog, [&](uint32_t i) { return comp[T.at(l - 1u - (offset + i))]; }, lim - offset, 0u, caps);
                } else {
                    if (strand == 0) found = vm_search_fn(prog, [&](uint32_t i) { return fw(offset + i); }, lim - offset, 0u, caps);
                    else found = vm_search_fn(prog, [&](uint32_t i) { return comp[fw(tl - 1u - (offset + i))]; }, lim - offset, 0u, caps);
                }
                if (!found) break;
                const uint32_t s = offset + caps[0], e = offset + caps[1];  // the match in the strand's coordinates [s, e)
                if (P.circular && s >= l) break;                            // "2nd clone of original part"
                // inside an earlier match of this pattern and strand?  Starts only grow, so it is a question of ends -- in
                // the coordinates of the ROWS: a '-' match that crosses the origin is shifted by l, an earlier one that does
                // not cross holds a later crossing one never, an earlier crossing one holds a later plain one iff it ends
                // l further on
                const bool cross = P.circular && strand != 0 && e > l;
                bool inside;
                if (cross) inside = any_x && far_x >= e;
                else inside = (any && far >= e) || (any_x && (uint64_t)far_x >= (uint64_t)l + e);
                if (!inside) {
                    int64_t begin = strand == 0 ? (int64_t)s + 1 : (int64_t)l - (int64_t)e + 1;
                    int64_t end = strand == 0 ? (int64_t)e : (int64_t)l - (int64_t)s;
                    if (cross) { begin += l; end += l; }
                    R.pat_len = e - s;
                    R.rc = strand != 0;
                    if (strand == 0) R.f = s;
                    else { uint64_t f = (uint64_t)tl - e; while (l && f >= l) f -= l; R.f = f; }
                    if (EMIT) bytes += row_put(o + bytes, R, strand == 0 ? '+' : '-', begin, end);
                    else bytes += row_len(R, begin, end);
                    ++nrows;
                    if (cross) { far_x = any_x ? (e > far_x ? e : far_x) : e; any_x = true; }
                    else { far = any ? (e > far ? e : far) : e; any = true; }
                }
                offset = P.non_greedy ? e + 1u : s + 1u;
                if (offset >= lim) break;
            }
        }
    }
    if (EMIT) {
        if (nrows) atomicAdd((unsigned long long*)rows, (unsigned long long)nrows);
    } else {
        out_len[g] = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bytes;
    }
}

// records with rows -> hit_list (any order).  Each block owns a contiguous chunk: count, ONE atomicAdd to reserve the
// slots, then write -- a few thousand atomics instead of one per hit.
__global__ __launch_bounds__(256) void k_compact_hits(const uint32_t* __restrict__ out_len, uint64_t n, uint32_t* __restrict__ hit_list,
                                                      unsigned long long* __restrict__ hit_count) {
    __shared__ unsigned int s_cnt;
    __shared__ unsigned long long s_base;
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned int mine = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) mine += out_len[i] != 0;
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) { s_base = s_cnt ? atomicAdd(hit_count, (unsigned long long)s_cnt) : 0ull; s_cnt = 0; }
    __syncthreads();
    if (lo >= hi) return;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
        if (out_len[i] != 0) hit_list[s_base + atomicAdd(&s_cnt, 1u)] = (uint32_t)i;
}

// out_len of a long record = bytes of all its cells
__global__ void k_locate_long_sizes(LocateParams P, uint32_t* __restrict__ out_len) {
    const uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= P.long_count) return;
    out_len[P.long_list[y]] = (uint32_t)(P.cell_off[P.cellbase[y + 1]] - P.cell_off[P.cellbase[y]]);
}

__global__ void k_locate_long_cells(RecordTable t, LocateParams P, uint32_t* __restrict__ ncells) {
    const uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= P.long_count) return;
    const uint64_t l = t.l_seq[P.long_list[y]];
    const uint64_t n = P.circular ? 2ull * l : l;
    ncells[y] = (uint32_t)((n + LOCATE_LONG_CH - 1) / LOCATE_LONG_CH) * (uint32_t)(P.npat * (P.both_strands ? 2 : 1));
}

}  // namespace

hipError_t launch_locate_long_cells(const RecordTable& t, const LocateParams& P, uint32_t* ncells, hipStream_t st) {
    if (!P.long_count) return hipSuccess;
    hipLaunchKernelGGL(k_locate_long_cells, dim3((unsigned)((P.long_count + 255) / 256)), dim3(256), 0, st, t, P, ncells);
    return hipGetLastError();
}

hipError_t launch_locate_long_sizes(const LocateParams& P, uint32_t* out_len, hipStream_t st) {
    if (!P.long_count) return hipSuccess;
    hipLaunchKernelGGL(k_locate_long_sizes, dim3((unsigned)((P.long_count + 255) / 256)), dim3(256), 0, st, P, out_len);
    return hipGetLastError();
}

hipError_t launch_compact_hits(const uint32_t* out_len, uint64_t n, uint32_t* hit_list, uint64_t* hit_count, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 4095) / 4096;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_compact_hits, dim3((unsigned)blocks), dim3(256), 0, st, out_len, n, hit_list, (unsigned long long*)hit_count);
    return hipGetLastError();
}

hipError_t launch_locate(bool emit, const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const TextTableH& tt,
                         const LocateParams& P, uint32_t* out_len, const uint64_t* out_off, uint8_t* out,
                         uint64_t* rows, hipStream_t st, uint64_t avg_record_bytes) {
    if (t.n == 0) return hipSuccess;
    TextTable d{tt.text_w, tt.lin_off, tt.lin, tt.lin_n};
    const uint64_t groups = emit ? P.nhit : (P.cand ? P.ncand : t.n);
    const bool small = (avg_record_bytes ? avg_record_bytes : buf_n / t.n) < 1024;  // bytes per record (not of a filtered table)
    const int G = small ? 4 : 16;
    const uint64_t blocks = (groups * G + 255) / 256;
    const dim3 gr((unsigned)blocks), bl(256);
#define BSK_LAUNCH_LOCATE(E, GE, GG) hipLaunchKernelGGL((k_locate<E, GE, GG, false>), gr, bl, 0, st, buf, buf_n, t, d, P, out_len, out_off, out, rows)
    if (groups) {
        if (small) {
            if (P.general) { if (emit) BSK_LAUNCH_LOCATE(true, true, 4); else BSK_LAUNCH_LOCATE(false, true, 4); }
            else { if (emit) BSK_LAUNCH_LOCATE(true, false, 4); else BSK_LAUNCH_LOCATE(false, false, 4); }
        } else {
            if (P.general) { if (emit) BSK_LAUNCH_LOCATE(true, true, 16); else BSK_LAUNCH_LOCATE(false, true, 16); }
            else { if (emit) BSK_LAUNCH_LOCATE(true, false, 16); else BSK_LAUNCH_LOCATE(false, false, 16); }
        }
    }
#undef BSK_LAUNCH_LOCATE
    if (P.long_thresh && P.long_count && P.long_cells) {
        // one wave per cell of every long record: 4 cells per block
        const dim3 lg((unsigned)((P.long_cells + 3u) / 4u));
#define BSK_LAUNCH_LONG(E, GE) hipLaunchKernelGGL((k_locate<E, GE, 64, true>), lg, bl, 0, st, buf, buf_n, t, d, P, out_len, out_off, out, rows)
        if (P.general) { if (emit) BSK_LAUNCH_LONG(true, true); else BSK_LAUNCH_LONG(false, true); }
        else { if (emit) BSK_LAUNCH_LONG(true, false); else BSK_LAUNCH_LONG(false, false); }
#undef BSK_LAUNCH_LONG
    }
    return hipGetLastError();
}

hipError_t launch_locate_vm(bool emit, const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const TextTableH& tt,
                            const LocateParams& P, const VmProgram* d_progs, uint32_t* out_len, const uint64_t* out_off,
                            uint8_t* out, uint64_t* rows, hipStream_t st, const uint32_t* list, uint64_t nlist) {
    if (t.n == 0 || (list && nlist == 0)) return hipSuccess;
    TextTable d{tt.text_w, tt.lin_off, tt.lin, tt.lin_n};
    const uint64_t items = list ? nlist : t.n;
    const dim3 gr((unsigned)((items + 63) / 64)), bl(64);
    if (emit) hipLaunchKernelGGL(k_locate_vm<true>, gr, bl, 0, st, buf, buf_n, t, d, P, d_progs, out_len, out_off, out, rows, list, nlist);
    else hipLaunchKernelGGL(k_locate_vm<false>, gr, bl, 0, st, buf, buf_n, t, d, P, d_progs, out_len, out_off, out, rows, list, nlist);
    return hipGetLastError();
}

}  // namespace bsk
