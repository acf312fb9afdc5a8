// ============================================================================
// ops_grep.hip -- Grep.grepGeneral (/root/reference/bigseqkit-lib/grep.go:367-542)
// for exact patterns, on the record table.
//   by sequence (-s): 16 lanes per record test the start positions in parallel;
//     the '-' strand is searched as reverse-complemented PATTERNS on the forward
//     text (RevCom is an involution on bytes, so  P in RevCom(S)  <=>  RevCom(P) in S),
//     which halves the traffic of the reference's RevCom(seq) copy (grep.go:445-448).
//     Regions (-R) and --circular are index maps on the same text.
//   by ID / name: one thread per record, exact comparison against the pattern set.
// Output of the kernel: bytes the record contributes (0 = not selected); the record
// itself is emitted by k_seq_emit after the scan.
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "ops_grep.hpp"
#include "ops_seq.hpp"
#include "ops_translate.hpp"  // TextTableH
#include "pattern_match_dev.hpp"
#include "regex_nfa.hpp"
#include "regex_vm.hpp"

namespace bsk {

namespace {

constexpr int GROUP = 16;
#ifndef BSK_GREP_LANES
#define BSK_GREP_LANES 4  // lanes per short record in k_grep_seq (measured: 16 -> 10.0 ms, 4 -> 6 ms at C3)
#endif

__device__ __forceinline__ uint8_t lower8(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }


// formatted length of the whole record, fastx.Record.Format(width)
__device__ __forceinline__ uint32_t format_len(uint32_t name_len, uint32_t L, int fastq, int width) {
    uint32_t w = L;
    if (width > 0 && L > 0) w += (L - 1) / (uint32_t)width;
    uint32_t n = 1 + name_len + 1 + w + 1;
    if (fastq) n += 2 + w + 1;
    return n;
}

// ---------------------------------------------------------------------------
// fast path for contiguous text: a lane tests 16 consecutive start positions from two
// unaligned 16-byte loads; the first min(m, 4) pattern bytes are compared as one dword taken
// with v_alignbyte at a static shift, the (rare) survivors are verified byte by byte.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fold_dword(uint32_t x) {  // ASCII lower-case, 4 bytes at once
    const uint32_t ge_a = (x & 0x7F7F7F7Fu) + 0x3F3F3F3Fu;   // bit 7 set where byte >= 'A'
    const uint32_t ge_z1 = (x & 0x7F7F7F7Fu) + 0x25252525u;  // bit 7 set where byte >= '[' 
    const uint32_t up = ge_a & ~ge_z1 & ~x & 0x80808080u;      // 'A'..'Z' (and byte < 0x80)
    return x | (up >> 2);
}

__device__ __forceinline__ uint32_t window_candidates(const uint32_t (&dw)[8], uint32_t p32, uint32_t pmask) {
    // per start position: v_alignbyte, xor, (and), min(.,1), v_lshl_or -- the mismatch bit is accumulated, no compares
    uint32_t miss = 0;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int d = b >> 2, sft = b & 3;
        const uint32_t w = sft == 0 ? dw[d] : __builtin_amdgcn_alignbyte(dw[d + 1], dw[d], sft);
        uint32_t t = (w ^ p32) & pmask;
        t = t < 1u ? t : 1u;
        miss |= t << b;
    }
    return ~miss & 0xFFFFu;
}

// GROUP lanes per record: the kernel is bound by the chain of dependent loads per wave (record table -> window -> verify),
// so short reads use 4 lanes (16 records per wave in flight), long sequences 16.
// LONG: sequences of at least P.long_thresh bases (chromosomes) are skipped here and searched by whole blocks,
// GREP_LONG_CH start positions per block (grid = chunks x long records); a hit sets P.long_hit[record slot] and
// k_grep_long_finish turns the flags into output sizes.
constexpr uint32_t GREP_LONG_CH = 256u * 1024u;

template <int GROUP, bool LONG>
__global__ __launch_bounds__(256) void k_grep_seq(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t, TextTable tt,
                                                  GrepParams P, uint32_t* __restrict__ out_len) {
    constexpr uint32_t LANES = LONG ? 256u : (uint32_t)GROUP;
    const uint64_t g = LONG ? (uint64_t)P.long_list[blockIdx.y] : ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
    const uint32_t gl = LONG ? threadIdx.x : threadIdx.x % GROUP;
    const uint32_t gshift = (threadIdx.x & 63) / GROUP * GROUP;  // position of this group inside the wave
    constexpr uint64_t GMASK = (1ull << GROUP) - 1ull;
    const bool live = g < t.n;
    const uint64_t gi = live ? g : 0;
    const Text T = text_of(buf, t, tt, gi);  // wrapped FASTA in place (W), irregular FASTA linearised
    const uint32_t L = live ? T.L : 0;
    const uint32_t lh = t.l_head[gi];
    if (!LONG && P.long_thresh && L >= P.long_thresh) return;  // searched by the LONG launch (whole groups leave)
    const uint64_t plo = LONG ? (uint64_t)blockIdx.x * GREP_LONG_CH : 0ull;  // this block's start positions [plo, phi)
    const uint64_t phi = LONG ? plo + GREP_LONG_CH : ~0ull;
    bool hit = false;
    // strand_only (1: '+', 2: '-'): one strand alone, for the per-(pattern, strand) hit bits of --delete-matched
    const int nstr = P.strand_only == 1 ? 1 : (P.both_strands ? 2 : 1);
    const int str0 = P.strand_only == 2 ? 1 : 0;
    // contiguous text: in the shard itself, or in the linear copies of wrapped FASTA records (text_dev.hpp)
    const bool in_buf = T.p >= buf && T.p < buf + buf_n;
    const bool in_lin = tt.lin_n != 0 && T.p >= tt.lin && T.p < tt.lin + tt.lin_n;
    const bool fast = live && T.W == 0 && !P.circular && (in_buf || in_lin);
    if (fast) {
        const uint8_t* const buf_end = in_buf ? buf + buf_n : tt.lin + tt.lin_n;  // (no wide load past it)
        for (int strand = str0; strand < nstr && !hit; ++strand) {
            uint32_t wb = 0, we = L;
            if (P.region_on) {
                uint32_t b, e;
                sub_location(L, P.region_start, P.region_end, &b, &e);
                if (strand == 0) { wb = b; we = e; }
                else { wb = L - e; we = L - b; }
            }
            const uint32_t wl = we - wb;
            for (int k = 0; k < P.npat && !hit; ++k) {
                const int pk = strand * P.npat + k;
                const uint8_t* pp = P.pat + P.pat_off[pk];
                const uint32_t m = P.pat_off[pk + 1] - P.pat_off[pk];
                if (m == 0) { hit = true; break; }
                if (m > wl) continue;
                const uint32_t npos = wl - m + 1;
                uint32_t p32 = 0;
                for (uint32_t q = 0; q < m && q < 4; ++q) p32 |= (uint32_t)pp[q] << (8 * q);
                const uint32_t pmask = m >= 4 ? 0xFFFFFFFFu : ((1u << (8 * m)) - 1u);
                const uint32_t iend = phi < npos ? (uint32_t)phi : npos;
                for (uint32_t i0 = (uint32_t)(plo < npos ? plo : npos); i0 < iend; i0 += LANES * 16) {
                    const uint32_t ib = i0 + gl * 16u;  // first start position of this lane
                    bool ok = false;
                    if (ib < iend) {
                        const uint8_t* src = T.p + wb + ib;
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
                        if (P.ignore_case) {
#pragma unroll
                            for (int d = 0; d < 8; ++d) dw[d] = fold_dword(dw[d]);
                        }
                        uint32_t cand = window_candidates(dw, p32, pmask);
                        const uint32_t left = npos - ib;  // valid start positions from ib
                        if (left < 16u) cand &= (1u << left) - 1u;
                        while (cand && !ok) {
                            const uint32_t b = (uint32_t)__ffs((int)cand) - 1u;
                            cand &= cand - 1u;
                            ok = verify_from4(src + b, buf_end, P.ignore_case, pp, m);
                        }
                    }
                    if (LONG) { if (ok) hit = true; }
                    else {
                        const uint64_t any = __ballot(ok);
                        if ((any >> gshift) & GMASK) { hit = true; break; }
                    }
                }
            }
        }
    } else if (live) {
        for (int strand = str0; strand < nstr && !hit; ++strand) {
            // window of the forward text that the strand's target covers
            uint32_t wb = 0, we = L;
            if (P.region_on) {
                uint32_t b, e;
                sub_location(L, P.region_start, P.region_end, &b, &e);
                if (strand == 0) { wb = b; we = e; }
                else { wb = L - e; we = L - b; }  // SubSeq of RevCom(S) mirrored onto S
            }
            const uint32_t wl = we - wb;
            const uint64_t tl = P.circular ? 2ull * wl : wl;  // --circular doubles the target (grep.go:449-454)
            for (int k = 0; k < P.npat && !hit; ++k) {
                const int pk = strand * P.npat + k;
                const uint8_t* pp = P.pat + P.pat_off[pk];
                const uint32_t m = P.pat_off[pk + 1] - P.pat_off[pk];
                if (m == 0) { hit = true; break; }  // bytes.Contains(x, "") is true
                if (m > tl) continue;
                const uint64_t npos = tl - m + 1;
                const uint8_t p0 = pp[0];
                const uint64_t iend = phi < npos ? phi : npos;
                for (uint64_t i0 = plo < npos ? plo : npos; i0 < iend; i0 += LANES) {
                    const uint64_t i = i0 + gl;
                    bool ok = false;
                    if (i < iend) {
                        uint64_t j = i >= wl ? i - wl : i;  // circular wrap
                        uint8_t c = T.at(wb + (uint32_t)j);
                        if (P.ignore_case) c = lower8(c);
                        if (c == p0) {
                            ok = true;
                            for (uint32_t q = 1; q < m; ++q) {
                                uint64_t jj = i + q;
                                if (jj >= wl) jj -= wl;  // only reachable when circular (jj < 2 wl)
                                uint8_t cc = T.at(wb + (uint32_t)jj);
                                if (P.ignore_case) cc = lower8(cc);
                                if (cc != pp[q]) { ok = false; break; }
                            }
                        }
                    }
                    if (LONG) { if (ok) hit = true; }
                    else {
                        const uint64_t any = __ballot(ok);
                        if ((any >> gshift) & GMASK) { hit = true; break; }
                    }
                }
            }
        }
    }
    if (LONG) {
        if (hit) atomicOr(&P.long_hit[blockIdx.y], 1u);
        return;
    }
    if (live && gl == 0) {
        const bool sel = P.invert ? !hit : hit;
        out_len[g] = sel ? format_len(lh > 0 ? lh - 1 : 0, L, P.fastq, P.line_width) : 0u;
    }
}

// output sizes of the long records from the flags the LONG search left
__global__ void k_grep_long_finish(RecordTable t, GrepParams P, uint32_t* __restrict__ out_len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.long_count) return;
    const uint64_t g = P.long_list[i];
    const bool hit = P.long_hit[i] != 0;
    const bool sel = P.invert ? !hit : hit;
    const uint32_t lh = t.l_head[g];
    out_len[g] = sel ? format_len(lh > 0 ? lh - 1 : 0, t.l_seq[g], P.fastq, P.line_width) : 0u;
}

// ---------------------------------------------------------------------------
// class patterns (-d / -m): 16 lanes per record, one start position per lane and step; the text is
// seen through the Text view (uniform-width FASTA in place, irregular FASTA linearised)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grep_seq_gen(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt,
                                                      GrepParams P, uint32_t* __restrict__ out_len) {
    const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
    const uint32_t gl = threadIdx.x % GROUP;
    const uint32_t gshift = (threadIdx.x & 63) / GROUP * GROUP;
    const bool live = g < t.n;
    const uint64_t gi = live ? g : 0;
    const Text T = text_of(buf, t, tt, gi);
    const uint32_t L = live ? T.L : 0;
    bool hit = false;
    // strand_only (1: '+', 2: '-'): one strand alone, for the per-(pattern, strand) hit bits of --delete-matched
    const int nstr = P.strand_only == 1 ? 1 : (P.both_strands ? 2 : 1);
    const int str0 = P.strand_only == 2 ? 1 : 0;
    for (int strand = str0; strand < nstr && !hit; ++strand) {
        uint32_t wb = 0, we = L;
        if (P.region_on) {
            uint32_t b, e;
            sub_location(L, P.region_start, P.region_end, &b, &e);
            if (strand == 0) { wb = b; we = e; }
            else { wb = L - e; we = L - b; }
        }
        const uint32_t wl = we - wb;
        const uint64_t tl = P.circular ? 2ull * wl : wl;
        Text V = T;  // the window as its own text (contiguous or linearised) or via an offset (wrapped FASTA)
        for (int k = 0; k < P.npat && !hit; ++k) {
            const int pk = strand * P.npat + k;
            const uint32_t m = P.pat_off[pk + 1] - P.pat_off[pk];
            const uint32_t* cls = P.cls + (uint64_t)P.pat_off[pk] * 8u;
            if (m == 0) { hit = true; break; }
            if (m > tl) continue;
            const uint64_t npos = tl - m + 1;
            for (uint64_t i0 = 0; i0 < npos; i0 += GROUP) {
                const uint64_t i = i0 + gl;
                bool ok = false;
                if (i < npos) {
                    int mm = 0;
                    ok = true;
                    for (uint32_t q = 0; q < m; ++q) {
                        uint64_t j = i + q;
                        if (j >= wl) j -= wl;
                        if (j >= wl) j -= wl;  // i < 2 wl and q < m <= 2 wl
                        const uint8_t c = V.at(wb + (uint32_t)j);
                        if (!((cls[q * 8u + (c >> 5)] >> (c & 31u)) & 1u))
                            if (++mm > P.max_mm) { ok = false; break; }
                    }
                }
                const uint64_t any = __ballot(ok);
                if ((any >> gshift) & 0xFFFFull) { hit = true; break; }
            }
        }
    }
    if (live && gl == 0) {
        const uint32_t lh = t.l_head[gi];
        const bool sel = P.invert ? !hit : hit;
        out_len[g] = sel ? format_len(lh > 0 ? lh - 1 : 0, L, P.fastq, P.line_width) : 0u;
    }
}

// ---------------------------------------------------------------------------
// class patterns of at most 64 positions, at most SA_MAX_MM mismatches, not circular: Shift-And with mismatch rows
// (Wu-Manber without insertions and deletions), one lane per record.  B[c] = bit q set where byte c is in the class of
// pattern position q, one 2 KB table per (strand, pattern) in LDS; per text byte
//     R_j = ((R_j << 1 | 1) & B[c]) | (R_{j-1} << 1 | 1),      hit when bit m-1 of R_maxmm is set.
// The same start positions and the same mismatch count as k_grep_seq_gen tests one by one (a window holds a match iff
// some m consecutive bytes of it differ from the classes in at most max_mm places).  The text comes 16 bytes per load.
// ---------------------------------------------------------------------------
constexpr int SA_MAX_TABLES = 8;
constexpr int SA_MAX_MM = 3;

// NT tables over the same window in one pass of the text (both strands of one pattern when no region is set)
template <int KMM, int NT>
__device__ __forceinline__ bool sa_search(const uint64_t* __restrict__ B, const uint32_t (&mt)[NT], const Text& T, uint32_t wb,
                                          uint32_t wl, bool contig, const uint8_t* hi) {
    uint64_t R[NT][KMM + 1];
    uint64_t last[NT], seen[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int j = 0; j <= KMM; ++j) R[t][j] = 0;
        last[t] = 1ull << (mt[t] - 1u);
        seen[t] = 0;
    }
    for (uint32_t x0 = 0; x0 < wl; x0 += 16u) {
        const uint32_t nb = wl - x0 < 16u ? wl - x0 : 16u;
        uint32_t w[4] = {0, 0, 0, 0};
        if (contig && T.p + wb + x0 + 16 <= hi) {
            uint4 v;
            __builtin_memcpy(&v, T.p + wb + x0, 16);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            for (uint32_t b = 0; b < nb; ++b) w[b >> 2] |= (uint32_t)T.at(wb + x0 + b) << (8u * (b & 3u));
        }
        if (nb == 16u) {
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const uint32_t ch = (w[b >> 2] >> (8 * (b & 3))) & 255u;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const uint64_t bc = B[t * 256 + ch];
#pragma unroll
                    for (int j = KMM; j >= 1; --j) R[t][j] = (((R[t][j] << 1) | 1ull) & bc) | ((R[t][j - 1] << 1) | 1ull);
                    R[t][0] = ((R[t][0] << 1) | 1ull) & bc;
                    seen[t] |= R[t][KMM];
                }
            }
        } else {
            for (uint32_t b = 0; b < nb; ++b) {
                const uint32_t ch = (w[b >> 2] >> (8u * (b & 3u))) & 255u;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const uint64_t bc = B[t * 256 + ch];
#pragma unroll
                    for (int j = KMM; j >= 1; --j) R[t][j] = (((R[t][j] << 1) | 1ull) & bc) | ((R[t][j - 1] << 1) | 1ull);
                    R[t][0] = ((R[t][0] << 1) | 1ull) & bc;
                    seen[t] |= R[t][KMM];
                }
            }
        }
        bool any = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) any |= (seen[t] & last[t]) != 0;
        if (any) return true;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_grep_shiftand(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t, TextTable tt,
                                                       GrepParams P, uint32_t* __restrict__ out_len) {
    __shared__ uint64_t s_B[SA_MAX_TABLES * 256];
    const int nstr_all = P.both_strands ? 2 : 1;
    const int ntab = nstr_all * P.npat;
    for (int tb = 0; tb < ntab; ++tb) {
        const uint32_t m = P.pat_off[tb + 1] - P.pat_off[tb];
        const uint32_t* cls = P.cls + (uint64_t)P.pat_off[tb] * 8u;
        const uint32_t c = threadIdx.x;  // blockDim.x == 256: one byte value per thread
        uint64_t bits = 0;
        for (uint32_t q = 0; q < m; ++q) bits |= (uint64_t)((cls[q * 8u + (c >> 5)] >> (c & 31u)) & 1u) << q;
        s_B[tb * 256 + c] = bits;
    }
    __syncthreads();
    const int nstr = P.strand_only == 1 ? 1 : nstr_all;
    const int str0 = P.strand_only == 2 ? 1 : 0;
    const bool pair = P.npat == 1 && nstr == 2 && str0 == 0 && !P.region_on;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Text T = text_of(buf, t, tt, i);
        const uint32_t L = T.L;
        const bool in_buf = T.p >= buf && T.p < buf + buf_n;
        const bool in_lin = tt.lin_n != 0 && T.p >= tt.lin && T.p < tt.lin + tt.lin_n;
        const bool contig = T.W == 0 && (in_buf || in_lin);
        const uint8_t* hi = in_buf ? buf + buf_n : tt.lin + tt.lin_n;
        bool hit = false;
        if (pair) {  // tables 0 ('+') and 1 ('-') of the one pattern, whole sequence
            const uint32_t m = P.pat_off[1] - P.pat_off[0];
            if (m == 0) hit = true;
            else if (m <= L) {
                const uint32_t mt[2] = {m, m};
                switch (P.max_mm) {
                    case 0: hit = sa_search<0, 2>(s_B, mt, T, 0, L, contig, hi); break;
                    case 1: hit = sa_search<1, 2>(s_B, mt, T, 0, L, contig, hi); break;
                    case 2: hit = sa_search<2, 2>(s_B, mt, T, 0, L, contig, hi); break;
                    default: hit = sa_search<3, 2>(s_B, mt, T, 0, L, contig, hi); break;
                }
            }
        } else
        for (int strand = str0; strand < nstr && !hit; ++strand) {
            uint32_t wb = 0, we = L;
            if (P.region_on) {
                uint32_t b, e;
                sub_location(L, P.region_start, P.region_end, &b, &e);
                if (strand == 0) { wb = b; we = e; }
                else { wb = L - e; we = L - b; }
            }
            const uint32_t wl = we - wb;
            for (int k = 0; k < P.npat && !hit; ++k) {
                const int pk = strand * P.npat + k;
                const uint32_t m = P.pat_off[pk + 1] - P.pat_off[pk];
                if (m == 0) { hit = true; break; }
                if (m > wl) continue;
                const uint64_t* B = s_B + pk * 256;
                const uint32_t mt[1] = {m};
                switch (P.max_mm) {
                    case 0: hit = sa_search<0, 1>(B, mt, T, wb, wl, contig, hi); break;
                    case 1: hit = sa_search<1, 1>(B, mt, T, wb, wl, contig, hi); break;
                    case 2: hit = sa_search<2, 1>(B, mt, T, wb, wl, contig, hi); break;
                    default: hit = sa_search<3, 1>(B, mt, T, wb, wl, contig, hi); break;
                }
            }
        }
        const uint32_t lh = t.l_head[i];
        const bool sel = P.invert ? !hit : hit;
        out_len[i] = sel ? format_len(lh > 0 ? lh - 1 : 0, L, P.fastq, P.line_width) : 0u;
    }
}

// ---------------------------------------------------------------------------
// -r: re.Match(target) (grep.go:459-468) as a bit-parallel position automaton, one lane per record.
// State = 64-bit set of active positions; per byte: S = (follow(S) | first) & accept[byte]; follow(S) is the OR of
// one table entry per 8 state bits.  The search is unanchored (first is injected before every byte); ^ and $ are
// positions that accept the virtual BEGIN / END symbols fed before the first and after the last byte.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool re_step(const RegexProgram& p, uint32_t nchunk, uint64_t& S, int sym) {
    uint64_t f = p.first;
    for (uint32_t k = 0; k < nchunk; ++k) f |= p.follow[k][(S >> (8u * k)) & 255u];
    S = f & p.accept[sym];
    return (S & p.last) != 0;
}

// One expression on contiguous text: its tables (accept 2 KB + follow 2 KB per 8 positions) and the complement map sit in
// LDS, the text comes 16 bytes per load.  With the tables in global memory every byte of the text cost two dependent
// cache round trips on one lane: 82 ms for 5 GB of 150-base reads (both strands).
struct RegexLds {
    const uint64_t* acc;     // [RE_NSYM]
    const uint64_t* fol;     // [nchunk][256]
    const uint8_t* comp;     // [256]
    uint64_t first, last;
    uint32_t nchunk;
};
__device__ __forceinline__ bool re_step_lds(const RegexLds& R, uint64_t& S, uint32_t sym) {
    uint64_t f = R.first;
    for (uint32_t k = 0; k < R.nchunk; ++k) f |= R.fol[k * 256u + ((uint32_t)(S >> (8u * k)) & 255u)];
    S = f & R.acc[sym];
    return (S & R.last) != 0;
}
// unanchored search over bases [wb, wb + wl) of the strand (rc: of the reverse complement) of the text p[0, L); `lo` / `hi`
// bound the 16-byte loads
__device__ bool re_search_contig(const RegexLds& R, const uint8_t* p, uint32_t L, uint32_t wb, uint32_t wl, bool rc,
                                 const uint8_t* lo, const uint8_t* hi) {
    uint64_t S = 0;
    bool hit = re_step_lds(R, S, RE_SYM_BEGIN);
    for (uint32_t x0 = 0; x0 < wl && !hit; x0 += 16u) {
        const uint32_t nb = wl - x0 < 16u ? wl - x0 : 16u;
        // strand bases x0 .. x0 + nb - 1 are text bytes wb + x0 .. (forward) or L - 1 - (wb + x0) downwards (reverse)
        const uint8_t* src = rc ? p + (L - (wb + x0) - nb) : p + wb + x0;
        uint32_t w[4] = {0, 0, 0, 0};
        if (nb == 16u && src >= lo && src + 16 <= hi) {
            uint4 v;
            __builtin_memcpy(&v, src, 16);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            for (uint32_t b = 0; b < nb; ++b) w[b >> 2] |= (uint32_t)src[b] << (8u * (b & 3u));
        }
        for (uint32_t b = 0; b < nb && !hit; ++b) {
            const uint32_t q = rc ? nb - 1u - b : b;
            uint32_t c = (w[q >> 2] >> (8u * (q & 3u))) & 255u;
            if (rc) c = R.comp[c];
            hit = re_step_lds(R, S, c);
        }
    }
    if (!hit) hit = re_step_lds(R, S, RE_SYM_END);
    return hit;
}

__global__ __launch_bounds__(256) void k_grep_regex(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt,
                                                    GrepParams P, uint32_t* __restrict__ out_len, uint64_t buf_n) {
    __shared__ uint64_t s_acc[RE_NSYM];
    __shared__ uint64_t s_fol[8 * 256];
    __shared__ uint8_t s_comp[256];
    const bool lds_ok = P.by_seq && P.npat == 1 && !P.circular;  // (block-uniform)
    RegexLds R{s_acc, s_fol, s_comp, 0, 0, 0};
    if (lds_ok) {
        const RegexProgram& p0 = P.regex[0];
        R.first = p0.first; R.last = p0.last; R.nchunk = (p0.npos + 7u) >> 3;
        for (uint32_t k = threadIdx.x; k < (uint32_t)RE_NSYM; k += blockDim.x) s_acc[k] = p0.accept[k];
        for (uint32_t k = threadIdx.x; k < R.nchunk * 256u; k += blockDim.x) s_fol[k] = p0.follow[k >> 8][k & 255u];
        for (uint32_t k = threadIdx.x; k < 256u; k += blockDim.x) s_comp[k] = P.comp ? P.comp[k] : (uint8_t)k;
        __syncthreads();
    }
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint32_t lh = t.l_head[i];
    const uint8_t* h = buf + t.start[i] + 1;
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    bool hit = false;
    if (P.by_seq) {
        const Text T = text_of(buf, t, tt, i);
        const uint32_t L = T.L;
        // strand_only (1: '+', 2: '-'): one strand alone, for the per-(pattern, strand) hit bits of --delete-matched
    const int nstr = P.strand_only == 1 ? 1 : (P.both_strands ? 2 : 1);
    const int str0 = P.strand_only == 2 ? 1 : 0;
        const bool in_buf = T.p >= buf && T.p < buf + buf_n;
        const bool in_lin = tt.lin_n != 0 && T.p >= tt.lin && T.p < tt.lin + tt.lin_n;
        const bool contig = lds_ok && T.W == 0 && (in_buf || in_lin) && !P.regex[0].nullable;
        for (int strand = str0; strand < nstr && !hit; ++strand) {
            uint32_t wb = 0, we = L;  // window in the strand's own coordinates
            if (P.region_on) sub_location(L, P.region_start, P.region_end, &wb, &we);
            const uint32_t wl = we - wb;
            if (contig) {
                hit = re_search_contig(R, T.p, L, wb, wl, strand != 0, in_buf ? buf : tt.lin, in_buf ? buf + buf_n : tt.lin + tt.lin_n);
                continue;
            }
            const uint64_t tl = P.circular ? 2ull * wl : wl;
            for (int k = 0; k < P.npat && !hit; ++k) {
                const RegexProgram& p = P.regex[k];
                if (p.nullable) { hit = true; break; }
                const uint32_t nchunk = (p.npos + 7u) >> 3;
                uint64_t S = 0;
                hit = re_step(p, nchunk, S, RE_SYM_BEGIN);
                for (uint64_t x = 0; x < tl && !hit; ++x) {
                    uint32_t j = (uint32_t)(x >= wl ? x - wl : x) + wb;
                    const uint8_t c = strand == 0 ? T.at(j) : P.comp[T.at(L - 1u - j)];
                    hit = re_step(p, nchunk, S, c);
                }
                if (!hit) hit = re_step(p, nchunk, S, RE_SYM_END);
            }
        }
    } else {
        uint32_t off = 0, tl = hl;
        if (!P.by_name) tl = id_span_rec(t, i, h, hl, P.id_mode, &off, P.buf_end);
        for (int k = 0; k < P.npat && !hit; ++k) {
            const RegexProgram& p = P.regex[k];
            if (p.nullable) { hit = true; break; }
            const uint32_t nchunk = (p.npos + 7u) >> 3;
            uint64_t S = 0;
            hit = re_step(p, nchunk, S, RE_SYM_BEGIN);
            for (uint32_t x = 0; x < tl && !hit; ++x) hit = re_step(p, nchunk, S, h[off + x]);
            if (!hit) hit = re_step(p, nchunk, S, RE_SYM_END);
        }
    }
    const bool sel = P.invert ? !hit : hit;
    out_len[i] = sel ? format_len(hl, t.l_seq[i], P.fastq, P.line_width) : 0u;
}

// -r through the thread-list matcher (regex_vm.hpp): the expressions the bit-parallel automaton does not take -- word
// boundaries, more than 64 positions.  One lane per record, thread lists in private memory: a rare path, not a fast one;
// the same targets, strands, region and circular doubling as k_grep_regex.
__global__ __launch_bounds__(64) void k_grep_vm(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt, GrepParams P,
                                                uint32_t* __restrict__ out_len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint32_t lh = t.l_head[i];
    const uint8_t* h = buf + t.start[i] + 1;
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    bool hit = false;
    uint32_t caps[4];
    if (P.by_seq) {
        const Text T = text_of(buf, t, tt, i);
        const uint32_t L = T.L;
        const int nstr = P.strand_only == 1 ? 1 : (P.both_strands ? 2 : 1);
        const int str0 = P.strand_only == 2 ? 1 : 0;
        for (int strand = str0; strand < nstr && !hit; ++strand) {
            uint32_t wb = 0, we = L;
            if (P.region_on) sub_location(L, P.region_start, P.region_end, &wb, &we);
            const uint32_t wl = we - wb;
            const uint64_t tl64 = P.circular ? 2ull * wl : wl;
            const uint32_t tl = tl64 > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)tl64;
            auto at = [&](uint32_t x) -> uint8_t {
                const uint32_t j = (x >= wl ? x - wl : x) + wb;
                return strand == 0 ? T.at(j) : P.comp[T.at(L - 1u - j)];
            };
            for (int k = 0; k < P.npat && !hit; ++k) hit = vm_search_fn(P.vm[k], at, tl, 0u, caps);
        }
    } else {
        uint32_t off = 0, tl = hl;
        if (!P.by_name) tl = id_span_rec(t, i, h, hl, P.id_mode, &off, P.buf_end);
        for (int k = 0; k < P.npat && !hit; ++k) hit = vm_search(P.vm[k], h + off, tl, 0u, caps);
    }
    const bool sel = P.invert ? !hit : hit;
    out_len[i] = sel ? format_len(hl, t.l_seq[i], P.fastq, P.line_width) : 0u;
}

__global__ __launch_bounds__(256) void k_grep_name(const uint8_t* __restrict__ buf, RecordTable t, GrepParams P,
                                                   uint32_t* __restrict__ out_len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint64_t s = t.start[i];
    const uint32_t lh = t.l_head[i];
    const uint8_t* h = buf + s + 1;
    uint32_t hl = lh > 0 ? lh - 1 : 0, off = 0;
    uint32_t tl = hl;
    if (!P.by_name) tl = id_span_rec(t, i, h, hl, P.id_mode, &off, P.buf_end);
    bool hit = false;
    if (P.set_keys) {  // pattern set: probe by hash, verify by bytes (patterns[k] is a map key in the reference, grep.go:501-511)
        const uint64_t key = fnv1a64(h + off, tl, P.ignore_case);
        for (uint64_t slot = key & P.set_mask;; slot = (slot + 1) & P.set_mask) {
            const uint64_t sk = P.set_keys[slot];
            if (sk == 0) break;
            if (sk != key) continue;
            const uint32_t k = P.set_idx[slot];
            const uint8_t* pp = P.pat + P.pat_off[k];
            if (P.pat_off[k + 1] - P.pat_off[k] != tl) continue;
            bool ok = true;
            for (uint32_t q = 0; q < tl; ++q) {
                uint8_t c = h[off + q];
                if (P.ignore_case) c = lower8(c);
                if (c != pp[q]) { ok = false; break; }
            }
            if (ok) { hit = true; break; }
        }
    }
    for (int k = 0; !P.set_keys && k < P.npat && !hit; ++k) {
        const uint8_t* pp = P.pat + P.pat_off[k];
        const uint32_t m = P.pat_off[k + 1] - P.pat_off[k];
        if (m != tl) continue;
        bool ok = true;
        for (uint32_t q = 0; q < m; ++q) {
            uint8_t c = h[off + q];
            if (P.ignore_case) c = lower8(c);
            if (c != pp[q]) { ok = false; break; }
        }
        hit = ok;
    }
    const bool sel = P.invert ? !hit : hit;
    out_len[i] = sel ? format_len(hl, t.l_seq[i], P.fastq, P.line_width) : 0u;
}

}  // namespace

hipError_t launch_grep_match(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const TextTableH* tt,
                             const GrepParams& Pin, uint32_t* out_len, hipStream_t st, uint64_t avg_record_bytes) {
    if (t.n == 0) return hipSuccess;
    GrepParams P = Pin;
    P.buf_end = buf + buf_n;
    if (P.vm) {
        TextTable d{tt ? tt->text_w : nullptr, tt ? tt->lin_off : nullptr, tt ? tt->lin : nullptr, tt ? tt->lin_n : 0};
        hipLaunchKernelGGL(k_grep_vm, dim3((unsigned)((t.n + 63) / 64)), dim3(64), 0, st, buf, t, d, P, out_len);
    } else if (P.regex) {
        TextTable d{tt ? tt->text_w : nullptr, tt ? tt->lin_off : nullptr, tt ? tt->lin : nullptr, tt ? tt->lin_n : 0};
        hipLaunchKernelGGL(k_grep_regex, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, t, d, P, out_len, buf_n);
    } else if (P.by_seq && P.general) {
        TextTable d{tt ? tt->text_w : nullptr, tt ? tt->lin_off : nullptr, tt ? tt->lin : nullptr, tt ? tt->lin_n : 0};
        if (Pin.sa_ok) {  // (the caller clears sa_ok for the context's switch grep_shiftand = off)
            uint64_t blocks = (t.n + 255) / 256;
            if (blocks > 256ull * 8ull) blocks = 256ull * 8ull;
            hipLaunchKernelGGL(k_grep_shiftand, dim3((unsigned)blocks), dim3(256), 0, st, buf, buf_n, t, d, P, out_len);
            return hipGetLastError();
        }
        const uint64_t blocks = (t.n * GROUP + 255) / 256;
        hipLaunchKernelGGL(k_grep_seq_gen, dim3((unsigned)blocks), dim3(256), 0, st, buf, t, d, P, out_len);
    } else if (P.by_seq) {
        TextTable d{tt ? tt->text_w : nullptr, tt ? tt->lin_off : nullptr, tt ? tt->lin : nullptr, tt ? tt->lin_n : 0};
        const uint64_t avg = avg_record_bytes ? avg_record_bytes : buf_n / t.n;  // bytes per record (a filtered table holds few of them)
        if (avg < 1024) {
            hipLaunchKernelGGL((k_grep_seq<BSK_GREP_LANES, false>), dim3((unsigned)((t.n * BSK_GREP_LANES + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, d, P, out_len);
        } else {
            hipLaunchKernelGGL((k_grep_seq<16, false>), dim3((unsigned)((t.n * 16 + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, d, P, out_len);
        }
        if (P.long_thresh && P.long_count) {
            const unsigned chunks = (unsigned)(((uint64_t)P.long_max * (P.circular ? 2 : 1) + GREP_LONG_CH - 1) / GREP_LONG_CH);
            hipLaunchKernelGGL((k_grep_seq<16, true>), dim3(chunks, (unsigned)P.long_count), dim3(256), 0, st, buf, buf_n, t, d, P, out_len);
            hipLaunchKernelGGL(k_grep_long_finish, dim3((unsigned)((P.long_count + 255) / 256)), dim3(256), 0, st, t, P, out_len);
        }
    } else {
        const uint64_t blocks = (t.n + 255) / 256;
        hipLaunchKernelGGL(k_grep_name, dim3((unsigned)blocks), dim3(256), 0, st, buf, t, P, out_len);
    }
    return hipGetLastError();
}

}  // namespace bsk
