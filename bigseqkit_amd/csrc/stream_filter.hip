// ============================================================================
// stream_filter.hip -- the record table of ONLY the records whose sequence line holds one of a few
// exact patterns, built in the ONE streaming pass that finds the records (FASTQ, 4-line layout).
//
// `grep -s -p` (Grep.grepGeneral, /root/reference/bigseqkit-lib/grep.go:442-482: bytes.Contains(seq, p) on the
// sequence and, unless -P, on its reverse complement) and `locate -p` (locate.go:583-667, 679-766: bytes.Index in a
// loop) decide per record; round 1 built the table of ALL records (k_index: a full pass that also writes 24 B per
// record) and then ran a per-record search over the same text.  Here the search runs on the tile while it sits in
// registers, and only the selected records (hit XOR invert) reach the table; the per-record kernels that follow
// (k_grep_seq / k_locate: output sizes, rows, coordinates) then see 2 % of the shard instead of all of it.
//
// Search (per 16-byte piece, every lane): an occurrence of a pattern of m >= 11 bytes contains two CONSECUTIVE dwords
// of the tile grid, whatever its alignment (the first one starts at most 3 bytes into the pattern).  Every (pattern,
// strand, alignment j in 0..3) is one ENTRY e < 32 with the pair (pat[j..j+4), pat[j+4..j+8)).  A dword is reduced to
// an 8-bit code -- bits 1 and 2 of each byte, which tell A, C, G, T apart and ignore case: one v_and and one
// v_dot4_u32_u8 with the weights (2, 8, 32, 128) give 4 x code, the byte offset into two 256-entry tables in LDS:
// T1[code] = set of entries whose FIRST dword has this code, T2[code] = ... SECOND dword.  A pair can only be an entry's
// pair if  T1[code(first)] & T2[code(second)] != 0  -- per dword: and, dot4, two LDS reads, one v_and_or (the mask of
// the dword before a lane's first one comes over DPP wave_shr:1, across pieces and tiles in a scalar).  No multiply
// (v_mul_lo_u32 / v_mad_u64_u32 are quarter rate on gfx950: the first version of this kernel, with a multiplicative
// pair hash, ran at 5.6 ms per 12.5 GB against 3.0 ms for the plain index pass).
// A flagged piece (exact match of 8 bases in the 2-bit alphabet: 8 entries x 4^-8 per dword of random text) is
// looked at again out of line: its pairs are re-read, the entries in the mask verified byte by byte against the pattern,
// and the start of a verified occurrence joins a per-wave list of pending hits; the record-end event of the skeleton
// asks "is there a pending hit inside my sequence line".  Lists that overflow (a pattern that matches nearly
// everywhere) raise ERR_FILTER_OVERFLOW and the host falls back to the record-table path -- never a different answer.
// HBM-bound byte work; no MFMA.
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

#include "index.hpp"
#include "anchor.hpp"
#include "stream_core_dev.hpp"
#include "stream_filter.hpp"

namespace bsk {

namespace {

using namespace stream;

constexpr int HITCAP = 64;  // pending hits per wave (between two record ends)

__device__ __forceinline__ uint32_t fold4(uint32_t x) {  // ASCII lower-case, 4 bytes at once (as ops_grep.hip)
    const uint32_t ge_a = (x & 0x7F7F7F7Fu) + 0x3F3F3F3Fu;
    const uint32_t ge_z1 = (x & 0x7F7F7F7Fu) + 0x25252525u;
    const uint32_t up = ge_a & ~ge_z1 & ~x & 0x80808080u;
    return x | (up >> 2);
}

// 4 x (c0 + 4 c1 + 16 c2 + 64 c3), c = bits 1..2 of a byte: the byte offset of a dword's code in T1 / T2, added to
// `base` by the instruction's accumulator (the LDS address of T1 costs no instruction of its own)
__device__ __forceinline__ uint32_t code_off(uint32_t w, uint32_t base = 0u) {
    return __builtin_amdgcn_udot4(w & 0x06060606u, 0x80200802u, base, false);
}

// ---- rare path of the filter (out of line, so that the streaming loop keeps its registers): one candidate per lane --
// a dword pair whose codes belong to at least one entry -- is verified against the full pattern.  The text is read in
// 16-byte pieces whose loads are all in flight together (an L2 hit each; a byte loop with one dependent load per letter
// cost this path 6 000 cycles per call and the whole kernel its gain); the patterns sit in LDS, zero-padded to 64 bytes.
// Verified start positions join the wave's pending list.  Returns the new list length, ~0 when the list is full.
__device__ __noinline__ uint32_t filter_verify(uint32_t nc, uint64_t tile_idx, uint64_t rs, uint64_t re, uint64_t n,
                                               const uint8_t* __restrict__ buf, const uint32_t* s_coff, const uint32_t* s_cmask,
                                               const uint16_t* s_ent, const uint32_t* s_pat, uint32_t* s_hits, uint32_t icase,
                                               uint32_t nh) {
    const int lane = threadIdx.x & 63;
    bool ok = false;
    uint32_t srel = 0;
    if ((uint32_t)lane < nc) {
        const uint32_t off = s_coff[lane];   // tile-relative offset of the SECOND dword of the pair
        uint32_t mask = s_cmask[lane];       // entries the pair may belong to
        while (mask && !ok) {
            const uint32_t e = (uint32_t)__ffs((int)mask) - 1u;
            mask &= mask - 1u;
            const uint32_t kj = s_ent[e];
            const uint32_t k = kj & 0x1Fu, j = (kj >> 5) & 3u, m = kj >> 8;
            const int64_t s_abs = (int64_t)tile_idx + off - 4 - (int64_t)j;  // the occurrence starts j bytes before the pair
            if (s_abs >= (int64_t)rs && (uint64_t)s_abs + m <= re) {
                const uint8_t* tp = buf + s_abs;
                const uint32_t* pp = s_pat + k * (FILTER_MAX_LEN / 4u);
                uint32_t diff = 0;
                if ((uint64_t)s_abs + 64u <= n) {
                    uint4 t[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if ((uint32_t)c * 16u < m) __builtin_memcpy(&t[c], tp + c * 16, 16);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if ((uint32_t)c * 16u < m) {
                            const uint32_t w[4] = {t[c].x, t[c].y, t[c].z, t[c].w};
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                const uint32_t q = (uint32_t)c * 16u + (uint32_t)d * 4u;  // bytes q .. q+3 of the occurrence
                                if (q < m) {
                                    const uint32_t left = m - q;
                                    const uint32_t bm = left >= 4u ? 0xFFFFFFFFu : ((1u << (8u * left)) - 1u);
                                    uint32_t x = w[d];
                                    if (icase) x = fold4(x);
                                    diff |= (x ^ pp[q >> 2]) & bm;
                                }
                            }
                        }
                    }
                } else {  // the last bytes of the shard: letter by letter
                    const uint8_t* pb = reinterpret_cast<const uint8_t*>(pp);
                    for (uint32_t q = 0; q < m; ++q) {
                        uint8_t ch = tp[q];
                        if (icase) ch = (ch >= 'A' && ch <= 'Z') ? (uint8_t)(ch + 32) : ch;
                        diff |= (uint32_t)(ch ^ pb[q]);
                    }
                }
                if (diff == 0) { ok = true; srel = (uint32_t)((uint64_t)s_abs - rs); }
            }
        }
    }
    const uint64_t bal = __ballot(ok);
    if (bal) {
        const uint32_t cnt = (uint32_t)__popcll(bal);
        if (nh + cnt > (uint32_t)HITCAP) return 0xFFFFFFFFu;
        if (ok) s_hits[nh + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = srel;
        nh += cnt;
    }
    wave_lds_fence();
    return nh;
}

struct FilterSink {
    static constexpr bool TILE_HOOK = true;
    IndexDev D;
    FilterDev F;
    const uint32_t* s_t1;    // LDS: [256] entries whose first dword has this code
    const uint32_t* s_t2;    // LDS: [256] entries whose second dword has this code
    const uint16_t* s_ent;   // LDS: [32] entry -> pattern index | alignment << 5 | pattern length << 8
    const uint32_t* s_pat;   // LDS: the patterns, zero-padded to FILTER_MAX_LEN bytes each
    uint32_t* s_coff;        // LDS: this wave's candidates of the current tile: offset of the pair's second dword ...
    uint32_t* s_cmask;       //      ... and the entries the pair may belong to
    uint64_t n_buf = 0;      // bytes in the shard (the verification may not read past it)
    uint32_t* s_hits;        // LDS: this wave's pending hit positions (range-relative start of the occurrence)
    uint32_t nh = 0;         // pending hits (wave-uniform)
    uint32_t carry = 0;      // T1 mask of the last dword before the current piece (wave-uniform)
    uint64_t base = 0, limit = 0;
    uint32_t nsel = 0;       // records selected so far in this range (wave-uniform)
    uint32_t err = 0;

    __device__ __forceinline__ void begin_range(uint64_t b, uint64_t lim) {
        base = b;
        limit = lim;
        nsel = 0;
        nh = 0;
        carry = 0;
    }

    // ---- the tile in registers (fast path: ~17 vector instructions and 4 LDS instructions per 16-byte piece) ---------
    __device__ __forceinline__ void tile(const uint4 (&cur)[NPIECE], uint64_t tile_idx, uint64_t rs, uint64_t re,
                                         const uint8_t* __restrict__ buf) {
        const int lane = threadIdx.x & 63;
        using lds_u32 = __attribute__((address_space(3))) const uint32_t;
        const uint32_t t1 = (uint32_t)(uintptr_t)(lds_u32*)s_t1;  // LDS byte address of T1; T2 follows it (k_filter)
        uint32_t nc = 0;  // candidates of this tile (wave-uniform)
#pragma unroll
        for (int p = 0; p < NPIECE; ++p) {
            const uint32_t o0 = code_off(cur[p].x, t1), o1 = code_off(cur[p].y, t1), o2 = code_off(cur[p].z, t1),
                           o3 = code_off(cur[p].w, t1);
            const uint32_t f0 = *(lds_u32*)(uintptr_t)o0, f1 = *(lds_u32*)(uintptr_t)o1, f2 = *(lds_u32*)(uintptr_t)o2,
                           f3 = *(lds_u32*)(uintptr_t)o3;
            const uint32_t g0 = ((lds_u32*)(uintptr_t)o0)[256], g1 = ((lds_u32*)(uintptr_t)o1)[256],
                           g2 = ((lds_u32*)(uintptr_t)o2)[256], g3 = ((lds_u32*)(uintptr_t)o3)[256];
            // lane l takes f3 of lane l - 1; lane 0 the last dword of the previous piece (previous tile for p == 0)
            const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)f3, 0x138, 0xf, 0xf, false);
            carry = (uint32_t)__builtin_amdgcn_readlane((int)f3, 63);
            const uint32_t td[4] = {pv & g0, f0 & g1, f1 & g2, f2 & g3};
            if (__ballot((td[0] | td[1] | td[2] | td[3]) != 0u)) {  // ~ every tenth piece of random text
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint64_t bal = __ballot(td[d] != 0u);
                    if (bal) {
                        const uint32_t idx = nc + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                        if (td[d] != 0u && idx < (uint32_t)WAVE) {
                            s_coff[idx] = (uint32_t)(p * PIECE_BYTES + lane * 16 + 4 * d);
                            s_cmask[idx] = td[d];
                        }
                        nc += (uint32_t)__popcll(bal);
                    }
                }
            }
        }
        if (nc) {
            if (nc > (uint32_t)WAVE) { err |= ERR_FILTER_OVERFLOW; return; }
            wave_lds_fence();
            const uint32_t r = filter_verify(nc, tile_idx, rs, re, n_buf, buf, s_coff, s_cmask, s_ent, s_pat, s_hits,
                                             (uint32_t)F.ignore_case, nh);
            if (r == 0xFFFFFFFFu) err |= ERR_FILTER_OVERFLOW;
            else nh = r;
        }
    }

    // ---- newline events: validation as k_index; the record-end event decides and writes ------------------------------
    template <bool FASTQ, bool ALL>
    __device__ __forceinline__ void batch(Lds<FASTQ, ALL>& L, uint32_t E, uint32_t wb, uint64_t tile_idx,
                                          uint32_t tile_rel, uint64_t re, const uint8_t* __restrict__ buf) {
        static_assert(FASTQ && !ALL, "the pattern filter runs on the sparse FASTQ path");
        const int lane = threadIdx.x & 63;
        for (uint32_t e0 = 0; e0 < E; e0 += WAVE) {
            const uint32_t e = e0 + lane;
            const bool on = e < E;
            const uint32_t s = HISTORY + (on ? e : 0);
            const uint32_t rank = wb + e;
            const uint32_t p = L.pos[s];
            const uint64_t abs_next = tile_idx + (uint64_t)(uint32_t)(p - tile_rel) + 1;
            const uint32_t role = rank & 3u;
            bool sel = false;
            uint32_t lh = 0, ls = 0, lp = 0;
            uint64_t start = 0;
            if (on) {
                if (role == 1u) {
                    if (next_char(L, s, abs_next, re, buf) != '+') err |= ERR_BAD_PLUS;
                } else if (role == 0u) {
                    if (next_char(L, s, abs_next, re, buf) == '+') err |= ERR_BAD_PLUS;
                } else if (role == 3u) {
                    const uint32_t p1 = L.pos[s - 1], p2 = L.pos[s - 2], p3 = L.pos[s - 3], p4 = L.pos[s - 4];
                    const uint32_t lq = p - p1 - 1u;
                    lp = p1 - p2 - 1u; ls = p2 - p3 - 1u; lh = p3 - p4 - 1u;
                    if (lq != ls) err |= ERR_LEN_MISMATCH;
                    if (abs_next < re && next_char(L, s, abs_next, re, buf) != '@') err |= ERR_BAD_HEADER;
                    start = abs_of(p4, tile_idx, tile_rel) + 1;
                    // a pending hit inside the sequence line (p3, p2): a pattern holds no newline, so an occurrence that
                    // starts inside the line ends inside it
                    bool hit = false;
                    for (uint32_t k = 0; k < nh; ++k) hit |= (s_hits[k] - p3 - 1u) < (p2 - p3 - 1u);
                    sel = hit != (F.invert != 0);
                }
            }
            const uint64_t bal = __ballot(sel);
            if (bal) {
                if (sel && D.write) {
                    const uint64_t g = base + nsel + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    if (g < limit) {
                        D.t.start[g] = start;
                        D.t.l_head[g] = lh;
                        D.t.l_seq[g] = ls;
                        D.t.aux[g] = lp;
                    } else {
                        err |= ERR_CAPACITY;
                    }
                }
                nsel += (uint32_t)__popcll(bal);
            }
        }
        // hits before the last record end seen so far can never be asked for again
        if (nh) {
            const uint32_t r = wb + E - 1u;                 // rank of the last event
            const uint32_t back = (r + 1u) & 3u;             // events after the last record end
            if (r >= back + 3u) {                            // (else: no record has ended yet in this range)
                const uint32_t cut = L.pos[(int)HISTORY + (int)E - 1 - (int)back];  // slot >= HISTORY - 3: kept history
                uint32_t h = 0;
                bool keep = false;
                if ((uint32_t)lane < nh) { h = s_hits[lane]; keep = (int32_t)(h - cut) > 0; }
                wave_lds_fence();
                const uint64_t kb = __ballot(keep);
                if (keep) s_hits[(uint32_t)__popcll(kb & ((1ull << lane) - 1ull))] = h;
                nh = (uint32_t)__popcll(kb);
                wave_lds_fence();
            }
        }
    }
};

// 4 waves per SIMD (128 VGPRs, 32 bytes of scratch) since round 3: at 5 (95 VGPRs) 34 registers were spilled around the
// out-of-line verification that nearly every tile calls -- the 3.3 GB of writes and the 1.44 x fetch that PMC showed for a
// 12.5 GB pass were scratch traffic.  k_filter 4.07 -> 3.72 ms, grep 4.76 -> 4.40 ms (scripts/r03_var.sh);
// 3 waves (no scratch at all): 4.42 ms.
#ifndef BSK_FILTER_WAVES
#define BSK_FILTER_WAVES 4
#endif
#if BSK_FILTER_WAVES
#define BSK_FILTER_ATTR __attribute__((amdgpu_waves_per_eu(BSK_FILTER_WAVES, 8)))
#else
#define BSK_FILTER_ATTR
#endif

template <bool DPP>
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) BSK_FILTER_ATTR void k_filter(const uint8_t* __restrict__ buf, uint64_t n,
                                                                     const uint64_t* __restrict__ anchors,
                                                                     uint32_t nranges, uint32_t* __restrict__ queue,
                                                                     IndexDev D, FilterDev F) {
    __shared__ Lds<true, false> s_l[WAVES_PER_BLOCK];
    __shared__ uint32_t s_t12[512];  // T1 ++ T2 (the fast path addresses T2 as T1 + 256 dwords)
    uint32_t* const s_t1 = s_t12;
    uint32_t* const s_t2 = s_t12 + 256;
    __shared__ uint16_t s_ent[FILTER_MAX_ENTRIES];
    __shared__ uint32_t s_hits[WAVES_PER_BLOCK][HITCAP];
    __shared__ uint32_t s_pat[FILTER_MAX_PATTERNS * FILTER_MAX_LEN / 4];
    __shared__ uint32_t s_cand[WAVES_PER_BLOCK][2 * WAVE];
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) { s_t1[i] = F.t1[i]; s_t2[i] = F.t1[256u + i]; }
    if (threadIdx.x < FILTER_MAX_ENTRIES) s_ent[threadIdx.x] = F.ent[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < FILTER_MAX_PATTERNS * FILTER_MAX_LEN / 4; i += blockDim.x) s_pat[i] = F.pat_padded[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Lds<true, false>& L = s_l[wave];
    FilterSink sink;
    sink.D = D;
    sink.F = F;
    sink.s_t1 = s_t1;
    sink.s_t2 = s_t2;
    sink.s_ent = s_ent;
    sink.s_pat = s_pat;
    sink.s_coff = s_cand[wave];
    sink.s_cmask = s_cand[wave] + WAVE;
    sink.n_buf = n;
    sink.s_hits = s_hits[wave];
    PredConsts P;  // unused (sparse path)
    P.k20 = P.k30 = 0;
    P.ngap = 0;
    const uint64_t n_eff = anchors[nranges];
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(queue, 1u);
        r = wave_first(r);
        if (r >= nranges) break;
        uint64_t rs = anchors[r], re = anchors[r + 1];
        rs = rs < n_eff ? rs : n_eff;
        re = re < n_eff ? re : n_eff;
        if (rs >= re) {
            if (D.write != 1 && lane == 0) D.range_count[r] = 0;
            continue;
        }
        uint64_t b = 0, lim = D.t.cap;
        if (D.write == 1) b = D.range_base[r];
        else if (D.write == 2) { b = (uint64_t)r * D.sparse_cap; lim = b + D.sparse_cap; if (lim > D.t.cap) lim = D.t.cap; }
        sink.begin_range(b, lim);
        stream_range<true, false, DPP>(L, buf, n, rs, re, re == n_eff, P, sink);
        if (D.write != 1 && lane == 0) D.range_count[r] = (uint64_t)sink.nsel;
    }
    const uint32_t err = wave_or_u32(sink.err);
    if (lane == 0 && err) atomicOr((unsigned long long*)&D.status[0], (unsigned long long)err);
}

}  // namespace

hipError_t launch_filter(bool dpp, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors, uint32_t nranges,
                         uint32_t* queue, const IndexDev& D, const FilterDev& F, hipStream_t st) {
    const dim3 b(WAVES_PER_BLOCK * WAVE);
    if (dpp) hipLaunchKernelGGL((k_filter<true>), dim3(blocks), b, 0, st, buf, n, anchors, nranges, queue, D, F);
    else hipLaunchKernelGGL((k_filter<false>), dim3(blocks), b, 0, st, buf, n, anchors, nranges, queue, D, F);
    return hipGetLastError();
}

int filter_max_blocks_per_cu(bool dpp) {
    int nb = 0;
    const void* f = dpp ? (const void*)k_filter<true> : (const void*)k_filter<false>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, WAVES_PER_BLOCK * WAVE, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

}  // namespace bsk
