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
#include "tile_lds_dev.hpp"

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
// ---- the verification of candidates (round 6: everything from LDS, in line, no call) ----------------------------------------
// A candidate is a dword pair whose codes belong to at least one entry; the occurrence it would be part of is compared with
// the pattern.  What the A/Bs of this round showed (scripts/r06_ab3.sh, profiles/r06_filter_ab.txt; k_filter on the 12.5 GB
// shard of C3, a candidate in every third tile, two in three of them planted hits that no prefilter can take away):
//   3.35 ms  round 5: an out-of-line function that read the text again from memory
//   3.33     the text from an LDS copy of the tile instead          3.27  the function in line, one 16-byte piece per pattern
//   3.26     ... without 64-bit arithmetic per candidate            2.27  with the verification compiled OUT
// -- none of the rewrites of the verification's BODY moved anything, and compiling it out gained a third.  The compiler's
// resource report said why: with any out-of-line callee in the kernel -- also one that a few tiles in a thousand reach -- the
// kernel needs its own registers PLUS the callee's across the call: 116 - 131 VGPRs = 4 waves per SIMD, where the streaming
// loop alone needs 77 (6 waves).  The pass is bound by latency, i.e. by waves in flight.  So: no callee, and little LDS (a
// copy of the whole tile per wave, 16.7 KB per block, held the kernel at 4 blocks per CU just the same).  Every candidate is
// verified from LDS one PIECE (64 lanes x 16 bytes) at a time -- the piece is written there when it has a candidate (one
// ds_write_b128 per lane) behind a 64-byte CARRY of the piece before, so that an occurrence that began there is contiguous;
// an occurrence that runs on into the NEXT piece is put back on the candidate list and verified when that piece is here.
// No load from memory, no call.
// Returns the new length of the pending-hit list, ~0 when a list is full.  *ndef: candidates left for the next tile, written
// to the front of s_coff / s_cmask (offsets relative to the next tile: negative).
constexpr uint32_t FCARRY = 64;  // bytes of the tile before, in front of the LDS copy (>= FILTER_MAX_LEN)
static_assert(FCARRY >= FILTER_MAX_LEN, "an occurrence that began in the tile before must lie in the carry");
__device__ __forceinline__ uint32_t filter_verify_lds(uint32_t nc, uint64_t tile_idx, uint64_t rs, uint64_t re, uint32_t* s_coff,
                                                      uint32_t* s_cmask, const uint16_t* s_ent, uint32_t e16, uint32_t pat_lds,
                                                      uint32_t* s_hits, uint32_t icase, uint32_t nh, uint32_t tile_lds, bool carry_ok,
                                                      uint32_t max_m, uint32_t* ndef) {
    constexpr uint32_t SPAN = PIECE_BYTES;  // the unit that is verified at a time: ONE piece of the tile (64 lanes x 16 bytes)
    const int lane = threadIdx.x & 63;
    bool ok = false, lost = false;
    uint32_t srel = 0, dmask = 0;
    // everything tile-relative and 32 bits wide (wave-uniform): an occurrence must lie in [lo, hi) of the tile's coordinates
    const int32_t lo = rs > tile_idx ? (int32_t)(rs - tile_idx) : (tile_idx - rs < (uint64_t)FCARRY ? -(int32_t)(tile_idx - rs) : -(int32_t)FCARRY);
    const uint32_t hi = re - tile_idx < (uint64_t)SPAN ? (uint32_t)(re - tile_idx) : SPAN;
    const bool after_ok = re > tile_idx + SPAN;              // the range goes on behind this tile
    const uint32_t base_rel = (uint32_t)(tile_idx - rs);     // (mod 2^32) + tile offset = range-relative position
    int32_t off = 0;
    if ((uint32_t)lane < nc) {
        off = (int32_t)s_coff[lane];
        uint32_t mask = s_cmask[lane];
        do {
            const uint32_t e = (uint32_t)__ffs((int)mask) - 1u;
            mask &= mask - 1u;
            const uint32_t kj = s_ent[e];
            const uint32_t k = kj & 0x1Fu, j = (kj >> 5) & 3u, m = kj >> 8;
            const int32_t st = off - 4 - (int32_t)j;  // tile-relative start of the occurrence
            if (st < lo) continue;                    // begins before the range (or further back than a pattern is long)
            if (st < 0 && !carry_ok) { lost = true; continue; }  // (cannot happen: the tiles of a range follow each other)
            if (st + (int32_t)m > (int32_t)hi) {
                if (after_ok && hi == SPAN) dmask |= 1u << e;  // runs on into the next tile: verified there
                continue;
            }
            const uint32_t a = tile_lds + (uint32_t)st;  // (st >= -FCARRY: the carry lies in front of the tile)
            uint32_t diff = 0;
            if (max_m <= 16u) {  // (wave-uniform) one piece of text, pattern and byte mask of the entry from the block's table
                uint32_t w[4];
                tilelds::lds_ld128(a, w);  // (the buffer is padded: bytes behind m are masked)
                const uint4 pt = tilelds::lds_r128(e16 + 32u * e), bm = tilelds::lds_r128(e16 + 32u * e + 16u);
                if (icase) { w[0] = fold4(w[0]); w[1] = fold4(w[1]); w[2] = fold4(w[2]); w[3] = fold4(w[3]); }
                diff = ((w[0] ^ pt.x) & bm.x) | ((w[1] ^ pt.y) & bm.y) | ((w[2] ^ pt.z) & bm.z) | ((w[3] ^ pt.w) & bm.w);
            } else {
#pragma unroll 1
                for (uint32_t q = 0; q < m; q += 4u) {  // a dword at a time (17 .. 64 bytes)
                    const uint32_t left = m - q;
                    const uint32_t bmq = left >= 4u ? 0xFFFFFFFFu : ((1u << (8u * left)) - 1u);
                    uint32_t x, hi2;
                    tilelds::lds_ld64(a + q, x, hi2);
                    (void)hi2;
                    if (icase) x = fold4(x);
                    diff |= (x ^ tilelds::lds_r32(pat_lds + k * FILTER_MAX_LEN + q)) & bmq;
                }
            }
            if (diff == 0u) { ok = true; srel = base_rel + (uint32_t)st; }
        } while (mask && !ok);
    }
    // what runs on into the next tile goes to the front of the candidate list (every lane has read its own entry above)
    const uint64_t dbal = __ballot(dmask != 0u && !ok);
    const uint32_t nd = (uint32_t)__popcll(dbal);
    wave_lds_fence();
    if (dmask != 0u && !ok) {
        const uint32_t at = (uint32_t)__popcll(dbal & ((1ull << lane) - 1ull));
        s_coff[at] = (uint32_t)(off - (int32_t)SPAN);
        s_cmask[at] = dmask;
    }
    *ndef = nd;
    if (__ballot(lost)) return 0xFFFFFFFFu;
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
    uint64_t prev_piece = ~0ull - TILE;  // the piece whose last bytes are the carry in LDS
    uint32_t ndef = 0;       // candidates of the tile before whose occurrence runs on into this one (front of s_coff / s_cmask)
    uint32_t pat_lds = 0;    // LDS byte address of the padded patterns
    uint32_t e16 = 0;        // LDS byte address of the per-entry table {first 16 pattern bytes, byte mask of its length}: patterns of <= 16 bytes
    uint32_t max_m = FILTER_MAX_LEN;  // the longest pattern (<= 16: the one-piece verification)
    uint32_t tile_lds = 0;   // LDS byte address of this wave's tile buffer (TILE + 80 bytes): written only for tiles with a candidate
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
        ndef = 0;
        prev_piece = ~0ull - TILE;
    }

    // ---- the tile in registers (fast path: ~17 vector instructions and 4 LDS instructions per 16-byte piece) ---------
    __device__ __forceinline__ void tile(const uint4 (&cur)[NPIECE], uint64_t tile_idx, uint64_t rs, uint64_t re,
                                         const uint8_t* __restrict__ buf) {
#if defined(BSK_FILTER_DIAG) && BSK_FILTER_DIAG == 2
        return;  // MEASUREMENT ONLY (scripts/r06_ab3.sh): the pass without its search -- the floor of the skeleton + record sink
#endif
        const int lane = threadIdx.x & 63;
        using lds_u32 = __attribute__((address_space(3))) const uint32_t;
        const uint32_t t1 = (uint32_t)(uintptr_t)(lds_u32*)s_t1;  // LDS byte address of T1; T2 follows it (k_filter)
#pragma unroll
        for (int p = 0; p < NPIECE; ++p) {
            const uint64_t piece_idx = tile_idx + (uint64_t)(p * PIECE_BYTES);
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
            const bool cont = prev_piece + PIECE_BYTES == piece_idx;  // (wave-uniform) the carry in LDS is the end of the piece before
            uint32_t nc = cont ? ndef : 0u;  // candidates of this piece (wave-uniform); the first `ndef` came over from the piece before
            ndef = 0;
            if (__ballot((td[0] | td[1] | td[2] | td[3]) != 0u) || nc) {  // ~ every tenth piece of random text
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint64_t bal = __ballot(td[d] != 0u);
                    if (bal) {
                        const uint32_t idx = nc + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                        if (td[d] != 0u && idx < (uint32_t)WAVE) {
                            s_coff[idx] = (uint32_t)(lane * 16 + 4 * d);
                            s_cmask[idx] = td[d];
                        }
                        nc += (uint32_t)__popcll(bal);
                    }
                }
#if defined(BSK_FILTER_DIAG) && BSK_FILTER_DIAG == 1
                err |= nc >> 31;  // MEASUREMENT ONLY: candidates are found but never verified (wrong answers)
                nc = 0;
#endif
                if (nc > (uint32_t)WAVE) { err |= ERR_FILTER_OVERFLOW; nc = 0; }
                if (nc) {
                    tilelds::lds_w128(tile_lds + (uint32_t)lane * 16u, cur[p]);
                    wave_lds_fence();
                    uint32_t nd = 0;
                    const uint32_t r = filter_verify_lds(nc, piece_idx, rs, re, s_coff, s_cmask, s_ent, e16, pat_lds, s_hits, (uint32_t)F.ignore_case,
                                                         nh, tile_lds, cont, max_m, &nd);
                    if (r == 0xFFFFFFFFu) err |= ERR_FILTER_OVERFLOW;
                    else { nh = r; ndef = nd; }
                }
            }
            // the last FCARRY bytes of this piece stay in front of the next one's LDS copy (one ds_write_b128 by four lanes)
            if (lane >= (int)(WAVE - FCARRY / 16u)) tilelds::lds_w128(tile_lds - FCARRY + (uint32_t)(lane - (int)(WAVE - FCARRY / 16u)) * 16u, cur[p]);
            prev_piece = piece_idx;
        }
        (void)buf;
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

// Rounds 3 - 5 ran 4 waves per SIMD: 128 VGPRs, because the out-of-line verification's registers came on top of the loop's.
// Round 6 (no callee, a 1 KiB LDS copy per wave instead of 4 KiB; profiles/r06_filter_ab.txt, one visit): 5 waves (89 VGPRs,
// what the compiler takes unasked) k_filter 3.09 - 3.12 ms, 6 waves (80 VGPRs, two spilled) 2.87 - 2.92 ms, 7 waves (spills
// in the loop) 3.20 ms; round 5's kernel 3.35 ms on the same box.
#ifndef BSK_FILTER_WAVES
#define BSK_FILTER_WAVES 6
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
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[WAVES_PER_BLOCK][FCARRY + PIECE_BYTES + 16];  // carry ++ a tile with a candidate (+ 16: the 16-byte read of an occurrence that ends on the last byte)
    __shared__ uint32_t s_maxm;
    __shared__ __attribute__((aligned(16))) uint32_t s_e16[FILTER_MAX_ENTRIES][8];
    if (threadIdx.x == 0) {
        uint32_t mm = 0;
        for (uint32_t e = 0; e < FILTER_MAX_ENTRIES; ++e) { const uint32_t m = (uint32_t)F.ent[e] >> 8; mm = m > mm ? m : mm; }
        s_maxm = mm;
    }
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) { s_t1[i] = F.t1[i]; s_t2[i] = F.t1[256u + i]; }
    if (threadIdx.x < FILTER_MAX_ENTRIES) s_ent[threadIdx.x] = F.ent[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < FILTER_MAX_PATTERNS * FILTER_MAX_LEN / 4; i += blockDim.x) s_pat[i] = F.pat_padded[i];
    __syncthreads();
    if (threadIdx.x < FILTER_MAX_ENTRIES) {  // (after the barrier: s_ent and s_pat are in LDS)
        const uint32_t kj = s_ent[threadIdx.x], k = kj & 0x1Fu, m = kj >> 8;
        for (uint32_t d = 0; d < 4u; ++d) {
            s_e16[threadIdx.x][d] = s_pat[k * (FILTER_MAX_LEN / 4u) + d];
            s_e16[threadIdx.x][4u + d] = m >= 4u * d + 4u ? 0xFFFFFFFFu : (m <= 4u * d ? 0u : ((1u << (8u * (m - 4u * d))) - 1u));
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Lds<true, false>& L = s_l[wave];
    FilterSink sink;
    sink.e16 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&s_e16[0][0];
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
    sink.max_m = s_maxm;
    sink.tile_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s_tile[wave] + FCARRY;
    sink.pat_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)s_pat;
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
