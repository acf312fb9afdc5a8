// ============================================================================
// stream_fasta2_dev.hpp -- the FASTA streaming pass that publishes only the newlines a record-level sink acts on
// (device code only; round 5).
//
// SeqParser.Read on FASTA (/root/reference/bigseqkit-lib/helper.go:271-283) joins the lines of a record; what
// Stats.Call needs of it (stats.go:88) is len(seq) = (bytes between the header's newline and the newline in front of the
// next '>') - (newlines in between).  stream_range<FASTA> (stream_core_dev.hpp) publishes an LDS event for EVERY newline --
// 67 per 4 KiB tile of 60-column text, two compaction rounds and two sink calls per tile -- although the sink acts on two
// kinds only: the newline in front of a '>' (a record closes) and the newline behind a header line (its sequence begins),
// and already measures a record as position minus newline index (key = p - rank).  Here:
//   * the 16-byte pieces that hold a byte below 0x20 (the prefilter of the sparse path) are QUEUED in LDS with their
//     position, across tiles; a round runs when 64 wait -- every round is full (1.05 per tile instead of 2);
//   * a round computes exact newline masks, ranks by one wave scan, and for every newline "is the next byte '>'" from the
//     piece itself (the byte behind the piece travels with it); "ends a header line" = the newline before it closed a
//     record -- one ballot pair across the lanes, one carried bit across rounds;
//   * only those two kinds become events (key = position - rank, two flag bits) in a second LDS ring; the sink takes 64 at
//     a time -- once per ~8 tiles of 1 kb records, ~40 tiles of 5 kb records;
//   * what the range leaves open for the stitch kernel (last newline, its flag) is carried in scalars.
// The sink interface:  sink.events2(L, first, E)  -- E <= 64 events at ring positions first, first + 1, ... (mod F2_EVENTS).
// HBM-bound byte work; no MFMA.
// ============================================================================
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "stream_core_dev.hpp"

namespace bsk {
namespace stream {

constexpr uint32_t F2_QUEUE = 128;   // flagged pieces that wait for a full round (ring)
constexpr uint32_t F2_EVENTS = 128;  // published events that wait for the sink (ring)
constexpr uint32_t F2_CLOSING = 1u, F2_HDR_END = 2u;

// GAPS (`stats -a`): the events also carry their position -- the sink looks at the bytes of every header line
template <bool GAPS>
struct LdsF2T {
    __attribute__((aligned(16))) uint4 qdata[F2_QUEUE];
    int32_t qpos[F2_QUEUE];    // position of the piece's first byte relative to the range start (negative: the piece begins before it)
    uint16_t qnx[F2_QUEUE];    // 0x100 | the byte behind the piece, 0 = unknown (memory is asked)
    uint32_t ekey[F2_EVENTS];  // position - rank of a published newline = bytes of the range before it that are no newlines
    uint32_t epos[GAPS ? F2_EVENTS : 1];  // its position
    uint8_t eflag[F2_EVENTS];  // F2_CLOSING | F2_HDR_END
};
using LdsF2 = LdsF2T<false>;

// what a range leaves behind for the sink's end_range() / the stitch kernel
struct F2Tail {
    uint32_t lines;      // newlines of the range (the virtual one of a file without a final newline included)
    uint32_t last_key;   // position - rank of the last newline
    bool last_closing;   // ... which closed a record (a '>' or the end of the shard follows it)
    bool last_hdr;       // ... which ended a header line
};

#ifndef BSK_F2_NT
#define BSK_F2_NT 1  // non-temporal tile loads: FASTA-5k 50 GB 9.72 -> 9.22 ms, FASTA-1k 20 GB 3.65 -> 3.63 (scripts/history/r05_d.sh)
#endif

// GAPS (`stats -a`): the gap letters of P among ALL bytes of the range are counted into sink.rgap (a uint32_t per lane; the
// caller folds it after every range) -- pieces without a byte below 0x20 by the "all sixteen above the largest gap letter"
// test of the FASTQ line-role path, queued pieces (and every piece of an edge tile) exactly, clipped to the range, in their
// round.  The gap count of the SEQUENCES is that sum minus the gap letters of the header lines, which the sink counts at the
// header-end events (their positions travel with the events).  The pieces in [skip_from, count_resume) -- the middle of a
// line longer than a chunk, whose chunks' own ranges count them (k_stats, RF_MID) -- are left out, as on the dense path.
template <bool DPP, bool GAPS = false, class Sink>
__device__ __forceinline__ F2Tail stream_range_fasta2(LdsF2T<GAPS>& L, const uint8_t* __restrict__ buf, uint64_t n, uint64_t rs, uint64_t re,
                                                      bool is_last, Sink& sink, const PredConsts& PC, uint64_t skip_from = ~0ull,
                                                      uint64_t count_resume = 0) {
    const uint32_t lane = threadIdx.x & 63u;
    const int32_t end_rel = (int32_t)(uint32_t)(re - rs);
    if (re - rs > 0x7FFFFFFFull) sink.err |= ERR_LINE_TOO_LONG;  // (positions are 32-bit and relative to the range)
    if (lane == 0 && rs == 0 && buf[rs] != '>') sink.err |= ERR_BAD_HEADER;
    uint32_t qhead = 0, qcnt = 0, ehead = 0, ecnt = 0;  // wave-uniform ring states
    uint32_t line_base = 0;                             // newlines seen so far (all of them, published or not)
    uint32_t carry_cl = buf[rs] == '>' ? 1u : 0u;       // "the newline before the next one closed a record": a range that begins with a header
    uint32_t last_pos = 0;                              // position of the last newline seen
    uint32_t carry_hd = 0;                              // ... which ended a header line
#if BSK_NL_SGPR
    uint32_t k_ctl;
    asm volatile("s_mov_b32 %0, 0x20202020" : "=s"(k_ctl));
#else
    const uint32_t k_ctl = 0x20202020u;
#endif
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64u - lane));
    // GAPS: the successor of the largest gap letter in every byte (kgap = 0x80 - (top + 1) per byte: capi.cpp)
    const uint32_t k_low = GAPS ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(0x80808080u - PC.kgap)) : 0u;

    auto run_sink = [&](uint32_t E) {
        wave_lds_fence();
        sink.events2(L, ehead, E);
        ehead = (ehead + E) & (F2_EVENTS - 1u);
        ecnt -= E;
    };

    // one round over the first m (<= 64) queued pieces
    auto round = [&](uint32_t m) {
        wave_lds_fence();
        const bool have = lane < m;
        const uint32_t qi = (qhead + lane) & (F2_QUEUE - 1u);
        uint4 v = make_uint4(0, 0, 0, 0);
        int32_t prel = 0;
        uint32_t nxw = 0;
        if (have) { v = L.qdata[qi]; prel = L.qpos[qi]; nxw = L.qnx[qi]; }
        uint32_t nl = have ? eq_mask16(v, 0x0A0A0A0Au) : 0u;
        const uint32_t nl_all = nl;  // (before the clip below)
        if (__ballot(have && (prel < 0 || prel + 16 > end_rel)) != 0ull) {  // pieces across the ends of the range
            int32_t lo = -prel, hi = end_rel - prel;
            lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
            hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
            nl &= hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
        }
        if constexpr (GAPS) {
            // the gap letters of the queued pieces, exactly (flags in the layout of pack_flags: 13 instructions per letter
            // instead of 31 for a byte-order mask), clipped to the range on the rounds that hold a piece across its ends
            uint32_t vp = have ? 0x0F0F0F0Fu : 0u;
            if (__ballot(have && (prel < 0 || prel + 16 > end_rel)) != 0ull) {
                int32_t lo = -prel, hi = end_rel - prel;
                lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
                hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
                vp = have ? packed_from_mask16(hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u) : 0u;
            }
            const uint64_t I = rs + (uint64_t)(int64_t)prel;
            if (I >= skip_from && I < count_resume) vp = 0u;
            // the queued pieces hold line feeds, so "sixteen bytes above the largest gap letter" never holds for them -- but
            // a piece whose bytes at or below that letter are ALL line feeds holds no gap letter either: sequence lines,
            // always, and header lines without blanks (a genome's few headers: no round pays for the letters)
            const uint32_t kg = PC.kgap;
            const uint32_t ca = popc4((((v.x & 0x7F7F7F7Fu) + kg) | v.x) & 0x80808080u, (((v.y & 0x7F7F7F7Fu) + kg) | v.y) & 0x80808080u,
                                      (((v.z & 0x7F7F7F7Fu) + kg) | v.z) & 0x80808080u, (((v.w & 0x7F7F7F7Fu) + kg) | v.w) & 0x80808080u);
            const bool cand = have && vp != 0u && (kg == 0xFFFFFFFFu || 16u - ca != (uint32_t)__popc(nl_all));
            if (__ballot(cand) != 0ull) {
#pragma nounroll
                for (int k = 0; k < PC.ngap; ++k) {
                    const uint32_t rep = PC.gap_rep[k];
                    sink.rgap += (uint32_t)__popc(pack_flags(zero_bytes(v.x ^ rep), zero_bytes(v.y ^ rep), zero_bytes(v.z ^ rep), zero_bytes(v.w ^ rep)) & vp);
                }
            }
        }
        const uint32_t cnt = (uint32_t)__popc(nl);
        const uint32_t incl = wave_incl_scan<DPP>(cnt);
        const uint32_t rank0 = line_base + incl - cnt;
        line_base += wave_last(incl);
        // which of this lane's newlines close a record (bit k = its k-th newline)
        uint32_t clmask = 0;
        {
            uint32_t mm = nl, k = 0;
            while (mm) {
                const uint32_t b = (uint32_t)__ffs((int)mm) - 1u;
                mm &= mm - 1u;
                uint32_t nc16 = 0;
                if (b < 15u) {
                    const uint32_t d = (b + 1u) >> 2, sh8 = ((b + 1u) & 3u) * 8u;
                    const uint32_t wd = d == 0 ? v.x : (d == 1 ? v.y : (d == 2 ? v.z : v.w));
                    nc16 = 0x100u | ((wd >> sh8) & 0xFFu);
                } else {
                    nc16 = nxw;
                }
                const uint64_t an = rs + (uint64_t)(int64_t)(prel + (int32_t)b) + 1ull;  // the byte after the newline
                const uint8_t nc = nc16 ? (uint8_t)nc16 : (an < n ? buf[an] : (uint8_t)0);
                const bool closing = nc == '>' || an >= n || (an >= re && is_last);
                clmask |= closing ? (1u << k) : 0u;
                ++k;
            }
        }
        const bool has = cnt != 0u;
        const uint32_t lastcl = has ? ((clmask >> (cnt - 1u)) & 1u) : 0u;
        const uint64_t bc = __ballot(has), bl = __ballot(lastcl != 0u);
        if (bc != 0ull) {
            const uint64_t lower = bc & below;
            const uint32_t prevcl = lower ? (uint32_t)((bl >> (63 - __clzll((long long)lower))) & 1ull) : carry_cl;
            // the k-th newline ends a header line iff the newline before it closed a record
            const uint32_t hdmask = ((clmask << 1) | prevcl) & ((1u << cnt) - 1u);
            const uint32_t pub = clmask | hdmask;
            const int top = 63 - __clzll((long long)bc);  // the lane of the last newline of this round
            carry_cl = (uint32_t)((bl >> top) & 1ull);
            carry_hd = ((uint32_t)__builtin_amdgcn_readlane((int)(hdmask >> ((cnt - 1u) & 31u)), top)) & 1u;
            {
                const uint32_t hb = 31u - (uint32_t)__clz((int)(nl | 1u));
                last_pos = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)prel + hb), top);
            }
            const uint32_t npub = (uint32_t)__popc(pub);
            if (__ballot(npub != 0u) != 0ull) {
                const uint32_t pincl = wave_incl_scan<DPP>(npub);
                const uint32_t total = wave_last(pincl);
                const uint32_t e0 = pincl - npub;  // index of this lane's first published event among the round's
                uint32_t done = 0;
                while (done < total) {  // (wave-uniform; one turn unless the lines are a few bytes long)
                    const uint32_t room = F2_EVENTS - ecnt, take = total - done < room ? total - done : room;
                    uint32_t mm = nl, k = 0, j = 0;
                    while (mm) {
                        const uint32_t b = (uint32_t)__ffs((int)mm) - 1u;
                        mm &= mm - 1u;
                        if ((pub >> k) & 1u) {
                            const uint32_t g = e0 + j;
                            if (g >= done && g < done + take) {
                                const uint32_t ei = (ehead + ecnt + (g - done)) & (F2_EVENTS - 1u);
                                L.ekey[ei] = ((uint32_t)prel + b) - (rank0 + k);
                                if constexpr (GAPS) L.epos[ei] = (uint32_t)prel + b;
                                L.eflag[ei] = (uint8_t)((((clmask >> k) & 1u) ? F2_CLOSING : 0u) | (((hdmask >> k) & 1u) ? F2_HDR_END : 0u));
                            }
                            ++j;
                        }
                        ++k;
                    }
                    ecnt += take;
                    done += take;
                    while (ecnt >= 64u) run_sink(64u);
                }
            }
        }
        qhead = (qhead + m) & (F2_QUEUE - 1u);
        qcnt -= m;
    };

    const uint64_t idx0 = rs & ~(uint64_t)15;
    const uint64_t ntiles = (re - idx0 + TILE - 1) / TILE;
    uint4 cur[NPIECE];
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) cur[p] = load16_tile<BSK_F2_NT != 0>(buf, n, idx0 + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16);

    for (uint64_t t = 0; t < ntiles; ++t) {
        const uint64_t tile_idx = idx0 + t * TILE;
        const int32_t tile_rel = (int32_t)(int64_t)(tile_idx - rs);  // (negative for the first tile of a range that begins off a 16-byte boundary)
        const bool edge = (tile_idx < rs) || (tile_idx + TILE > re);    // wave-uniform
#pragma unroll
        for (int p = 0; p < NPIECE; ++p) {
            const uint4 v = cur[p];
            const uint32_t h = (((v.x - k_ctl) & ~v.x) | ((v.y - k_ctl) & ~v.y) | ((v.z - k_ctl) & ~v.z) | ((v.w - k_ctl) & ~v.w)) & 0x80808080u;
            const int32_t prel = tile_rel + p * PIECE_BYTES + (int32_t)lane * 16;
            bool f = h != 0u;
            if (edge) f = (f || GAPS) && prel + 16 > 0 && prel < end_rel;  // (GAPS: every piece of an edge tile is counted in a round, clipped)
            if constexpr (GAPS) {
                // a piece without a newline (nothing below 0x20): sixteen bytes above the largest gap letter hold no gap
                // letter -- sequence text, always; only a wave with a candidate looks at the letters
                // ("some byte at or below it": the borrow trick of the newline filter with the letter's successor -- exact as
                // a yes / no, two instructions per dword where the count of the bytes above takes five)
                const uint32_t lowb = (((v.x - k_low) & ~v.x) | ((v.y - k_low) & ~v.y) | ((v.z - k_low) & ~v.z) | ((v.w - k_low) & ~v.w)) & 0x80808080u;
                const uint64_t I = tile_idx + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16;
                const bool mine = !f && !edge && !(I >= skip_from && I < count_resume);
                if (__ballot(mine && (lowb != 0u || PC.kgap == 0xFFFFFFFFu)) != 0ull) {
                    uint32_t cg = 0;
#pragma nounroll
                    for (int k = 0; k < PC.ngap; ++k) {
                        const uint32_t rep = PC.gap_rep[k];
                        cg += popc4(zero_bytes(v.x ^ rep), zero_bytes(v.y ^ rep), zero_bytes(v.z ^ rep), zero_bytes(v.w ^ rep));
                    }
                    sink.rgap += mine ? cg : 0u;
                }
            }
            // the byte behind the piece: the first byte of the next lane's piece, for lane 63 of the next piece's lane 0
            const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x130, 0xf, 0xf, true) & 0xFFu;
            // (lane 63: lane 0 of the next piece -- a scalar read, outside any divergent branch; unknown behind the tile's last piece)
            const uint32_t nx63 = p + 1 < NPIECE ? (0x100u | ((uint32_t)__builtin_amdgcn_readlane((int)cur[p + 1 < NPIECE ? p + 1 : p].x, 0) & 0xFFu)) : 0u;
            const uint32_t nxw = lane == 63u ? nx63 : (0x100u | nx);
            const uint64_t bal = __ballot(f);
            if (bal != 0ull) {
                if (f) {
                    const uint32_t qi = (qhead + qcnt + (uint32_t)__popcll(bal & below)) & (F2_QUEUE - 1u);
                    L.qdata[qi] = v;
                    L.qpos[qi] = prel;
                    L.qnx[qi] = (uint16_t)nxw;
                }
                qcnt += (uint32_t)__popcll(bal);
                if (qcnt >= 64u) round(64u);
            }
        }
        if (t + 1 < ntiles) {
            // the newline-free stretch of a line that is longer than the range's nominal chunk: on to its last tile
            uint64_t tgt = ntiles - 1;
            if constexpr (GAPS) {  // (the bytes from count_resume on are this range's to count: the tile that holds them is the target)
                if (count_resume > idx0 && (count_resume - idx0) / TILE < tgt) tgt = (count_resume - idx0) / TILE;
            }
            if (tile_idx + TILE >= skip_from && t + 1 < tgt) {
                t = tgt - 1;
                const uint64_t tgt_idx = idx0 + tgt * TILE;
#pragma unroll
                for (int p = 0; p < NPIECE; ++p) cur[p] = load16_tile<BSK_F2_NT != 0>(buf, n, tgt_idx + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16);
                continue;
            }
#pragma unroll
            for (int p = 0; p < NPIECE; ++p)
                cur[p] = load16_tile<BSK_F2_NT != 0>(buf, n, tile_idx + TILE + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16);
        }
    }
    if (qcnt) round(qcnt);
    // a shard that does not end with a newline: one virtual newline behind its last byte
    if (is_last && buf[re - 1] != '\n') {
        if (lane == 0) {
            const uint32_t ei = (ehead + ecnt) & (F2_EVENTS - 1u);
            L.ekey[ei] = (uint32_t)end_rel - line_base;
            if constexpr (GAPS) L.epos[ei] = (uint32_t)end_rel;
            L.eflag[ei] = (uint8_t)(F2_CLOSING | (carry_cl ? F2_HDR_END : 0u));
        }
        ecnt += 1;
        last_pos = (uint32_t)end_rel;
        line_base += 1;
        carry_hd = carry_cl;
        carry_cl = 1u;
    }
    while (ecnt) run_sink(ecnt < 64u ? ecnt : 64u);
    F2Tail T;
    T.lines = line_base;
    T.last_key = last_pos - (line_base - 1u);
    T.last_closing = carry_cl != 0u;
    T.last_hdr = carry_hd != 0u;
    return T;
}

}  // namespace stream
}  // namespace bsk
