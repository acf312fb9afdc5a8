// ============================================================================
// stream_subseq.hip -- `subseq -r a:b` on FASTQ inside the streaming pass.
//
// SubseqTransform.Call in region mode (/root/reference/bigseqkit-lib/subseq.go:167-191, subseqByRegion :314-317) prints
// EVERY record: head as read, Seq.SubSeq(start, end) of the bases and of the qualities, FASTQ forced to one line each.
// The output of a record is therefore four pieces that each depend on ONE line of the input only:
//     line 0  "@head\n"            the header line as it stands
//     line 1  bases[b, e) "\n"     (b, e) = SubLocation(len(line), start, end)
//     line 2  "+\n"
//     line 3  quals[b, e) "\n"     the same (b, e): SeqParser.Read demands len(qual) == len(seq) (helper.go:294-297)
// so the newline EVENT of a line is all a lane needs to emit that line's piece: one wave scan over the piece lengths of
// up to 64 events gives every piece its place, and the lanes copy 16 bytes at a time from the wave's tile in LDS
// (tile_lds_dev.hpp: the tile behind a 512-byte carry, so a line that began in the tile before is contiguous) -- the
// last 16 bytes of a piece overlap the ones before instead of a byte tail, their last byte forced to '\n'.
// No record table, no size / scan / emit passes (round 2: k_index + k_seq_size + scans + k_seq_emit with 4 lanes per
// 117-byte record read the shard 2.1 x and wrote 1.25 x the output).  Every range writes into its own slice of a
// scratch buffer sized from the shard head; k_names_compact gathers the slices in range order (= file order).  A slice
// that overflows raises ERR_CAPACITY and the caller takes the record-table path.  HBM-bound byte work; no MFMA.
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

// Round 6 tried a sixth wave per SIMD: the pass is bound by neither memory nor instruction issue (476 VALU per tile = 4.7 ms of
// issue under an 8.4 ms pass that moves 34 GB), 27.1 KB of LDS per block and 81 registers hold it at 5 waves; a 256-byte carry
// in front of the LDS tile (-DBSK_TILE_CARRY=256: 26.1 KB = 6 blocks per CU) and -DBSK_SUBSEQ_WAVES=6 (78 registers, no spill)
// give it the sixth.  One box: 9.06 / 9.20 -> 8.86 / 8.71 ms; another, five alternations: 8.44 +- 0.04 -> 8.53 +- 0.07 ms, and
// 9.03 in the bench run of a third (scripts/r06_ab6.sh).  Not a gain that survives a change of box: the round-5 shape stays.
#include "anchor.hpp"
#include "ops_seq.hpp"
#include "stream_core_dev.hpp"
#include "stream_subseq.hpp"
#include "tile_lds_dev.hpp"

#ifndef BSK_SUBSEQ_WAVES
#define BSK_SUBSEQ_WAVES 0
#endif
#if BSK_SUBSEQ_WAVES
#define BSK_SUBSEQ_ATTR __attribute__((amdgpu_waves_per_eu(BSK_SUBSEQ_WAVES, 8)))
#else
#define BSK_SUBSEQ_ATTR
#endif

namespace bsk {

namespace {

using namespace stream;
using namespace tilelds;

constexpr uint32_t SUB_TBUF = TBUF + 16;  // a short piece is fetched as 16 bytes from its start: up to 19 past the tile

// sub_location (ops_seq.hpp: Seq.SubSeq's 1-based inclusive region, negative = from the end) in 32-bit arithmetic: the
// lines of this pass are shorter than 2^31 bytes (the skeleton raises ERR_LINE_TOO_LONG beyond), so every intermediate of
// the 64-bit original fits once the negative positions are taken as distances from the end (-x <= L or the bound clamps)
__device__ __forceinline__ void sub_location32(uint32_t L, int start, int end, uint32_t* b, uint32_t* e) {
    *b = *e = 0;
    if (L == 0) return;
    uint32_t s, t;  // 1-based, inclusive
    if (start < 0) { const uint32_t d = 0u - (uint32_t)start; s = d > L ? 1u : L - d + 1u; }   // L + start + 1, at least 1
    else s = start == 0 ? 1u : (uint32_t)start;
    if (end < 0) { const uint32_t d = 0u - (uint32_t)end; if (d > L) return; t = L - d + 1u; }  // (L + end + 1 < 1: empty)
    else t = (uint32_t)end > L ? L : (uint32_t)end;
    if (s > L || t < 1u || s > t) return;
    *b = s - 1u;
    *e = t;
}

template <bool DPP>
struct SubseqSink {
    static constexpr bool TILE_HOOK = true;
    SubseqDev D;
    TileLds T;
    uint8_t* slice = nullptr;  // this range's output slice
    uint32_t cursor = 0;       // bytes written to it so far (wave-uniform)
    uint32_t nrec = 0;         // records closed in this range (wave-uniform)
    uint32_t err = 0;
    const uint8_t* lim = nullptr;  // one past the last byte of the shard

    __device__ __forceinline__ void begin_range(uint32_t r) {
        slice = D.slices + (uint64_t)r * D.slice_cap;
        cursor = 0;
        nrec = 0;
        T.reset();
    }

    template <class CUR>
    __device__ __forceinline__ void tile(const CUR& cur, uint64_t tile_idx, uint64_t rs, uint64_t re, const uint8_t* __restrict__ buf) {
        T.stage(cur, tile_idx);
    }

    template <bool FASTQ, bool ALL>
    __device__ __forceinline__ void batch(Lds<FASTQ, ALL>& L, uint32_t E, uint32_t wb, uint64_t tile_idx,
                                          uint32_t tile_rel, uint64_t re, const uint8_t* __restrict__ buf) {
        static_assert(FASTQ && !ALL, "the subseq sink runs on the sparse FASTQ path");
        const int lane = threadIdx.x & 63;
        for (uint32_t e0 = 0; e0 < E; e0 += WAVE) {
            const uint32_t e = e0 + lane;
            const bool on = e < E;
            const uint32_t s = HISTORY + (on ? e : 0);
            const uint32_t rank = wb + e;
            const uint32_t p = L.pos[s], prev = L.pos[s - 1];
            const uint64_t abs_next = tile_idx + (uint64_t)(uint32_t)(p - tile_rel) + 1;
            const uint32_t role = rank & 3u;
            const uint32_t ll = p - prev - 1u;  // bytes of the line without its newline
            uint32_t from = 0, cnt = 0;         // the piece: cnt bytes of the line from byte `from`, the last one a '\n'
            if (on) {
                // the structural validation of the stats / index kernels (strict 4-line FASTQ)
                if (role == 0u) {
                    if (next_char(L, s, abs_next, re, buf) == '+') err |= ERR_BAD_PLUS;
                    cnt = ll + 1u;
                } else if (role == 2u) {
                    cnt = 2u;
                } else {
                    if (role == 1u) {
                        if (next_char(L, s, abs_next, re, buf) != '+') err |= ERR_BAD_PLUS;
                    } else {
                        const uint32_t p2 = L.pos[s - 2], p3 = L.pos[s - 3];
                        if (ll != p2 - p3 - 1u) err |= ERR_LEN_MISMATCH;
                        if (abs_next < re && next_char(L, s, abs_next, re, buf) != '@') err |= ERR_BAD_HEADER;
                    }
                    uint32_t b, en;
                    sub_location32(ll, D.region_start, D.region_end, &b, &en);
                    from = b;
                    cnt = en - b + 1u;
                }
            }
            const uint32_t incl = wave_incl_scan<DPP>(cnt);
            const uint32_t tot = wave_last(incl);
            if (cnt) {
                const uint32_t at = cursor + incl - cnt;
                if ((uint64_t)at + cnt <= D.slice_cap) {
                    uint8_t* dst = slice + at;
                    if (role == 2u) {
                        const uint16_t pl = 0x0A2Bu;  // "+\n"
                        __builtin_memcpy(dst, &pl, 2);
                    } else {
                        const int32_t so = (int32_t)(prev + 1u + from - tile_rel);  // first byte of the piece, from the tile start
                        const bool in_lds = T.holds(tile_idx, so) && (uint32_t)(so + (int32_t)CARRY) + cnt <= CARRY + (uint32_t)TILE;
                        const uint8_t* gp = buf + (int64_t)tile_idx + (int64_t)so;
                        const uint32_t la = T.addr(so);
                        auto ld = [&](uint32_t o, uint32_t (&w)[4]) {
                            if (in_lds) {
                                lds_ld128(la + o, w);
                            } else if (gp + o + 16 <= lim) {
                                __builtin_memcpy(w, gp + o, 16);
                            } else {  // the last bytes of the shard
                                w[0] = w[1] = w[2] = w[3] = 0;
#pragma unroll
                                for (int i = 0; i < 16; ++i)
                                    if (gp + o + i < lim) w[i >> 2] |= (uint32_t)gp[o + i] << (8 * (i & 3));
                            }
                        };
                        uint32_t w[4];
                        if (cnt >= 16u) {
                            for (uint32_t i = 0; i + 16u < cnt; i += 16u) {
                                ld(i, w);
                                __builtin_memcpy(dst + i, w, 16);
                            }
                            ld(cnt - 16u, w);
                            w[3] = (w[3] & 0x00FFFFFFu) | 0x0A000000u;
                            __builtin_memcpy(dst + (cnt - 16u), w, 16);
                        } else {
                            ld(0u, w);
                            {   // byte cnt - 1 := '\n'
                                const uint32_t m = cnt - 1u, sh = (m & 3u) * 8u, d = m >> 2;
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    if (d == (uint32_t)q) w[q] = (w[q] & ~(0xFFu << sh)) | (0x0Au << sh);
                            }
                            uint32_t q = 0;  // dwords consumed
                            if (cnt & 8u) { __builtin_memcpy(dst, w, 8); q = 2; }
                            if (cnt & 4u) {
                                const uint32_t v = q ? w[2] : w[0];
                                __builtin_memcpy(dst + 4u * q, &v, 4);
                                q += 1;
                            }
                            if (cnt & 3u) {
                                const uint32_t v = q == 0 ? w[0] : q == 1 ? w[1] : q == 2 ? w[2] : w[3];
                                uint8_t* d2 = dst + 4u * q;
                                if (cnt & 2u) {
                                    const uint16_t h2 = (uint16_t)v;
                                    __builtin_memcpy(d2, &h2, 2);
                                    if (cnt & 1u) d2[2] = (uint8_t)(v >> 16);
                                } else {
                                    d2[0] = (uint8_t)v;
                                }
                            }
                        }
                    }
                } else {
                    err |= ERR_CAPACITY;
                }
            }
            cursor += tot;
            nrec += (uint32_t)__popcll(__ballot(on && role == 3u));
        }
    }
};

template <bool DPP>
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) BSK_SUBSEQ_ATTR
void k_subseq_stream(const uint8_t* __restrict__ buf, uint64_t n, const uint64_t* __restrict__ anchors, uint32_t nranges,
                     uint32_t* __restrict__ queue, SubseqDev D) {
    __shared__ Lds<true, false> s_l[WAVES_PER_BLOCK];
    __shared__ __attribute__((aligned(16))) uint8_t s_tb[WAVES_PER_BLOCK][SUB_TBUF];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Lds<true, false>& L = s_l[wave];
    SubseqSink<DPP> sink;
    sink.D = D;
    sink.lim = buf + n;
    sink.T.tb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s_tb[wave];
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
            if (lane == 0) { D.range_bytes[r] = 0; D.range_count[r] = 0; }
            continue;
        }
        sink.begin_range(r);
        stream_range<true, false, DPP>(L, buf, n, rs, re, re == n_eff, P, sink);
        if (lane == 0) { D.range_bytes[r] = sink.cursor; D.range_count[r] = sink.nrec; }
    }
    const uint32_t err = wave_or_u32(sink.err);
    if (lane == 0 && err) atomicOr((unsigned long long*)&D.status[0], (unsigned long long)err);
}

}  // namespace

hipError_t launch_subseq_stream(bool dpp, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors,
                                uint32_t nranges, uint32_t* queue, const SubseqDev& D, hipStream_t st) {
    const dim3 b(WAVES_PER_BLOCK * WAVE);
    if (dpp) hipLaunchKernelGGL((k_subseq_stream<true>), dim3(blocks), b, 0, st, buf, n, anchors, nranges, queue, D);
    else hipLaunchKernelGGL((k_subseq_stream<false>), dim3(blocks), b, 0, st, buf, n, anchors, nranges, queue, D);
    return hipGetLastError();
}

int subseq_stream_max_blocks_per_cu(bool dpp) {
    int nb = 0;
    const void* f = dpp ? (const void*)k_subseq_stream<true> : (const void*)k_subseq_stream<false>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, WAVES_PER_BLOCK * WAVE, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

}  // namespace bsk
