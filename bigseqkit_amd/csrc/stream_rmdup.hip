// ============================================================================
// stream_rmdup.hip -- RmDupPrepare (/root/reference/bigseqkit-lib/rmdup.go:43-90) for `-s` on FASTQ inside the
// streaming pass: the record table rows AND the keys of every sequence leave from the ONE pass that finds the records.
//
// Round 2 read a shard four times for `rmdup -s` (k_index, k_rmdup_hash, the byte verification, the emit).  Here the
// pass that parses the records also hashes them, from the tile it holds anyway:
//   * the sink's tile hook copies the wave's 4 KiB tile into LDS (four ds_write_b128 per lane), behind a CARRY of the
//     last 512 bytes of the tile before it, so that a sequence line which ends in this tile and began up to 512 bytes
//     before it is contiguous in LDS (longer / cut lines are read from global memory by the same code);
//   * the event that ends a sequence line (line index & 3 == 1) knows the line; in a group of 64 events those are
//     every fourth one, at most 16 -- one per QUAD of lanes.  The four lanes of a quad run XXH64's four accumulators
//     over the 32-byte stripes (lane k takes the word at 32 s + 8 k; unaligned 8-byte words come from three aligned
//     ds_read_b32 and two v_alignbyte), fold them with quad-permute DPP moves, and finish merge, tail and avalanche;
//     the words also feed the four chains of the second key k2 (hash_dev.hpp);
//   * keys and table rows go to the same per-range slices (k_rmdup_compact gathers both).
// 64-bit multiplies are three v_mad_u64_u32 / v_mul_lo_u32 at full VOP3 rate on gfx950
// (profiles/r02_valu_issue_rates_gfx950.txt).  HBM-bound byte work; no MFMA.
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

#include "anchor.hpp"
#include "hash_dev.hpp"
#include "index.hpp"
#include "stream_core_dev.hpp"
#include "stream_rmdup.hpp"
#include "tile_lds_dev.hpp"

namespace bsk {

namespace {

using namespace stream;
using namespace hashdev;

using namespace tilelds;

// value of lane K of this lane's quad (quad_perm DPP)
template <int K>
__device__ __forceinline__ uint32_t quad_bcast32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, false);
}
template <int K>
__device__ __forceinline__ uint64_t quad_bcast64(uint64_t v) {
    return ((uint64_t)quad_bcast32<K>((uint32_t)(v >> 32)) << 32) | quad_bcast32<K>((uint32_t)v);
}
// sum over the four lanes of a quad (every lane gets it)
__device__ __forceinline__ uint64_t quad_sum64(uint64_t v) {
    uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0xB1, 0xf, 0xf, false) << 32) |
                 (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    v += o;
    o = ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x4E, 0xf, 0xf, false) << 32) |
        (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x4E, 0xf, 0xf, false);           // quad_perm [2,3,0,1]
    return v + o;
}

// MODE 1: XXH64 + k2 (two-key decisions, the multi-GPU exchange); MODE 0: XXH64 alone -- a quarter of the hashing
// instructions and 8 bytes per record less (8.7 -> 7.5 ms per 25 GB, scripts/history/r04_rmstream.sh);
// MODE 2 (round 5, rmdup's default: the bytes of every duplicate are compared afterwards, ops_host_rmdup.cpp): the
// chain-free grouping key of hash_dev.hpp instead of XXH64
constexpr int MODE_K1 = 0, MODE_K1K2 = 1, MODE_GROUP = 2;
template <bool DPP, bool FOLD, int MODE>
struct RmdupSink {
    static constexpr bool K2 = MODE == MODE_K1K2;
    static constexpr bool TILE_HOOK = true;
    static constexpr bool TILE_NT = true;  // every byte once: non-temporal tile loads (stream_core_dev.hpp)
    IndexDev D;
    HashDev H;
    TileLds T;                  // this wave's tile in LDS (tile_lds_dev.hpp)
    uint32_t gk = 0;            // MODE_GROUP: LDS byte address of the block's key table (hash_dev.hpp, GKEY_BYTES)
    const uint8_t* buf_end = nullptr;  // (the group key loads 16 bytes at a time: where the shard's memory ends)
    uint64_t base = 0, limit = 0;
    uint32_t err = 0;

    __device__ __forceinline__ void begin_range(uint64_t b, uint64_t lim) {
        base = b;
        limit = lim;
        T.reset();
    }

    template <class CUR>
    __device__ __forceinline__ void tile(const CUR& cur, uint64_t tile_idx, uint64_t rs, uint64_t re, const uint8_t* __restrict__ buf) {
        if (!D.write) return;  // count pass of the exact fallback: no keys
        T.stage(cur, tile_idx);
    }

    // keys of up to 16 sequence lines, one per quad: `so` = offset of the line's first byte relative to the tile,
    // `ln` = its length, vq = this quad has a line, g = the record's slot
    __device__ __forceinline__ void hash_quads(int32_t so, uint32_t ln, bool vq, uint64_t tile_idx,
                                               const uint8_t* __restrict__ buf, uint64_t g) {
        const uint32_t k = threadIdx.x & 3u;
        const bool in_lds = vq && T.holds(tile_idx, so);
        const uint8_t* gp = buf + (int64_t)tile_idx + (int64_t)so;  // the line in global memory
        const uint32_t la = T.addr(so);                             // ... and in LDS
        auto ld64 = [&](uint32_t o) -> uint64_t {
            uint32_t lo, hi;
            if (in_lds) {
                lds_ld64(la + o, lo, hi);
            } else {
                uint64_t v;
                __builtin_memcpy(&v, gp + o, 8);
                lo = (uint32_t)v;
                hi = (uint32_t)(v >> 32);
            }
            if (FOLD) { lo = fold4(lo); hi = fold4(hi); }
            return ((uint64_t)hi << 32) | lo;
        };
        const uint32_t nst = vq ? ln >> 5 : 0u;
        uint64_t v = k == 0 ? P1 + P2 : (k == 1 ? P2 : (k == 2 ? 0ull : 0ull - P1));
        uint64_t b = k2_init(k);
        const uint64_t qk = k2_q(k);
        for (uint32_t st = 0; __ballot(st < nst) != 0ull; ++st) {
            if (st < nst) {
                const uint64_t w = ld64(32u * st + 8u * k);
                v = xround(v, w);
                if (K2) b = k2_step(b, w, qk);
            }
        }
        // merge: h = sum of the rotated accumulators, then the four xmerge steps (their xround halves in parallel)
        uint64_t h;
        {
            const uint64_t r = xround(0, v);
            const uint32_t ra = k == 0 ? 1u : (k == 1 ? 7u : (k == 2 ? 12u : 18u));
            const uint64_t rot = (v << ra) | (v >> (64u - ra));
            h = quad_sum64(rot);
            h = (h ^ quad_bcast64<0>(r)) * P1 + P4;
            h = (h ^ quad_bcast64<1>(r)) * P1 + P4;
            h = (h ^ quad_bcast64<2>(r)) * P1 + P4;
            h = (h ^ quad_bcast64<3>(r)) * P1 + P4;
            if (ln < 32u) h = P5;  // seed 0
        }
        h += (uint64_t)ln;
        // tail: up to three 8-byte words (their xround halves by lanes 0..2), a 4-byte word, up to three bytes
        const uint32_t t0 = nst << 5, rem = ln & 31u, nw = rem >> 3;
        uint64_t tr = 0;
        if (vq && k < nw) {
            const uint64_t w = ld64(t0 + 8u * k);
            tr = xround(0, w);
            if (K2) b = k2_step(b, w, qk);
        }
        const uint64_t tr0 = quad_bcast64<0>(tr), tr1 = quad_bcast64<1>(tr), tr2 = quad_bcast64<2>(tr);
        if (nw > 0u) { h ^= tr0; h = rotl64(h, 27) * P1 + P4; }
        if (nw > 1u) { h ^= tr1; h = rotl64(h, 27) * P1 + P4; }
        if (nw > 2u) { h ^= tr2; h = rotl64(h, 27) * P1 + P4; }
        const uint32_t cnt = rem & 7u, ro = t0 + 8u * nw;
        uint64_t rest = 0;
        if (vq && cnt) {
            if (in_lds) {
                rest = ld64(ro);  // (reads past the line inside the padded buffer; masked below)
            } else {
                for (uint32_t i = 0; i < cnt; ++i) rest |= (uint64_t)gp[ro + i] << (8u * i);
                if (FOLD) rest = ((uint64_t)fold4((uint32_t)(rest >> 32)) << 32) | fold4((uint32_t)rest);
            }
            rest &= (1ull << (8u * cnt)) - 1ull;
        }
        uint64_t tailb = rest;
        if (rem & 4u) {
            h ^= (uint64_t)(uint32_t)rest * P1;
            h = rotl64(h, 23) * P2 + P3;
            tailb = rest >> 32;
        }
        const uint32_t nb = rem & 3u;
        if (nb > 0u) { h ^= (tailb & 0xFFull) * P5; h = rotl64(h, 11) * P1; }
        if (nb > 1u) { h ^= ((tailb >> 8) & 0xFFull) * P5; h = rotl64(h, 11) * P1; }
        if (nb > 2u) { h ^= ((tailb >> 16) & 0xFFull) * P5; h = rotl64(h, 11) * P1; }
        h = xavalanche(h);
        const uint64_t key2 = !K2 ? 0ull : k2_finish(quad_bcast64<0>(b), quad_bcast64<1>(b), quad_bcast64<2>(b), quad_bcast64<3>(b), rest, ln);
        if (vq && k == 0u && g < limit) {
            H.k1[g] = h;
            if (K2) H.k2[g] = key2;
        }
    }

    // the grouping key of up to 16 sequence lines, one per quad (hash_dev.hpp "GROUPING key"): lane k of the quad takes the
    // 16-byte chunks k, k + 4, ... of its line -- one unaligned ds_read_b128 of text, two aligned ones of keys, eight adds and
    // four v_mad_u64_u32 per chunk, nothing carried from chunk to chunk; two DPP adds fold the quad
    __device__ __forceinline__ void group_quads(int32_t so, uint32_t ln, bool vq, uint64_t tile_idx,
                                                const uint8_t* __restrict__ buf, uint64_t g) {
        const uint32_t k = threadIdx.x & 3u;
        const bool in_lds = vq && T.holds(tile_idx, so);
        const uint8_t* gp = buf + (int64_t)tile_idx + (int64_t)so;  // the line in global memory
        const uint32_t la = T.addr(so);                             // ... and in LDS
        const uint32_t nch = vq ? (ln + 15u) >> 4 : 0u;
        uint64_t a1 = 0, a2 = 0;
        for (uint32_t c = k; __ballot(c < nch) != 0ull; c += 4u) {
            if (c < nch) {
                // (named dwords, no indexed array: one dynamic index sends the four of them through scratch memory)
                uint32_t w0, w1, w2, w3;
                const uint32_t rem = ln - 16u * c;  // bytes of the line from this chunk on (>= 1)
                if (in_lds) {
                    const u32x4 v = *(lds_u32x4_any*)(uintptr_t)(la + 16u * c);  // (reads past the line inside the padded buffer; masked below)
                    w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
                } else {  // a line longer than the carry, or cut by the tile before: from global memory, never past the shard
                    uint64_t lo = 0, hi = 0;
                    const uint8_t* q = gp + 16u * c;
                    if (q + 16 <= buf_end) {
                        __builtin_memcpy(&lo, q, 8);
                        __builtin_memcpy(&hi, q + 8, 8);
                    } else {
                        for (uint32_t i = 0; i < 8u && i < rem; ++i) lo |= (uint64_t)q[i] << (8u * i);
                        for (uint32_t i = 8; i < 16u && i < rem; ++i) hi |= (uint64_t)q[i] << (8u * (i - 8u));
                    }
                    w0 = (uint32_t)lo; w1 = (uint32_t)(lo >> 32); w2 = (uint32_t)hi; w3 = (uint32_t)(hi >> 32);
                }
                if (rem < 16u) {  // zero padding of the last chunk: byte j of the chunk stays iff j < rem
                    const uint32_t sh = (rem & 3u) * 8u, part = (1u << sh) - 1u;  // (rem & 3 == 0: part = 0)
                    const uint32_t q = rem >> 2;                                     // whole dwords kept
                    w0 &= q > 0u ? 0xFFFFFFFFu : part;
                    w1 &= q > 1u ? 0xFFFFFFFFu : (q == 1u ? part : 0u);
                    w2 &= q > 2u ? 0xFFFFFFFFu : (q == 2u ? part : 0u);
                    w3 &= q == 3u ? part : 0u;
                }
                if (FOLD) { w0 = fold4(w0); w1 = fold4(w1); w2 = fold4(w2); w3 = fold4(w3); }
                const uint32_t ka = gk + (c & (GKEY_POS - 1u)) * 32u;
                const uint4 K = lds_r128(ka), Q = lds_r128(ka + 16u);
                a1 += (uint64_t)(w0 + K.x) * (uint64_t)(w1 + K.y);
                a1 += (uint64_t)(w2 + K.z) * (uint64_t)(w3 + K.w);
                a2 += (uint64_t)(w0 + Q.x) * (uint64_t)(w1 + Q.y);
                a2 += (uint64_t)(w2 + Q.z) * (uint64_t)(w3 + Q.w);
                if ((c & (GKEY_POS - 1u)) >= GKEY_POS - 4u) { a1 = gkey_stir1(a1); a2 = gkey_stir2(a2); }
            }
        }
        a1 = quad_sum64(a1);
        a2 = quad_sum64(a2);
        const uint64_t h = gkey_finish(a1, a2, (uint64_t)ln);
        if (vq && k == 0u && g < limit) H.k1[g] = h;
    }

    template <bool FASTQ, bool ALL>
    __device__ __forceinline__ void batch(Lds<FASTQ, ALL>& L, uint32_t E, uint32_t wb, uint64_t tile_idx,
                                          uint32_t tile_rel, uint64_t re, const uint8_t* __restrict__ buf) {
        static_assert(FASTQ && !ALL, "the rmdup sink runs on the sparse FASTQ path");
        const int lane = threadIdx.x & 63;
        for (uint32_t e0 = 0; e0 < E; e0 += WAVE) {
            const uint32_t e = e0 + lane;
            const bool on = e < E;
            const uint32_t s = HISTORY + (on ? e : 0);
            const uint32_t rank = wb + e;
            const uint32_t p = L.pos[s];
            const uint64_t abs_next = tile_idx + (uint64_t)(uint32_t)(p - tile_rel) + 1;  // byte after the newline
            const uint32_t role = rank & 3u;
            int32_t line_off = 0;
            uint32_t line_len = 0;
            bool is_seq = false;
            if (on && D.write) {
                // the structural validation and the table row of k_index (strict 4-line FASTQ)
                if (role == 1u) {
                    if (next_char(L, s, abs_next, re, buf) != '+') err |= ERR_BAD_PLUS;
                    const uint32_t prev = L.pos[s - 1];
                    line_len = p - prev - 1u;
                    line_off = (int32_t)(prev - tile_rel) + 1;
                    is_seq = true;
                } else if (role == 0u) {
                    if (next_char(L, s, abs_next, re, buf) == '+') err |= ERR_BAD_PLUS;
                } else if (role == 3u) {
                    const uint32_t p1 = L.pos[s - 1], p2 = L.pos[s - 2], p3 = L.pos[s - 3], p4 = L.pos[s - 4];
                    const uint32_t lq = p - p1 - 1u, lp = p1 - p2 - 1u, ls = p2 - p3 - 1u, lh = p3 - p4 - 1u;
                    if (lq != ls) err |= ERR_LEN_MISMATCH;
                    if (abs_next < re && next_char(L, s, abs_next, re, buf) != '@') err |= ERR_BAD_HEADER;
                    const uint64_t g = base + (rank >> 2);
                    if (g < limit) {
                        D.t.start[g] = abs_of(p4, tile_idx, tile_rel) + 1;
                        D.t.l_head[g] = lh;
                        D.t.l_seq[g] = ls;
                        D.t.aux[g] = lp;
                    } else {
                        err |= ERR_CAPACITY;
                    }
                }
            }
            if (D.write) {
                // the sequence-line events of these 64 are lanes first, first + 4, ...: quad q takes the q-th of them
                const uint32_t first = (1u - (wb + e0)) & 3u;
                const int src = (int)(first + 4u * ((uint32_t)lane >> 2));
                const int32_t so = __shfl(line_off, src, 64);
                const uint32_t ln = (uint32_t)__shfl((int)line_len, src, 64);
                const bool vq = __shfl((int)is_seq, src, 64) != 0;
                if (__ballot(vq) != 0ull) {
                    const uint64_t g = base + ((wb + e0 + (uint32_t)src) >> 2);
                    if constexpr (MODE == MODE_GROUP) group_quads(so, ln, vq, tile_idx, buf, g);
                    else hash_quads(so, ln, vq, tile_idx, buf, g);
                }
            }
        }
    }
};

#ifndef BSK_RMSTREAM_WAVES
#define BSK_RMSTREAM_WAVES 4  // both keys: 96 -> 128 VGPRs, no spills: 9.0 ms per 25 GB against 9.7 at 5 (scripts/r03_var.sh); the pass is bound by its instructions
#endif
#ifndef BSK_RMSTREAM_WAVES_K1
#define BSK_RMSTREAM_WAVES_K1 5  // k1 alone: 98 VGPRs wanted, 5 waves per SIMD fit without spills: 7.5 ms against 8.0 at 4
#endif

#ifndef BSK_RMSTREAM_WAVES_G
#define BSK_RMSTREAM_WAVES_G 5
#endif

template <bool DPP, bool FOLD, int MODE>
__device__ __forceinline__ void rmdup_stream_body(const uint8_t* __restrict__ buf, uint64_t n, const uint64_t* __restrict__ anchors,
                                                  uint32_t nranges, uint32_t* __restrict__ queue, const IndexDev& D, const HashDev& H) {
    __shared__ Lds<true, false> s_l[WAVES_PER_BLOCK];
    __shared__ __attribute__((aligned(16))) uint8_t s_tb[WAVES_PER_BLOCK][TBUF];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Lds<true, false>& L = s_l[wave];
    RmdupSink<DPP, FOLD, MODE> sink;
    sink.D = D;
    sink.H = H;
    sink.T.tb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s_tb[wave];
    sink.buf_end = buf + n;
    if constexpr (MODE == MODE_GROUP) {
        __shared__ __attribute__((aligned(16))) uint32_t s_gk[GKEY_BYTES / 4];
        for (uint32_t i = threadIdx.x; i < GKEY_BYTES / 4u; i += blockDim.x) s_gk[i] = gkey_word(i >> 3, i & 7u);
        __syncthreads();
        sink.gk = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)s_gk;
    }
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
        const uint32_t lines = stream_range<true, false, DPP>(L, buf, n, rs, re, re == n_eff, P, sink);
        if (D.write != 1 && lane == 0) D.range_count[r] = (uint64_t)(lines >> 2);
    }
    const uint32_t err = wave_or_u32(sink.err);
    if (lane == 0 && err) atomicOr((unsigned long long*)&D.status[0], (unsigned long long)err);
}

template <bool DPP, bool FOLD>
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) __attribute__((amdgpu_waves_per_eu(BSK_RMSTREAM_WAVES, 8))) void k_rmdup_stream(
    const uint8_t* __restrict__ buf, uint64_t n, const uint64_t* __restrict__ anchors, uint32_t nranges, uint32_t* __restrict__ queue,
    IndexDev D, HashDev H) {
    rmdup_stream_body<DPP, FOLD, MODE_K1K2>(buf, n, anchors, nranges, queue, D, H);
}
template <bool DPP, bool FOLD>
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) __attribute__((amdgpu_waves_per_eu(BSK_RMSTREAM_WAVES_K1, 8))) void k_rmdup_stream_k1(
    const uint8_t* __restrict__ buf, uint64_t n, const uint64_t* __restrict__ anchors, uint32_t nranges, uint32_t* __restrict__ queue,
    IndexDev D, HashDev H) {
    rmdup_stream_body<DPP, FOLD, MODE_K1>(buf, n, anchors, nranges, queue, D, H);
}
template <bool DPP, bool FOLD>
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) __attribute__((amdgpu_waves_per_eu(BSK_RMSTREAM_WAVES_G, 8))) void k_rmdup_stream_g(
    const uint8_t* __restrict__ buf, uint64_t n, const uint64_t* __restrict__ anchors, uint32_t nranges, uint32_t* __restrict__ queue,
    IndexDev D, HashDev H) {
    rmdup_stream_body<DPP, FOLD, MODE_GROUP>(buf, n, anchors, nranges, queue, D, H);
}

// one block per range: its slice of the sparse table and of the sparse keys to their dense positions
__global__ __launch_bounds__(256) void k_rmdup_compact(RecordTable sp, uint64_t sparse_cap, const uint64_t* __restrict__ range_count,
                                                       const uint64_t* __restrict__ range_base, RecordTable dn, HashDev hs, HashDev hd) {
    const uint32_t r = blockIdx.x;
    const uint64_t cnt = range_count[r], src = (uint64_t)r * sparse_cap, dst = range_base[r];
    for (uint64_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        dn.start[dst + i] = sp.start[src + i];
        dn.l_head[dst + i] = sp.l_head[src + i];
        dn.l_seq[dst + i] = sp.l_seq[src + i];
        dn.aux[dst + i] = sp.aux[src + i];
        hd.k1[dst + i] = hs.k1[src + i];
        if (hs.k2) hd.k2[dst + i] = hs.k2[src + i];
    }
}

template <bool DPP, bool FOLD>
const void* kernel_ptr(int mode) {
    return mode == MODE_K1K2 ? (const void*)k_rmdup_stream<DPP, FOLD>
                             : (mode == MODE_GROUP ? (const void*)k_rmdup_stream_g<DPP, FOLD> : (const void*)k_rmdup_stream_k1<DPP, FOLD>);
}
const void* kernel_of(bool dpp, bool fold, int mode) {
    return dpp ? (fold ? kernel_ptr<true, true>(mode) : kernel_ptr<true, false>(mode))
               : (fold ? kernel_ptr<false, true>(mode) : kernel_ptr<false, false>(mode));
}

}  // namespace

hipError_t launch_rmdup_stream(bool dpp, bool fold, int mode, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors,
                               uint32_t nranges, uint32_t* queue, const IndexDev& D, const HashDev& H, hipStream_t st) {
    IndexDev d = D;
    HashDev h = H;
    void* args[] = {(void*)&buf, (void*)&n, (void*)&anchors, (void*)&nranges, (void*)&queue, (void*)&d, (void*)&h};
    return hipLaunchKernel(kernel_of(dpp, fold, mode), dim3(blocks), dim3(WAVES_PER_BLOCK * WAVE), args, 0, st);
}

int rmdup_stream_max_blocks_per_cu(bool dpp, bool fold, int mode) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel_of(dpp, fold, mode), WAVES_PER_BLOCK * WAVE, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

hipError_t launch_rmdup_compact(const RecordTable& sparse, uint64_t sparse_cap, const uint64_t* range_count,
                                const uint64_t* range_base, uint32_t nranges, const RecordTable& dense, const HashDev& hs,
                                const HashDev& hd, hipStream_t st) {
    hipLaunchKernelGGL(k_rmdup_compact, dim3(nranges), dim3(256), 0, st, sparse, sparse_cap, range_count, range_base, dense, hs, hd);
    return hipGetLastError();
}

}  // namespace bsk
