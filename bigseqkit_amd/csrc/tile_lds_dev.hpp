// ============================================================================
// tile_lds_dev.hpp -- a wave's current 4 KiB tile kept in LDS (device code only).
//
// Sinks of the streaming skeleton that need the TEXT of a line when its newline event arrives (the sequence line for
// `rmdup -s`, the header line for `seq -n`) used to re-load it from global memory -- a second fetch of lines the L2 has
// already dropped (k_names: 1.23 x the file in FETCH_SIZE).  With Sink::TILE_HOOK the sink sees the tile while it sits in
// registers: stage() copies it into the wave's LDS buffer (four ds_write_b128 per lane) behind a CARRY of the last
// 512 bytes of the tile before it, so a line that ends in this tile and began up to CARRY bytes before it is contiguous
// in LDS.  Unaligned words come from aligned ds_read_b32 and v_alignbyte.
// ============================================================================
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "stream_core_dev.hpp"

namespace bsk {
namespace tilelds {

#ifndef BSK_TILE_CARRY
#define BSK_TILE_CARRY 512  // (a source may choose: LDS per block decides the blocks per CU -- stream_subseq.hip)
#endif
constexpr uint32_t CARRY = BSK_TILE_CARRY;            // bytes of the previous tile kept in front of the current one
static_assert(CARRY % 16 == 0 && CARRY >= 64 && CARRY <= 1024, "the carry is copied by lanes, 16 bytes each");
constexpr uint32_t TBUF = CARRY + stream::TILE + 16;  // + 16: a word that ends on the last tile byte is read as whole dwords

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ uint4 lds_r128(uint32_t a) {
    const u32x4 v = *(lds_u32x4*)(uintptr_t)a;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_w128(uint32_t a, const uint4& v) {
    u32x4 w;
    w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    *(lds_u32x4*)(uintptr_t)a = w;
}
__device__ __forceinline__ uint32_t lds_r32(uint32_t a) { return *(lds_u32*)(uintptr_t)a; }

struct TileLds {
    uint32_t tb = 0;          // LDS byte address of this wave's buffer (CARRY ++ tile ++ pad), 16-byte aligned
    uint64_t staged = ~0ull;  // tile_idx of the tile in LDS (~0: none)
    bool carry_ok = false;    // the CARRY bytes in front of it are the end of the tile before

    __device__ __forceinline__ void reset() { staged = ~0ull; carry_ok = false; }

    template <class CUR>
    __device__ __forceinline__ void stage(const CUR& cur, uint64_t tile_idx) {
        const uint32_t lane = threadIdx.x & 63u;
        const bool cont = staged != ~0ull && tile_idx == staged + stream::TILE;  // (wave-uniform)
        // The carry is READ first and WRITTEN last: a wave's LDS instructions execute in order, so the tile's own writes
        // (the last of which covers the bytes the carry is read from) may be issued behind the read without waiting for
        // its data -- the read's latency passes while they issue, where read -> wait -> write used to stand in front of them.
        uint4 v = make_uint4(0, 0, 0, 0);
        if (cont && lane < CARRY / 16u) v = lds_r128(tb + stream::TILE + lane * 16u);  // the last CARRY bytes of the old tile
        // (the bytes this lane reads are overwritten by OTHER lanes' stores below: per thread the compiler sees no alias and
        // may sink the load under them -- it did.  A compiler-only barrier: the order of issue is what has to hold)
        asm volatile("" ::: "memory");
        carry_ok = cont;
#pragma unroll
        for (int p = 0; p < stream::NPIECE; ++p) lds_w128(tb + CARRY + (uint32_t)p * stream::PIECE_BYTES + lane * 16u, cur[p]);
        if (cont && lane < CARRY / 16u) lds_w128(tb + lane * 16u, v);
        staged = tile_idx;
        stream::wave_lds_fence();
    }
    // a line whose first byte sits `so` bytes from the start of tile `tile_idx` (negative: before it) and that ends
    // inside that tile: is all of it in LDS?
    __device__ __forceinline__ bool holds(uint64_t tile_idx, int32_t so) const {
        return tile_idx == staged && so >= (carry_ok ? -(int32_t)CARRY : 0);
    }
    __device__ __forceinline__ uint32_t addr(int32_t so) const { return tb + CARRY + (uint32_t)so; }
};

// gfx950 takes LDS accesses of any alignment in ONE instruction (measured for ds_write_b128 in
// scripts/experiments/probe_lds_realign.hip; the parity tests of the sinks below hold the reads): BSK_LDS_UNALIGNED=0 keeps
// the round-2 composition from aligned dwords and v_alignbyte (9 instructions per 16 bytes instead of one)
#ifndef BSK_LDS_UNALIGNED
#define BSK_LDS_UNALIGNED 1
#endif
typedef __attribute__((address_space(3))) u32x4 __attribute__((aligned(1))) lds_u32x4_any;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 __attribute__((aligned(1))) lds_u32x2_any;

// 8 / 4 bytes at LDS byte address a (any alignment; reads whole dwords up to a + 11)
__device__ __forceinline__ void lds_ld64(uint32_t a, uint32_t& lo, uint32_t& hi) {
#if BSK_LDS_UNALIGNED
    const u32x2 v = *(lds_u32x2_any*)(uintptr_t)a;
    lo = v.x; hi = v.y;
    return;
#endif
    const uint32_t a4 = a & ~3u;
    const uint32_t d0 = lds_r32(a4), d1 = lds_r32(a4 + 4u), d2 = lds_r32(a4 + 8u);
    lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u);
    hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
}
// 16 bytes at LDS byte address a (any alignment; reads whole dwords up to a + 19)
__device__ __forceinline__ void lds_ld128(uint32_t a, uint32_t (&w)[4]) {
#if BSK_LDS_UNALIGNED
    const u32x4 v = *(lds_u32x4_any*)(uintptr_t)a;
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    return;
#endif
    const uint32_t a4 = a & ~3u, sh = a & 3u;
    const uint32_t d0 = lds_r32(a4), d1 = lds_r32(a4 + 4u), d2 = lds_r32(a4 + 8u), d3 = lds_r32(a4 + 12u), d4 = lds_r32(a4 + 16u);
    w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
    w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
    w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
    w[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
}

}  // namespace tilelds
}  // namespace bsk
