// Class patterns: every pattern position accepts a SET of bytes (256-bit mask).  One matcher serves
//   -d / --degenerate   Degenerate2Regexp classes, e.g. R -> [AG]   (bigseqkit-lib/grep.go:141-147, locate.go:141-146)
//   -m / --max-mismatch Hamming distance <= k in place of the reference's per-record FM-index
//                       (grep.go:297-339, locate.go:236-254); a position accepts exactly its letter
//   -i                  both cases of a letter are in the set
// The '-' strand uses the reverse-complemented class pattern on the forward text, like the exact path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "text_dev.hpp"

namespace bsk {

// does the class pattern (m positions, 8 dwords each) match at forward position f with <= max_mm misses?
// l = sequence length; positions >= l wrap (second copy of a --circular text)
__device__ __forceinline__ bool class_match_at(const Text& T, uint32_t l, const uint32_t* __restrict__ cls, uint32_t m,
                                               uint64_t f, int max_mm) {
    int mm = 0;
    for (uint32_t q = 0; q < m; ++q) {
        uint64_t j = f + q;
        if (j >= l) j -= l;
        const uint8_t c = T.at((uint32_t)j);
        if (!((cls[q * 8u + (c >> 5)] >> (c & 31u)) & 1u))
            if (++mm > max_mm) return false;
    }
    return true;
}

// ASCII lower-case of four bytes at once
__device__ __forceinline__ uint32_t fold_dword4(uint32_t x) {
    const uint32_t ge_a = (x & 0x7F7F7F7Fu) + 0x3F3F3F3Fu;   // bit 7 set where byte >= 'A'
    const uint32_t ge_z1 = (x & 0x7F7F7F7Fu) + 0x25252525u;  // bit 7 set where byte >= '['
    const uint32_t up = ge_a & ~ge_z1 & ~x & 0x80808080u;      // 'A'..'Z' (and byte < 0x80)
    return x | (up >> 2);
}

// pattern bytes [4, m) against the text at `t` (a candidate whose first four bytes already matched): ONE 16-byte load of
// text and of pattern per 16 bytes instead of a byte loop whose every load waits for the previous compare.
// The pattern buffer is padded by 16 bytes; text bytes at or beyond buf_end read as 0.
__device__ __forceinline__ bool verify_from4(const uint8_t* t, const uint8_t* buf_end, bool fold, const uint8_t* pp, uint32_t m) {
    for (uint32_t q = 4; q < m; q += 16) {
        const uint32_t nb = m - q < 16u ? m - q : 16u;
        uint32_t a[4] = {0, 0, 0, 0};
        if (t + q + 16 <= buf_end) {
            uint4 v;
            __builtin_memcpy(&v, t + q, 16);
            a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
        } else {
            for (uint32_t k = 0; k < 16u && t + q + k < buf_end; ++k) a[k >> 2] |= (uint32_t)t[q + k] << (8 * (k & 3));
        }
        uint4 pv;
        __builtin_memcpy(&pv, pp + q, 16);
        const uint32_t b[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int left = (int)nb - 4 * d;
            if (left <= 0) break;
            const uint32_t mask = left >= 4 ? 0xFFFFFFFFu : ((1u << (8 * left)) - 1u);
            const uint32_t x = fold ? fold_dword4(a[d]) : a[d];
            if ((x ^ b[d]) & mask) return false;
        }
    }
    return true;
}

// FNV-1a over bytes (optionally ASCII-lower-cased): key of the ID / name pattern set, same function on the host
__host__ __device__ __forceinline__ uint64_t fnv1a64(const uint8_t* p, uint32_t n, bool fold) {
    uint64_t h = 1469598103934665603ull;
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t c = p[i];
        if (fold && c >= 'A' && c <= 'Z') c += 32;
        h = (h ^ c) * 1099511628211ull;
    }
    return h ? h : 1ull;  // 0 marks an empty slot
}

}  // namespace bsk
