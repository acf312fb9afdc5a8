// Class patterns: every pattern position accepts a SET of bytes (256-bit mask).  One matcher serves
//   -d / --degenerate   Degenerate2Regexp classes, e.g. R -> [AG]   (bigseqkit-lib/grep.go:141-147, locate.go:141-146)
//   -m / --max-mismatch Hamming distance <= k in place of the reference's per-record FM-index
//                       (grep.go:297-339, locate.go:236-254); a position accepts exactly its letter
//   -i                  both cases of a letter are in the set
// The '-' strand uses the reverse-complemented class pattern on the forward text, like the exact path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "text.cuh"

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
