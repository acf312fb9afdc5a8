// Device side of anchor.hpp with a whole WAVE per search: the same rules (find_line_start, find_fasta_start,
// find_fastq_start -- ReadFixer's job, /root/reference/bigseqkit-lib/helper.go:41-66), the newline searches 1 KiB per step
// with all loads of a step in flight.  One thread per range boundary reading 8 bytes per dependent step is fine for
// 150-base reads (a boundary is a few lines away from a record start) and hopeless for long lines: 2 GB of nanopore-sized
// reads (lines of 2-30 kb) spent 9.3 ms in k_prep and 1.1 ms in everything else; a chromosome on one line would take
// seconds.  Every argument and every result is wave-uniform; all 64 lanes must call together.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "anchor.hpp"

namespace bsk {
namespace wave_anchor {

// position (0..15) of the first byte equal to c among the 16 bytes of v, 16 if none
__device__ __forceinline__ uint32_t first_eq16(const uint4& v, uint32_t rep) {
    const uint32_t w[4] = {v.x ^ rep, v.y ^ rep, v.z ^ rep, v.w ^ rep};
    uint32_t pos = 16u;
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        const uint32_t x = w[d];
        const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 where the byte is zero (exact)
        if (z) pos = 4u * (uint32_t)d + (((uint32_t)__ffs((int)z) - 1u) >> 3);
    }
    return pos;
}

// first index >= from with buf[idx] == c, else n
__device__ __forceinline__ uint64_t find_byte(const uint8_t* __restrict__ buf, uint64_t n, uint64_t from, uint8_t c) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rep = 0x01010101u * c;
    // the first KiB alone: the lines of short reads end inside it, and a second KiB loaded for nothing showed in the step
    // time of the 100 GB run (32 768 boundaries, six searches each)
    {
        const uint64_t i = from + lane * 16u;
        uint32_t p0 = 16u;
        if (i + 16u <= n) {
            uint4 v;
            __builtin_memcpy(&v, buf + i, 16);
            p0 = first_eq16(v, rep);
        } else {
            for (uint32_t b = 0; b < 16u && i + b < n; ++b)
                if (buf[i + b] == c) { p0 = b; break; }
        }
        const uint64_t bal = __ballot(p0 < 16u);
        if (bal) {
            const int l = __ffsll((long long)bal) - 1;
            return from + (uint64_t)l * 16u + (uint32_t)__builtin_amdgcn_readlane((int)p0, l);
        }
    }
    for (uint64_t base = from + 1024u; base < n; base += 2048u) {
        uint32_t pos[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint64_t i = base + (uint64_t)u * 1024u + lane * 16u;
            pos[u] = 16u;
            if (i + 16u <= n) {
                uint4 v;
                __builtin_memcpy(&v, buf + i, 16);
                pos[u] = first_eq16(v, rep);
            } else {
                for (uint32_t b = 0; b < 16u && i + b < n; ++b)
                    if (buf[i + b] == c) { pos[u] = b; break; }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint64_t bal = __ballot(pos[u] < 16u);
            if (bal) {
                const int l = __ffsll((long long)bal) - 1;
                const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)pos[u], l);
                return base + (uint64_t)u * 1024u + (uint64_t)l * 16u + p;
            }
        }
    }
    return n;
}

// first line start in [from, limit), else ANCHOR_NONE: the search of ONE nominal chunk (k_prep, FASTA) -- a chromosome on
// one line spans hundreds of chunks, and every boundary inside it searching to the line's end made k_prep the slowest
// kernel of every FASTA command (390 ms per 250 MB line: one wave walks the whole line)
__device__ __forceinline__ uint64_t find_line_start_within(const uint8_t* buf, uint64_t n, uint64_t from, uint64_t limit) {
    if (from == 0) return 0;
    if (from >= n) return n;
    const uint64_t stop = limit < n ? limit : n;   // a line start below `stop` follows a newline below stop - 1
    if (stop <= from) return ANCHOR_NONE;
    const uint64_t j = find_byte(buf, stop - 1, from - 1, '\n');
    return j < stop - 1 ? j + 1 : ANCHOR_NONE;
}

__device__ __forceinline__ uint64_t find_line_start(const uint8_t* buf, uint64_t n, uint64_t from) {
    if (from == 0) return 0;
    if (from >= n) return n;
    const uint64_t j = find_byte(buf, n, from - 1, '\n');
    return j < n ? j + 1 : n;
}

__device__ __forceinline__ uint64_t find_fasta_start(const uint8_t* buf, uint64_t n, uint64_t from) {
    if (from >= n) return n;
    if (from == 0 && buf[0] == '>') return 0;
    uint64_t j = from == 0 ? find_byte(buf, n, 0, '\n') : find_byte(buf, n, from - 1, '\n');
    while (j < n) {
        if (j + 1 < n && buf[j + 1] == '>') return j + 1;
        j = find_byte(buf, n, j + 1, '\n');
    }
    return n;
}

// the rule of bsk::fastq_record_at (anchor.hpp), line ends found by the wave
__device__ __forceinline__ bool fastq_record_at(const uint8_t* buf, uint64_t n, uint64_t p) {
    if (p >= n || buf[p] != '@') return false;
    const uint64_t e0 = find_byte(buf, n, p, '\n');
    if (e0 >= n) return false;
    const uint64_t s0 = e0 + 1;
    const uint64_t e1 = find_byte(buf, n, s0, '\n');
    if (e1 >= n) return false;
    if (e1 > s0 && buf[s0] == '+') return false;
    const uint64_t p0 = e1 + 1;
    if (p0 >= n || buf[p0] != '+') return false;
    const uint64_t e2 = find_byte(buf, n, p0, '\n');
    if (e2 >= n) return false;
    const uint64_t q0 = e2 + 1;
    const uint64_t e3 = find_byte(buf, n, q0, '\n');  // may be n: last line without '\n'
    if (e3 - q0 != e1 - s0) return false;
    if (e3 + 1 < n) {
        const uint8_t c = buf[e3 + 1];
        if (c != '@' && c != '\n') return false;
    }
    return true;
}

__device__ __forceinline__ uint64_t find_fastq_start(const uint8_t* buf, uint64_t n, uint64_t from, uint64_t limit) {
    if (from >= n) return n;
    if (from == 0 && fastq_record_at(buf, n, 0)) return 0;
    uint64_t j = from == 0 ? find_byte(buf, n, 0, '\n') : find_byte(buf, n, from - 1, '\n');
    while (j < n) {
        if (fastq_record_at(buf, n, j + 1)) return j + 1;
        if (j >= limit) return ANCHOR_NONE;
        j = find_byte(buf, n, j + 1, '\n');
    }
    return n;
}

}  // namespace wave_anchor
}  // namespace bsk
