// Random-access view of a record's sequence (device code).
// FASTQ sequences are contiguous in the shard.  A multi-line FASTA sequence is used in
// place when all of its lines but the last have one width W (base i lives at i + i / W);
// irregularly wrapped records are copied once into a linear side buffer (ops_text.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {

constexpr uint32_t TEXT_IRREGULAR = 0xFFFFFFFFu;

struct TextTable {            // device arrays; null for FASTQ
    const uint32_t* text_w;   // [n] 0 contiguous, W uniform width, TEXT_IRREGULAR -> linear copy
    const uint64_t* lin_off;  // [n + 1]
    const uint8_t* lin;
    uint64_t lin_n = 0;       // bytes in lin (0: unknown -- kernels then treat linear copies as unbounded-unsafe for wide loads)
};

struct Text {
    const uint8_t* p;
    uint32_t L;
    uint32_t W;
    __device__ __forceinline__ uint8_t at(uint32_t i) const { return W ? p[i + i / W] : p[i]; }
};

__device__ __forceinline__ Text text_of(const uint8_t* buf, const RecordTable& t, const TextTable& tt, uint64_t i) {
    Text T;
    T.p = buf + t.start[i] + t.l_head[i] + 1;
    T.L = t.l_seq[i];
    T.W = 0;
    if (tt.text_w) {
        const uint32_t w = tt.text_w[i];
        if (w == TEXT_IRREGULAR) T.p = tt.lin + tt.lin_off[i];
        else T.W = w;
    }
    return T;
}

// index of the first byte `c` in h[0, n), or n.  With `lim` (one past the last readable byte of the buffer) the search
// takes 16 bytes per load and finds the byte with SWAR instead of one dependent byte load per character.
__device__ inline uint32_t find_byte_in(const uint8_t* h, uint32_t n, uint8_t c, const uint8_t* lim) {
    uint32_t i = 0;
    const uint32_t rep = 0x01010101u * c;
    while (i < n) {
        if (lim && h + i + 16 <= lim) {
            uint4 v;
            __builtin_memcpy(&v, h + i, 16);
            const uint32_t w[4] = {v.x ^ rep, v.y ^ rep, v.z ^ rep, v.w ^ rep};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t z = ~(((w[d] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w[d] | 0x7F7F7F7Fu);  // 0x80 in every zero byte
                if (z) {
                    const uint32_t at = i + 4u * d + (((uint32_t)__ffs((int)z) - 1u) >> 3);
                    return at < n ? at : n;
                }
            }
            i += 16;
        } else {
            if (h[i] == c) return i;
            ++i;
        }
    }
    return n;
}

// ID length inside a header (marker excluded): parseHeadIDAndDesc,
// /root/reference/bigseqkit-lib/helper.go:329-369 (default regexp and --id-ncbi)
__device__ inline uint32_t id_span_of(const uint8_t* h, uint32_t n, int id_mode, uint32_t* id_off, const uint8_t* lim = nullptr) {
    *id_off = 0;
    if (id_mode == 0) {
        const uint32_t s = find_byte_in(h, n, ' ', lim);   // up to the first ' ' (a leading one falls through), else '\t'
        if (s < n && s > 0) return s;
        const uint32_t tb = find_byte_in(h, n, '\t', lim);
        if (tb < n && tb > 0) return tb;
        return n;
    }
    uint32_t a = 0;
    while (a < n && h[a] != '|') ++a;
    while (a < n) {
        uint32_t b = a + 1;
        while (b < n && h[b] != '|') ++b;
        if (b >= n) break;
        if (b > a + 1 && b + 1 < n && h[b + 1] == ' ') { *id_off = a + 1; return b - a - 1; }
        a = b;
    }
    return n;
}

// the same for record `rec` of a table: with a custom --id-regexp the spans were computed once per shard by the
// position-reporting matcher (k_id_spans, ops_idre.hip: FindSubmatch(head)[1], helper.go:362-368) and sit in the table
__device__ inline uint32_t id_span_rec(const RecordTable& t, uint64_t rec, const uint8_t* h, uint32_t n, int id_mode,
                                       uint32_t* id_off, const uint8_t* lim = nullptr) {
    if (t.id_len) { *id_off = t.id_off[rec]; return t.id_len[rec]; }
    return id_span_of(h, n, id_mode, id_off, lim);
}

// description after the ID (default regexp only; helper.go:331-345, incl. its skip-two loop)
__device__ inline uint32_t desc_of(const uint8_t* h, uint32_t n, int id_mode, uint32_t id_len, uint32_t* desc_off) {
    *desc_off = n;
    if (id_mode != 0 || id_len >= n) return 0;
    uint32_t j = id_len + 1;
    for (; j < n; j++) {
        if (h[j] == ' ' || h[j] == '\t') j++;
        else break;
    }
    if (j >= n) return 0;
    *desc_off = j;
    return n - j;
}

}  // namespace bsk
