// The subject of a record for `rmdup` (device code): what RmDupPrepare hashes and RmDupCheck compares
// (/root/reference/bigseqkit-lib/rmdup.go:54-84, 193-199) -- the sequence (-s), the whole name (-n) or the ID -- as a
// byte accessor over the shard text.  Shared by ops_rmdup.hip and ops_rmdup_xcheck.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ops_rmdup.hpp"
#include "text_dev.hpp"

namespace bsk {
namespace {

__device__ __forceinline__ uint8_t lower8(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

// the subject of a record as a byte accessor
struct Subject {
    Text T;  // by sequence (W may be > 0)
    const uint8_t* h;
    uint32_t len;
    bool seq, fold;
    __device__ __forceinline__ uint8_t at(uint32_t i) const {
        const uint8_t c = seq ? T.at(i) : h[i];
        return fold ? lower8(c) : c;
    }
};

__device__ __forceinline__ Subject subject_of(const uint8_t* buf, const RecordTable& t, const TextTable& tt,
                                              const RmDupParams& P, uint64_t i) {
    Subject s;
    s.fold = P.ignore_case;
    s.seq = P.by_seq;
    if (P.by_seq) {
        s.T = text_of(buf, t, tt, i);
        s.h = nullptr;
        s.len = s.T.L;
    } else {
        const uint32_t lh = t.l_head[i];
        const uint8_t* h = buf + t.start[i] + 1;
        uint32_t hl = lh > 0 ? lh - 1 : 0, off = 0;
        if (!P.by_name) hl = id_span_rec(t, i, h, hl, P.id_mode, &off, P.buf_end);
        s.h = h + off;
        s.len = hl;
        s.T.p = nullptr; s.T.L = 0; s.T.W = 0;
    }
    return s;
}

// bytes of fastx.Record.Format(width) for a record with this name and sequence length
__device__ __forceinline__ uint32_t format_len(uint32_t name_len, uint32_t L, int fastq, int width) {
    uint32_t w = L;
    if (width > 0 && L > 0) w += (L - 1) / (uint32_t)width;
    uint32_t n = 1 + name_len + 1 + w + 1;
    if (fastq) n += 2 + w + 1;
    return n;
}

}  // namespace
}  // namespace bsk
