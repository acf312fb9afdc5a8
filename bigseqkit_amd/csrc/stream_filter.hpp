// Host-visible interface of stream_filter.hip: the record table of the records whose sequence line holds an exact
// pattern (grep -s -p, locate -p on FASTQ), built in the streaming pass itself.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {

constexpr uint32_t ERR_FILTER_OVERFLOW = 1u << 20;  // too many pending hits between two record ends: use the table path
constexpr uint32_t FILTER_MAX_ENTRIES = 32;         // (pattern, strand) x 4 alignments: one bit each in the table masks
constexpr uint32_t FILTER_MIN_LEN = 11;             // shorter patterns need not contain two whole dwords of the tile grid
constexpr uint32_t FILTER_MAX_LEN = 64;
constexpr uint32_t FILTER_MAX_PATTERNS = FILTER_MAX_ENTRIES / 4;  // pattern strings (forward and reverse-complemented count separately)

struct FilterDev {
    const uint32_t* t1;       // [512]: T1[256] (entries by the code of their first dword) ++ T2[256] (second dword)
    const uint16_t* ent;      // [FILTER_MAX_ENTRIES] entry -> pattern index | alignment << 5 | pattern length << 8
    const uint32_t* pat_padded;  // [FILTER_MAX_PATTERNS][FILTER_MAX_LEN / 4]: the pattern strings, zero-padded (lower-cased with -i)
    int ignore_case;
    int invert;               // select the records WITHOUT a hit (grep -v)
};

// code of a dword as the kernel computes it (bits 1..2 of every byte): host side of the tables
inline uint32_t filter_code(uint32_t w) {
    return ((w >> 1) & 3u) | (((w >> 9) & 3u) << 2) | (((w >> 17) & 3u) << 4) | (((w >> 25) & 3u) << 6);
}

hipError_t launch_filter(bool dpp, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors, uint32_t nranges,
                         uint32_t* queue, const IndexDev& D, const FilterDev& F, hipStream_t st);
int filter_max_blocks_per_cu(bool dpp);

}  // namespace bsk
