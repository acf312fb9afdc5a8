// Host-visible interface of stream_names.hip: `seq -n` / `seq -n -i` on FASTQ written by the streaming pass itself.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace bsk {

struct NamesDev {
    uint8_t* slices;        // [nranges][slice_cap]: range r writes its names at r * slice_cap
    uint64_t slice_cap;     // bytes per slice, a multiple of 16, < 2^32
    uint64_t* range_bytes;  // [nranges] bytes range r produced
    uint64_t* range_count;  // [nranges] records of range r
    uint64_t* status;       // [0] error flags (stream_stats.hpp)
    int only_id;            // print the ID (parseHeadIDAndDesc, helper.go:329-369) instead of the whole head
    int id_mode;            // 0 default regexp, 1 --id-ncbi
};

hipError_t launch_names(bool dpp, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors, uint32_t nranges,
                        uint32_t* queue, const NamesDev& D, hipStream_t st);
int names_max_blocks_per_cu(bool dpp);
// out[range_base[r] .. + range_bytes[r]) = slice r
hipError_t launch_names_compact(const NamesDev& D, const uint64_t* range_base, uint32_t nranges, uint8_t* out, hipStream_t st);

}  // namespace bsk
