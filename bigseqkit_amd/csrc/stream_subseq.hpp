// Host-visible interface of stream_subseq.hip: `subseq -r a:b` on FASTQ written by the streaming pass itself.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace bsk {

struct SubseqDev {
    uint8_t* slices;        // [nranges][slice_cap]: range r writes its records at r * slice_cap
    uint64_t slice_cap;     // bytes per slice, a multiple of 16, < 2^32
    uint64_t* range_bytes;  // [nranges] bytes range r produced
    uint64_t* range_count;  // [nranges] records of range r
    uint64_t* status;       // [0] error flags (stream_stats.hpp)
    int region_start, region_end;  // Seq.SubSeq(start, end): 1-based, inclusive, negative = from the end
};

hipError_t launch_subseq_stream(bool dpp, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors,
                                uint32_t nranges, uint32_t* queue, const SubseqDev& D, hipStream_t st);
int subseq_stream_max_blocks_per_cu(bool dpp);

}  // namespace bsk
