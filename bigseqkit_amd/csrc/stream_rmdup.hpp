// Host-visible interface of stream_rmdup.hip: the record table of a FASTQ shard AND the two 64-bit keys of every record's
// sequence (hash_dev.hpp) from ONE streaming pass -- `rmdup -s` reads the shard once before it groups.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {

struct HashDev {
    uint64_t* k1;  // XXH64(sequence, seed 0), indexed like IndexDev::t (sparse slices in mode 2, dense in mode 1)
    uint64_t* k2;  // the second key
};

// mode: 0 = XXH64 alone, 1 = XXH64 + k2, 2 = the grouping key of hash_dev.hpp (k1 only; the caller compares the bytes)
hipError_t launch_rmdup_stream(bool dpp, bool fold, int mode, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors,
                               uint32_t nranges, uint32_t* queue, const IndexDev& D, const HashDev& H, hipStream_t st);
int rmdup_stream_max_blocks_per_cu(bool dpp, bool fold, int mode);
// gather the per-range slices of the sparse table and of the sparse keys into the dense arrays (k_index_compact + keys)
hipError_t launch_rmdup_compact(const RecordTable& sparse, uint64_t sparse_cap, const uint64_t* range_count,
                                const uint64_t* range_base, uint32_t nranges, const RecordTable& dense, const HashDev& hs,
                                const HashDev& hd, hipStream_t st);

}  // namespace bsk
