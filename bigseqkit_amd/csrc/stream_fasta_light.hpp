// Host-visible interface of stream_fasta_light.hip: record starts of a FASTA shard without looking at its lines.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {

constexpr uint32_t ERR_LIGHT_UNFIT = 1u << 22;  // a record whose first line is shorter than 16 bases but not its only one

// phase 1: per nominal chunk r (bytes [r * chunk, (r + 1) * chunk)) the positions of the '>' bytes that begin a line, in
// file order, into slice r of `sparse` (sparse_cap entries per chunk; more: ERR_CAPACITY in status[0]); range_count[r]
hipError_t launch_fasta_starts(int blocks, const uint8_t* buf, uint64_t n_eff, uint64_t chunk, uint32_t nranges, uint32_t* queue,
                               uint64_t* sparse, uint64_t sparse_cap, uint64_t* range_count, uint64_t* status, hipStream_t st);
int fasta_starts_max_blocks_per_cu();
// slices -> t.start[0 .. N), t.start[N] = n_eff
hipError_t launch_fasta_starts_compact(const uint64_t* sparse, uint64_t sparse_cap, const uint64_t* range_count,
                                       const uint64_t* range_base, uint32_t nranges, uint64_t n_eff, uint64_t total, RecordTable t,
                                       hipStream_t st);
// phase 2, one thread per record: header length, length of the first sequence line, region; l_seq and text_w AS IF every
// line but the last is as long as the first (the caller must have the text validated: k_translate_wide does)
hipError_t launch_fasta_heads(const uint8_t* buf, uint64_t n_eff, RecordTable t, uint64_t* status, hipStream_t st);

}  // namespace bsk
