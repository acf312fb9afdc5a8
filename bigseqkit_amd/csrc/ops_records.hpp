// Whole-record operators: `range` / `head` (RangePrepare + RangeFilter, bigseqkit-lib/range.go:20-43) and `duplicate`
// (bigseqkit-lib/duplicate.go:13-30).  Neither parses a record: the element is the record text as PlainFile + ReadFixer
// hand it over (bigseqkit-lib/helper.go:41-66 -- one trailing '\n' stripped), written back followed by '\n'.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {

struct RecordsParams {
    int fastq;
    int64_t first_record;  // index of record 0 of this shard in the whole input (MapWithIndex)
    int64_t lo, hi;        // records with lo <= index < hi are kept (range.go:33-37)
    uint32_t times;        // copies of every kept record (duplicate.go:24-28)
};

constexpr uint32_t ERR_RECORD_TOO_LARGE = 2048u;

// out_len[i] = bytes record i contributes: times x (text + '\n'), or 0
hipError_t launch_records_size(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const RecordsParams& P,
                               uint32_t* out_len, uint64_t* status, hipStream_t st);
// out[out_off[i] ...] = the copies of record i; one block per 16 KiB of OUTPUT (any mix of record sizes balances);
// tile_first: scratch, records_copy_tiles(total) entries
uint64_t records_copy_tiles(uint64_t total);
hipError_t launch_records_copy(const uint8_t* buf, const RecordTable& t, const RecordsParams& P, const uint64_t* out_off,
                               uint32_t* tile_first, uint8_t* out, uint64_t total, hipStream_t st);

}  // namespace bsk
