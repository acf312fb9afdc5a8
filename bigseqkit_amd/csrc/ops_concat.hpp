// `concat` (ConcatPrepare + GroupByKey + ConcatJoin, bigseqkit-lib/concat.go:39-165): for every ID
// that occurs in both files, every record of file 1 joined with every record of file 2 of that ID -- header = the ID,
// sequence (and quality) = A followed by B, wrapped at LineWidth; with Full the records of IDs that occur in one file
// only are kept as they are.  The shard is file 1 followed by file 2; records [0, first2) belong to file 1.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"
#include "ops_translate.hpp"  // TextTableH

namespace bsk {

struct ConcatParams {
    int fastq;
    int full;
    int id_mode;
    int line_width;       // 0 for FASTQ
    uint32_t first2;      // first record of file 2
    const uint8_t* buf_end;
};

// per sorted entry (group << 32 | index): seg[i] = {segment start, members of file 1, members of file 2} of record i
hipError_t launch_concat_segments(const uint64_t* sorted, uint64_t n, uint32_t first2, uint32_t* seg /* [3 n] */, hipStream_t st);
// out_len[i] = bytes of all elements record i produces, count[i] = how many elements
hipError_t launch_concat_size(const uint8_t* buf, const RecordTable& t, const ConcatParams& P, const uint64_t* sorted,
                              const uint32_t* seg, uint32_t* out_len, uint32_t* count, uint64_t* status, hipStream_t st);
hipError_t launch_concat_emit(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const ConcatParams& P,
                              const uint64_t* sorted, const uint32_t* seg, const uint32_t* out_len, const uint64_t* out_off,
                              uint8_t* out, uint64_t avg_bytes, hipStream_t st, const uint8_t* seg_done = nullptr);
// FASTQ: five segments per element for the segmented copy (seg_src / seg_off: 5 * elements entries; the caller sets
// seg_off[5 * elements] = total), seg_done[i] = 1 for the records it covers, *n_other counts the others
hipError_t launch_concat_segs(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const ConcatParams& P, const uint64_t* sorted,
                              const uint32_t* seg, const uint32_t* out_len, const uint64_t* out_off, const uint64_t* cnt_off,
                              uint64_t* seg_src, uint64_t* seg_off, uint8_t* seg_done, uint64_t* n_other, hipStream_t st);

}  // namespace bsk
