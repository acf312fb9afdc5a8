// `faidx` index rows (Faidx.Call, bigseqkit-lib/faidx.go:91-229): one row per record,
// "<ID>\t<length>\t<offset>\t<linebases>\t<linewidth>[\t<qualoffset>]" -- the samtools .fai columns, which fall out of the
// record table: length = l_seq, offset = first byte of the sequence, linebases = text_w.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {

struct FaidxParams {
    int fastq;
    int full_head;        // -f: the whole header line is the name
    int id_mode;
    uint64_t base_offset; // file offset of the shard (FaidxOffset, faidx.go:38-48)
    const uint8_t* buf_end;
};

constexpr uint32_t ERR_LINE_LENGTHS = 4096u;  // "different line length in sequence" (faidx.go:129-137); status[1] = record

// out_len[i] = bytes of row i incl. '\n'; irregular records are checked line by line (the reference's rule)
hipError_t launch_faidx_size(const uint8_t* buf, const RecordTable& t, const FaidxParams& P, uint32_t* out_len,
                             uint32_t* linebases, uint64_t* status, hipStream_t st);
hipError_t launch_faidx_rows(const uint8_t* buf, const RecordTable& t, const FaidxParams& P, const uint32_t* linebases,
                             const uint64_t* out_off, uint8_t* out, hipStream_t st);

}  // namespace bsk
