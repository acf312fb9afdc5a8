// Multi-line FASTQ -> strict 4-line FASTQ on the device (ops_mlfq.hip); host side: normalize_multiline_fastq in
// ops_host.cpp.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace bsk {

struct MlfqScratch {     // one entry per line of the text
    uint32_t* next;      // line after the record that would begin here (0xFFFFFFFF: the line does not begin with '@')
    uint32_t* plus;      // its '+' line
    uint32_t* slen;      // its bases (0xFFFFFFFF: not well-formed)
    uint32_t* exitp;     // first line of its orbit beyond its chunk of lines
    uint32_t* is_start;  // 1: a record of the text begins here (zeroed by the caller)
    uint32_t* entry;     // [mlfq_chunks(L)] first line of the orbit inside a chunk (0xFF-filled by the caller)
    uint64_t* status;    // [0]: 1 record without '@', 2 no '+' line or unmatched lengths, 4 record too large (zeroed by the caller)
};

uint64_t mlfq_blocks(uint64_t n);    // 4 KiB blocks of the newline passes
uint32_t mlfq_chunks(uint32_t L);
hipError_t launch_nl_count(const uint8_t* buf, uint64_t n, uint32_t* cnt, hipStream_t st);
// ls[1 + k] = byte after the k-th newline; base = exclusive scan of cnt
hipError_t launch_nl_write(const uint8_t* buf, uint64_t n, const uint64_t* base, uint64_t* ls, hipStream_t st);
hipError_t launch_trailing_blank(const uint64_t* ls, uint32_t L, uint32_t* tb, hipStream_t st);
hipError_t launch_mlfq_resolve(const uint8_t* buf, const uint64_t* ls, uint32_t L, uint32_t tb, const MlfqScratch& S, hipStream_t st);
hipError_t launch_mlfq_list(const uint64_t* ls, uint32_t L, const MlfqScratch& S, const uint64_t* rank, uint32_t* rec_line,
                            uint32_t* out_len, hipStream_t st);
hipError_t launch_mlfq_emit(const uint8_t* buf, const uint64_t* ls, const MlfqScratch& S, const uint32_t* rec_line,
                            const uint64_t* out_off, uint64_t nrec, uint8_t* out, hipStream_t st);

// the records of the text as they stand: start[r] = first byte of record r, start[nrec] = the byte behind the last record
hipError_t launch_mlfq_starts(const uint64_t* ls, const MlfqScratch& S, const uint32_t* rec_line, uint64_t nrec, uint64_t n,
                              uint64_t* start, hipStream_t st);

}  // namespace bsk
