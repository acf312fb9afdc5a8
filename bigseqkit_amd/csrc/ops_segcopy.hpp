// Segmented copy: the output text is a sequence of segments, segment k = `seg_off[k + 1] - seg_off[k]` bytes copied
// verbatim from the absolute device address seg_src[k] (0: someone else writes these bytes).  Driven by the OUTPUT -- one
// wave per 4 KiB of it, aligned 16-byte stores, unaligned 16-byte loads -- so the stores are full lines whatever the
// record size (ops_segcopy.hip).  Used for the operators whose output records are byte-for-byte copies of input records
// (FASTQ records that Format() reproduces exactly: seq, grep, rmdup, pair, common ...; bigseqkit-lib/seq.go:176-269).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {

constexpr uint32_t SEG_TILE = 4096;  // output bytes per wave

inline uint64_t seg_tiles(uint64_t total) { return (total + SEG_TILE - 1) / SEG_TILE; }

// seg_src of the records of a FASTQ table whose output (out_len[i] != 0 bytes at out_off[i]) is the record text itself:
// bare '+' line, and the text incl. its final newline lies inside the shard.  Other records with output get 0 and are
// counted in *n_other (they stay with the record-wise emit kernel).
hipError_t launch_seg_build_fastq(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const uint32_t* out_len,
                                  uint64_t* seg_src, uint64_t* n_other, hipStream_t st,
                                  const uint32_t* ren_ord = nullptr /* rename: records with ren_ord[i] != 0 get a new head */);
// sort: the same for the records in the order perm[0], perm[1], ... (seg_sorted[k] belongs to record perm[k]; seg_rec[i] to
// record i, for the record-wise emit of what is left)
hipError_t launch_seg_build_fastq_perm(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const uint32_t* out_len,
                                       const uint32_t* perm, uint64_t* seg_sorted, uint64_t* seg_rec, uint64_t* n_other, hipStream_t st);
// range / head: out_len[i] = text + 1; verbatim when the byte after the text is the '\n' (else counted in *n_other and
// written by launch_seg_fix_text)
hipError_t launch_seg_build_text(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const uint32_t* out_len,
                                 uint64_t* seg_src, uint64_t* n_other, hipStream_t st);
hipError_t launch_seg_fix_text(const uint8_t* buf, const RecordTable& t, const uint32_t* out_len, const uint64_t* out_off,
                               const uint64_t* seg_src, uint8_t* out, hipStream_t st);
// duplicate: `times` segments per record (seg arrays of t.n * times (+ 1) entries)
hipError_t launch_seg_build_text_times(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const uint32_t* out_len,
                                       const uint64_t* out_off, uint32_t times, uint64_t* seg_src, uint64_t* seg_off2, uint64_t* n_other,
                                       hipStream_t st);
hipError_t launch_seg_fix_text_times(const uint8_t* buf, const RecordTable& t, const uint32_t* out_len, const uint64_t* out_off,
                                     uint32_t times, const uint64_t* seg_src, uint8_t* out, hipStream_t st);
// first4k[T] = the segment that holds output byte T * SEG_TILE (T < seg_tiles(total))
hipError_t launch_seg_first(const uint64_t* seg_off, uint64_t nseg, uint32_t* first4k, hipStream_t st);
hipError_t launch_seg_copy(const uint64_t* seg_src, const uint64_t* seg_off, uint64_t nseg, const uint32_t* first4k,
                           uint8_t* out, uint64_t total, const uint8_t* lo, const uint8_t* hi, hipStream_t st);

// the output bytes [from, to) of the text of `whole` bytes into dst[0, to - from) (from: a multiple of SEG_TILE, dst 16-byte
// aligned): how a consumer gathers a result that is still a list of slices piece by piece (store.cpp)
hipError_t launch_seg_copy_range(const uint64_t* seg_src, const uint64_t* seg_off, uint64_t nseg, const uint32_t* first4k, uint8_t* dst,
                                 uint64_t from, uint64_t to, uint64_t whole, const uint8_t* lo, const uint8_t* hi, hipStream_t st);
// src[r] = address of slice r of a per-range slice buffer (the segment form of the streaming passes' slices)
hipError_t launch_slice_srcs(const uint8_t* slices, uint64_t slice_cap, uint32_t nranges, uint64_t* src, hipStream_t st);

}  // namespace bsk
