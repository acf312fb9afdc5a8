// Record table ("SoA columnarisation"): what SeqParser.Read
// (/root/reference/bigseqkit-lib/helper.go:219-325) yields per record -- head,
// seq, qual -- kept as offsets into the shard text that stays where it is in HBM.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "stream_stats.hpp"

namespace bsk {

// SoA, device memory.  Record i:
//   header line   = [start[i], start[i] + l_head[i])            (marker byte included)
//   FASTQ: seq    = [start + l_head + 1, +l_seq)                 plus line = aux bytes, qual = same length as seq
//          qual   = [start + l_head + 1 + l_seq + 1 + aux + 1, +l_seq)
//   FASTA: sequence region = [start + l_head + 1, +aux) (may contain '\n'), l_seq = bases in it
//   start[n] = end of the last record
struct RecordTable {
    uint64_t* start = nullptr;
    uint32_t* l_head = nullptr;
    uint32_t* l_seq = nullptr;
    uint32_t* aux = nullptr;
    // FASTA only (written by the index pass): 0 = the sequence is one line (or empty), W = every line but the last has
    // W >= 16 bases and the last 1..W (base i sits at i + i / W), 0xFFFFFFFF = irregular wrapping (text_dev.hpp)
    uint32_t* text_w = nullptr;
    // custom --id-regexp only (else null): ID of record i = header bytes [1 + id_off[i], + id_len[i])
    const uint32_t* id_off = nullptr;
    const uint32_t* id_len = nullptr;
    uint64_t n = 0;
    uint64_t cap = 0;
};

// FASTA ranges begin on line starts: what a range knows about the record that is open at its start (head part) and
// about its own last record when that one is still open at the range end (tail part); k_index_stitch joins them.
struct RangePart {
    uint64_t head_bases, head_end_abs;  // bases before the first header; one past the last byte of that record (if closed here)
    uint64_t tail_bases;
    uint32_t head_nlines, head_first, head_last;  // sequence lines of the head part, length of the first / last one
    uint32_t tail_nlines, tail_first, tail_last;
    uint32_t flags;                               // IP_*
    uint32_t pad;
};
constexpr uint32_t IP_VISITED = 1u, IP_HAS_HEADER = 2u, IP_HEAD_CLOSED = 4u, IP_TAIL_OPEN = 8u, IP_HEAD_IRR = 16u, IP_TAIL_IRR = 32u;

struct IndexDev {
    RecordTable t;
    RangePart* parts;            // [nranges] FASTA with line-start ranges, else null
    uint64_t* range_count;       // [nranges] records per range (count pass writes, write pass reads base)
    const uint64_t* range_base;  // [nranges + 1] exclusive scan of range_count (write pass)
    uint64_t* status;            // [0] error flags
    int write;                   // 0: count pass, 1: write pass (exact bases), 2: one-pass sparse write
    uint64_t sparse_cap;         // mode 2: slots reserved per range (range r writes at r * sparse_cap)
};

// skip_chunk (FASTA, anchors from launch_prep with `raw`): the nominal chunk size -- the newline-free middle of a line
// longer than a chunk is not read; 0: every byte
hipError_t launch_index(bool fastq, bool dpp, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors,
                        uint32_t nranges, uint32_t* queue, const IndexDev& D, hipStream_t st, uint64_t skip_chunk = 0);
int index_max_blocks_per_cu(bool fastq, bool dpp);
// exclusive scan of u64 counts (n <= a few 10^4; one block): out[0..n], out[n] = total
hipError_t launch_scan_small(const uint64_t* in, uint64_t* out, uint32_t n, hipStream_t st, uint64_t* total_at = nullptr);
// exclusive scan u32 -> u64 over N items (N up to 2^32): out[0..N], out[N] = total; tmp: u64[(N + 2047) / 2048 + 1]
// two scans of n values each in one launch; total0 / total1 (may be null) also receive the sums
hipError_t launch_scan_small2(const uint64_t* in0, uint64_t* out0, uint64_t* total0, const uint64_t* in1, uint64_t* out1, uint64_t* total1,
                              uint32_t n, hipStream_t st);
hipError_t launch_scan_u32(const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, hipStream_t st);
hipError_t launch_scan_u32_fin(const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, uint32_t thresh, uint32_t* long_list,
                               uint64_t* fin, hipStream_t st);
hipError_t launch_reset_queue(uint32_t* queue, hipStream_t st);
// gather the per-range slices of a sparse table (mode 2) into a dense one
// FASTA, line-start ranges: complete l_seq / aux / text_w of the records that span ranges (dense table, exact bases)
hipError_t launch_index_stitch(const RecordTable& dense, const RangePart* parts, const uint64_t* range_count,
                               const uint64_t* range_base, uint32_t nranges, uint64_t* status, hipStream_t st);
hipError_t launch_index_compact(const RecordTable& sparse, uint64_t sparse_cap, const uint64_t* range_count,
                                const uint64_t* range_base, uint32_t nranges, const RecordTable& dense, hipStream_t st);

struct VmProgram;
// custom --id-regexp: id_off / id_len of every record of the table (ops_idre.hip)
hipError_t launch_id_spans(const uint8_t* buf, const RecordTable& t, const VmProgram* d_prog, uint32_t* id_off, uint32_t* id_len,
                           hipStream_t st);

}  // namespace bsk
