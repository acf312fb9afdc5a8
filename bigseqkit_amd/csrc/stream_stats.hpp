// Host-visible interface of stream_stats.hip
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace bsk {

constexpr int STATS_HDR = 8;  // == BSK_STATS_HDR in include/bsk.h
constexpr int MAX_GAP_LETTERS = 8;

// error flags raised by the kernels (status[0]); mirrored in stream_core_dev.hpp
constexpr uint32_t ERR_BAD_HEADER = 1u;     // record does not start with '@' / '>'
constexpr uint32_t ERR_BAD_PLUS = 2u;       // FASTQ: third line does not start with '+' (or sequence line does)
constexpr uint32_t ERR_LEN_MISMATCH = 4u;   // FASTQ: len(seq) != len(qual)
constexpr uint32_t ERR_TRUNCATED = 8u;      // FASTQ: shard ends inside a record
constexpr uint32_t ERR_ANCHOR = 16u;        // a range did not end on a record boundary
constexpr uint32_t ERR_LINE_TOO_LONG = 32u; // a line longer than 2^31 bytes
constexpr uint32_t ERR_CAPACITY = 64u;      // an output table was too small

// constants of the byte predicates (only read by the -a kernels)
struct PredConsts {
    uint32_t k20, k30;  // (0x80 - threshold) replicated in the 4 bytes
    uint32_t gap_rep[MAX_GAP_LETTERS];
    int ngap;
    // prefilter of the gap count (`stats -a` on FASTQ, line-role path): (0x80 - (largest gap letter + 1)) replicated --
    // 16 bytes that are all above the largest gap letter hold no gap; 0xFFFFFFFF = no prefilter (a gap letter >= 127)
    uint32_t kgap;
};

struct StatsDev {
    uint64_t* vec;       // stats vector (see include/bsk.h)
    uint64_t* status;    // [0] error flags, [1] number of overflow lengths
    uint64_t* overflow;  // lengths >= hist_cap
    uint64_t overflow_cap;
    uint32_t hist_cap;
    PredConsts pred;
    // FASTA ranges begin on LINE starts, so a record (a chromosome) may span many ranges.  Per range: bases of the part
    // before its first header (r_head), bases after its last header when that record is still open at the range end
    // (r_tail), and flags; k_stats_stitch adds up the records that cross range boundaries.
    uint64_t* r_head;
    uint64_t* r_tail;
    uint32_t* r_flags;
};

constexpr uint32_t RF_VISITED = 1u, RF_HAS_HEADER = 2u, RF_HEAD_CLOSED = 4u, RF_TAIL_OPEN = 8u;
// `-a` and lines longer than a chunk: RF_MID = an empty range whose nominal chunk lies inside a long line; r_head holds the
// gap letters among its bytes.  RF_SKIP_SEQ = the range's last line is a sequence line: the RF_MID ranges behind it are its
constexpr uint32_t RF_MID = 16u, RF_SKIP_SEQ = 32u;

// line_mode (FASTA only): anchors are line starts, not record starts
// raw (FASTA line mode; [nranges + 1] scratch): every boundary searches only its own chunk for a line start and the
// boundaries without one take the next anchor (k_prep_fill) -- a line longer than a chunk (a chromosome on one line) then
// lies in ONE range followed by empty ones, and the streaming kernels skip its newline-free middle (skip_chunk below)
hipError_t launch_prep(bool fastq, const uint8_t* buf, uint64_t n, uint64_t chunk, uint32_t nranges,
                       uint64_t* anchors, uint32_t* queue, hipStream_t st, bool line_mode = false, uint64_t* raw = nullptr);
hipError_t launch_stats_stitch(uint32_t nranges, const StatsDev& D, hipStream_t st);
// more than MAX_GAP_LETTERS gap letters (`stats -a -G ...`; the reference takes any number, bigseqkit-lib/stats.go:36-43,
// 102): the streaming pass counts none and this pass adds, over the record table of the shard, the sequence bytes that are
// in the 256-bit set -- a rare path, one wave per record
struct RecordTable;
hipError_t launch_gap_set_count(const uint8_t* buf, const RecordTable& t, const uint32_t (&set)[8], bool fastq, uint64_t* gap_slot,
                                hipStream_t st);
// *out := one past the highest non-zero entry of hist[0, cap)
hipError_t launch_hist_extent(const uint64_t* hist, uint32_t cap, uint64_t* out, hipStream_t st);
// skip_chunk (FASTA default row, anchors from launch_prep with `raw`): the nominal chunk size; 0: read every byte
hipError_t launch_stats(bool fastq, bool all, bool dpp, int blocks, const uint8_t* buf, uint64_t n,
                        const uint64_t* anchors, uint32_t nranges, uint32_t* queue, const StatsDev& D,
                        hipStream_t st, bool a_dense = false, uint64_t skip_chunk = 0);  // a_dense: FASTQ -a on the dense path (BSK_STATS_A=dense)
int stats_max_blocks_per_cu(bool fastq, bool all, bool dpp, bool a_dense = false);
hipError_t launch_stream_read(int blocks, const uint8_t* buf, uint64_t n, uint64_t chunk, uint32_t nranges,
                              uint32_t* queue, uint32_t* sink, hipStream_t st);
hipError_t launch_scan_selftest(bool dpp, const uint32_t* in, uint32_t* out, hipStream_t st);

}  // namespace bsk
