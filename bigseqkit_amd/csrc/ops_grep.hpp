// `grep` on the record table: match kernel -> sizes -> scan -> k_seq_emit (full record).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {

struct GrepParams {  // Grep options after Before() (bigseqkit-lib/grep.go:41-253), exact patterns
    int fastq;
    int by_seq, by_name, invert, ignore_case, circular;
    int region_on, region_start, region_end;
    int both_strands;        // search the reverse complement too (grep.go:432-448)
    int strand_only;         // 0: as both_strands says; 1: '+' only; 2: '-' only (hit bits per pattern AND strand)
    int id_mode;             // 0 default ID regexp, 1 --id-ncbi
    int line_width;          // of the emitted record (0 for FASTQ)
    int npat;                // patterns; with both_strands the reverse-complemented copies follow
    const uint8_t* pat;      // concatenated pattern bytes (already lower-cased when ignore_case)
    const uint32_t* pat_off; // [npat_total + 1]
    // class patterns (-d, -m): 8 dwords per pattern position, same offsets as `pat` (pattern_match_dev.hpp)
    int general;
    int max_mm;
    const uint32_t* cls;
    int sa_ok;               // every pattern <= 64 positions, <= 8 (strand, pattern) tables, max_mm <= 3, not circular: k_grep_shiftand
    // ID / name pattern set (many patterns, e.g. -f ids.txt): open addressing on fnv1a64, verified by bytes
    // -r: Glushkov programs (regex_nfa.hpp), `npat` of them, in device memory; comp: complement map for the '-' strand
    const struct RegexProgram* regex;
    // -r with an expression the automaton does not take (\b, > 64 positions): thread-list programs (regex_vm.hpp) instead
    const struct VmProgram* vm;
    const uint8_t* comp;
    // sequences of at least long_thresh bases are searched by whole blocks (k_grep_seq<.., LONG>)
    const uint32_t* long_list;  // their record indices
    uint32_t* long_hit;         // one flag per entry of long_list (zeroed by the caller)
    uint64_t long_count, long_max;
    uint32_t long_thresh;
    const uint8_t* buf_end;    // one past the shard (lets the ID search read 16 bytes at a time)
    const uint64_t* set_keys;  // null: linear scan over the patterns
    const uint32_t* set_idx;
    uint64_t set_mask;
};

struct TextTableH;
hipError_t launch_grep_match(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const TextTableH* tt,
                             const GrepParams& P, uint32_t* out_len, hipStream_t st, uint64_t avg_record_bytes = 0);

}  // namespace bsk
