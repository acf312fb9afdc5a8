// `locate`: all match positions of exact patterns, both strands, as text rows.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"
#include "ops_translate.hpp"  // TextTableH

namespace bsk {

struct LocateParams {  // Locate options after Before() (bigseqkit-lib/locate.go:33-193), exact patterns
    int fastq;
    int ignore_case, circular, non_greedy, both_strands;
    int format;              // 0 TSV, 1 TSV without the matched column (-M), 2 GTF, 3 BED
    int id_mode;
    int npat;
    const uint8_t* pat;      // effective pattern bytes: npat forward, then npat reverse-complemented
    const uint32_t* pat_off; // [2 * npat + 1]
    const uint8_t* name;     // pattern names as given (npat)
    const uint32_t* name_off;
    // class patterns (-d, -m, -F; pattern_match_dev.hpp): 8 dwords per position, offsets as `pat`
    int general, max_mm;
    const uint32_t* cls;
    const struct RegexProgram* pre_regex;  // -r: boolean automata of the expressions (null: none); bound the matcher's walks
    const uint32_t* cand;    // count pass: null, or the records that hold an occurrence at all (k_grep_shiftand went first); the
    uint64_t ncand;          // others keep the 0 the prefilter left in out_len
    int fmi_order;           // -m / -F: all patterns on '+', then all on '-'; no +l shift of '-' coordinates (locate.go:208-391)
    int matched_lower;       // the matched column shows the lower-cased text (-i without -d)
    const uint8_t* comp;     // 256-byte complement map of the shard's alphabet ('-' strand matched column)
    // -r: the pattern column shows the expression, whose length is not the match length (null: the pattern itself)
    const uint8_t* disp;
    const uint32_t* disp_off;
    // records with at least one row: appended (any order) by the count pass, the emit pass runs over them only
    uint32_t* hit_list;
    uint64_t* hit_count;
    uint64_t nhit;
    // sequences of at least long_thresh bases (chromosomes): one WAVE per cell = (pattern, strand, chunk of LOCATE_LONG_CH
    // start positions); the cells of a record are numbered in the reference's row order, so a scan of their byte counts
    // gives every cell its place inside the record's rows
    const uint32_t* long_list;    // record indices of the long records (any order)
    const uint64_t* cellbase;     // [long_count + 1] first cell of each long record (scan of their cell counts)
    uint32_t* cell_bytes;         // [total cells] written by the count pass
    const uint64_t* cell_off;     // [total cells + 1] exclusive scan of cell_bytes
    uint64_t long_count, long_cells;  // long records, cells of all of them
    uint32_t long_thresh;
};

constexpr uint32_t LOCATE_LONG_CH = 64u * 1024u;

// count pass: out_len[i] = bytes of all rows of record i; emit pass writes them at out_off[i]
hipError_t launch_locate(bool emit, const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const TextTableH& tt,
                         const LocateParams& P, uint32_t* out_len, const uint64_t* out_off, uint8_t* out,
                         uint64_t* rows, hipStream_t st, uint64_t avg_record_bytes = 0);

struct VmProgram;
// locate -r with matches of variable length: one lane per record runs the Pike VM (count pass: out_len; emit pass: rows)
hipError_t launch_locate_vm(bool emit, const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const TextTableH& tt,
                            const LocateParams& P, const VmProgram* d_progs, uint32_t* out_len, const uint64_t* out_off,
                            uint8_t* out, uint64_t* rows, hipStream_t st, const uint32_t* list = nullptr /* records to visit (else all) */,
                            uint64_t nlist = 0);

// hit_list := indices of the records with out_len != 0 (any order), *hit_count := how many (zeroed by the caller)
// out_len of the long records := sum of their cells' bytes (after the scan of cell_bytes)
hipError_t launch_locate_long_sizes(const LocateParams& P, uint32_t* out_len, hipStream_t st);
// ncells[y] = chunks x patterns x strands of long record long_list[y]
hipError_t launch_locate_long_cells(const RecordTable& t, const LocateParams& P, uint32_t* ncells, hipStream_t st);
hipError_t launch_compact_hits(const uint32_t* out_len, uint64_t n, uint32_t* hit_list, uint64_t* hit_count, hipStream_t st);

}  // namespace bsk
