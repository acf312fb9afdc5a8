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
    // class patterns (-d, -m, -F; pattern_match.cuh): 8 dwords per position, offsets as `pat`
    int general, max_mm;
    const uint32_t* cls;
    int fmi_order;           // -m / -F: all patterns on '+', then all on '-'; no +l shift of '-' coordinates (locate.go:208-391)
    int matched_lower;       // the matched column shows the lower-cased text (-i without -d)
    const uint8_t* comp;     // 256-byte complement map of the shard's alphabet ('-' strand matched column)
    // records with at least one row: appended (any order) by the count pass, the emit pass runs over them only
    uint32_t* hit_list;
    uint64_t* hit_count;
    uint64_t nhit;
};

// count pass: out_len[i] = bytes of all rows of record i; emit pass writes them at out_off[i]
hipError_t launch_locate(bool emit, const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const TextTableH& tt,
                         const LocateParams& P, uint32_t* out_len, const uint64_t* out_off, uint8_t* out,
                         uint64_t* rows, hipStream_t st);

// hit_list := indices of the records with out_len != 0 (any order), *hit_count := how many (zeroed by the caller)
hipError_t launch_compact_hits(const uint32_t* out_len, uint64_t n, uint32_t* hit_list, uint64_t* hit_count, hipStream_t st);

}  // namespace bsk
