// Ordinal of a record inside its group of equal keys, in file order (`rename`, bigseqkit-lib/rename.go:95-131:
// the 2nd, 3rd, ... record of an ID becomes ID_1, ID_2, ...).  Groups come from the rmdup hash table
// (group[i] = index of the first record with the subject of record i); only the records that are not the first
// of their group are compacted, sorted by (group, index) with rocPRIM's device radix sort and ranked.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace bsk {

// list := (group[i] << 32 | i) of every i with group[i] != i (any order); *count := how many (zeroed by the caller)
hipError_t launch_group_compact(const uint64_t* group, uint64_t n, uint64_t* list, uint64_t* count, hipStream_t st);
// bytes of temporary storage rocprim::radix_sort_keys needs for m keys
hipError_t group_sort_temp_bytes(uint64_t m, size_t* bytes);
// records: number of records (groups are < records; 0 = unknown, all 64 bits); index_ordered: `in` is in index order
hipError_t launch_group_sort(void* tmp, size_t tmp_bytes, const uint64_t* in, uint64_t* out, uint64_t m, hipStream_t st,
                             uint64_t records = 0, bool index_ordered = false);
// ord[i] := 1 + number of earlier records of the same group, for the m sorted entries (ord is zero elsewhere)
hipError_t launch_group_ordinals(const uint64_t* sorted, uint64_t m, uint32_t* ord, hipStream_t st);

// ---- pair (bigseqkit-lib/pair.go:86-121): records of two files grouped by ID; the k-th record of an ID in file 1
// goes with the k-th of file 2.  Records [0, first2) belong to file 1.
// list[i] = group[i] << 32 | i for every record
hipError_t launch_group_all(const uint64_t* group, uint64_t n, uint64_t* list, hipStream_t st);
// *count += records whose start is below x (zeroed by the caller)
hipError_t launch_count_below(const uint64_t* start, uint64_t n, uint64_t x, uint64_t* count, hipStream_t st);
// state[i]: 1 paired (file 1), 2 paired (file 2), 3 unpaired (file 1), 4 unpaired (file 2); partner[i] for 1 and 2
hipError_t launch_pair_classify(const uint64_t* sorted, uint64_t n, uint32_t first2, uint8_t* state, uint32_t* partner,
                                hipStream_t st);
// totals[k] += bytes, totals[4 + k] += records of state k + 1 (zeroed by the caller)
hipError_t launch_pair_totals(const uint8_t* state, const uint32_t* fmt_len, uint64_t n, uint64_t* totals, hipStream_t st);
// len_k[i] = state[i] == k ? fmt_len[i] : 0
hipError_t launch_pair_select(const uint8_t* state, const uint32_t* fmt_len, uint64_t n, uint8_t k, uint32_t* len_k, hipStream_t st);
// w[i] = state[i] == 1 ? fmt_len[partner[i]] : 0   (file order of the paired file-1 records = order of both outputs)
hipError_t launch_pair_partner_len(const uint8_t* state, const uint32_t* partner, const uint32_t* fmt_len, uint64_t n,
                                   uint32_t* w, hipStream_t st);
// off2[partner[i]] = offw[i] for state[i] == 1
hipError_t launch_pair_partner_off(const uint8_t* state, const uint32_t* partner, const uint64_t* offw, uint64_t n,
                                   uint64_t* off2, hipStream_t st);

// ---- common: the shard is k files back to back, file_ends[f] = one past the last byte of file f (device array).
// masks[g] |= 1 << file(i) for g = group[i] (masks zeroed by the caller, k <= 64)
hipError_t launch_common_masks(const uint64_t* group, const uint64_t* start, uint64_t n, const uint64_t* file_ends, uint32_t k,
                               uint64_t* masks, hipStream_t st);
// len[i] = fmt_len[i] when i is the first record of its group, lies in file 0 and the group has members in all k files
hipError_t launch_common_select(const uint64_t* group, const uint64_t* start, uint64_t n, const uint64_t* file_ends, uint32_t k,
                                const uint64_t* masks, uint32_t* len, hipStream_t st);

// ---- grep --delete-matched: a[i] = b[i] ? a[i] : 0;  keep only the first non-zero entry of a
hipError_t launch_mask_u32(uint32_t* a, const uint32_t* b, uint64_t n, hipStream_t st);
// *first = min index with a[i] != 0 (set to ~0 by the caller), then every other entry is zeroed
hipError_t launch_first_nonzero(const uint32_t* a, uint64_t n, uint64_t* first, hipStream_t st);
hipError_t launch_keep_only(uint32_t* a, uint64_t n, const uint64_t* first, hipStream_t st);
// grep --delete-matched with several patterns (bigseqkit-lib/grep.go:463-511): per-pattern hit bits, the first record that
// matches a remaining pattern, and the final mask of the selected records (bit 31)
hipError_t launch_or_bit(uint32_t* masks, const uint32_t* hit, uint64_t n, uint32_t bit, hipStream_t st);
hipError_t launch_first_masked(const uint32_t* masks, uint64_t n, uint32_t remaining, uint64_t from, uint64_t* first, hipStream_t st);
hipError_t launch_keep_selected(uint32_t* out_len, const uint32_t* masks, uint64_t n, hipStream_t st);

}  // namespace bsk
