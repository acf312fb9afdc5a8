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
hipError_t launch_group_sort(void* tmp, size_t tmp_bytes, const uint64_t* in, uint64_t* out, uint64_t m, hipStream_t st);
// ord[i] := 1 + number of earlier records of the same group, for the m sorted entries (ord is zero elsewhere)
hipError_t launch_group_ordinals(const uint64_t* sorted, uint64_t m, uint32_t* ord, hipStream_t st);

}  // namespace bsk
