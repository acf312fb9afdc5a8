// `sort` (driver bigseqkit/sort.go:91-147, executor bigseqkit-lib/sort.go:38-166): the key of every
// record -- sequence length (-l), non-gap bases (-b), ID, full name (-n) or a sequence prefix (-s), lower-cased with -i
// -- and a STABLE device radix sort (rocPRIM) of (key, record index): one pass for the integer keys, an LSD sweep over
// the 8-byte chunks of the string keys.  The records are then emitted through the `seq` emit kernels at the offsets
// of the sorted order.  Natural order (-N): the keys are first rewritten so that byte order is natural order.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"
#include "ops_translate.hpp"  // TextTableH

namespace bsk {

struct SortParams {
    int fastq;
    int mode;            // 0 ID, 1 full name, 2 sequence prefix, 3 length, 4 non-gap bases
    int ignore_case;
    int id_mode;
    uint32_t prefix_len; // -s: bytes of the sequence that count (0: all)
    uint32_t gap_set[8]; // -b
    const uint8_t* buf_end;
    // natural order (-N, modes 0 / 1): the keys are first rewritten so that byte order == natural order (k_sort_natkeys)
    const uint8_t* nat;       // transformed keys, or null
    const uint64_t* nat_off;  // [n + 1]
};

// natural order: nat_len[i] = bytes of the transformed key of record i;  then the keys themselves at nat_off[i]
hipError_t launch_sort_natlen(const uint8_t* buf, const RecordTable& t, const SortParams& P, uint32_t* nat_len, hipStream_t st);
hipError_t launch_sort_natkeys(const uint8_t* buf, const RecordTable& t, const SortParams& P, const uint64_t* nat_off, uint8_t* nat,
                               hipStream_t st);
// perm[i] = i
hipError_t launch_sort_iota(uint32_t* perm, uint64_t n, hipStream_t st);
// key_len[i] = bytes of the string key of record i; *max_len = max (zeroed by the caller).  Modes 0..2.
hipError_t launch_sort_keylen(const uint8_t* buf, const RecordTable& t, const SortParams& P, uint32_t* key_len,
                              uint32_t* max_len, hipStream_t st);
// keys[j] = bytes [8 chunk, 8 chunk + 8) of the key of record perm[j], big-endian, zero padded
hipError_t launch_sort_chunk(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const SortParams& P,
                             const uint32_t* key_len, const uint32_t* perm, uint32_t chunk, uint64_t* keys, hipStream_t st,
                             uint64_t count = ~0ull /* entries of perm; default: all records */);
// long keys: after the order by the two leading chunks only the positions whose 16 key bytes equal a neighbour's need the
// rest of the key.  tied / start flags of every position; the tied positions gathered (rank = exclusive scan of tied, run =
// exclusive scan of start); the re-ordered subset put back
hipError_t launch_sort_tie_flags(const uint64_t* k0, const uint64_t* k1, uint64_t n, uint32_t* tied, uint32_t* start, hipStream_t st);
hipError_t launch_sort_tie_gather(const uint32_t* tied, const uint64_t* rank, const uint64_t* run, const uint32_t* start,
                                  const uint32_t* perm, uint64_t n, uint32_t* sub_pos, uint32_t* sub_perm, uint64_t* run_of /* by record */, hipStream_t st);
hipError_t launch_sort_gather_keys(const uint64_t* by_record, const uint32_t* perm, uint64_t m, uint64_t* keys, hipStream_t st);
hipError_t launch_sort_tie_scatter(const uint32_t* sub_pos, const uint32_t* sub_perm, uint64_t m, uint32_t* perm, hipStream_t st);
// modes 3, 4: keys[i] = length / non-gap bases of record i (as 64-bit keys)
hipError_t launch_sort_intkeys(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const SortParams& P,
                               uint64_t* keys, hipStream_t st);
hipError_t sort_pairs_temp_bytes(uint64_t n, size_t* bytes);
// stable; bits [0, end_bit) of the keys take part
hipError_t launch_sort_pairs(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                             uint32_t* vout, uint64_t n, bool descending, int end_bit, hipStream_t st);
// bits [begin_bit, end_bit) of the keys take part (the radix-bucket pass of rmdup sorts by 16 bits only)
hipError_t sort_pairs_bits_temp_bytes(uint64_t n, int begin_bit, int end_bit, size_t* bytes);
hipError_t launch_sort_pairs_bits(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                                  uint32_t* vout, uint64_t n, int begin_bit, int end_bit, hipStream_t st);
// ... with the values 0 .. n - 1 as input (no iota array)
hipError_t sort_pairs_bits_iota_temp_bytes(uint64_t n, int begin_bit, int end_bit, size_t* bytes);
hipError_t launch_sort_pairs_bits_iota(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, uint32_t* vout, uint64_t n,
                                       int begin_bit, int end_bit, hipStream_t st);
// len_perm[j] = out_len[perm[j]];  out_off[perm[j]] = off_perm[j]
hipError_t launch_sort_gather(const uint32_t* out_len, const uint32_t* perm, uint64_t n, uint32_t* len_perm, hipStream_t st);
hipError_t launch_sort_scatter(const uint64_t* off_perm, const uint32_t* perm, uint64_t n, uint64_t* out_off, hipStream_t st);

}  // namespace bsk
