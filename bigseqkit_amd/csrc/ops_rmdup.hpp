// `rmdup`: XXH64 key per record -> open-addressing table keyed by the hash (first
// occurrence wins) -> exact verification -> survivors emitted in file order.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"
#include "ops_translate.hpp"  // TextTableH

namespace bsk {

struct RmDupParams {  // RmDupPrepare / RmDupCheck options (bigseqkit-lib/rmdup.go:23-90)
    int fastq;
    int by_seq, by_name, ignore_case;
    int id_mode;
    int line_width;
    const uint8_t* buf_end;  // one past the shard, or null
    // sequences of at least this many bases on contiguous text are left out by k_rmdup_hash and hashed by a wave each
    // (k_rmdup_hash_long); 0: none
    uint32_t hash_long_min;
};

constexpr uint32_t ERR_HASH_COLLISION = 512u;
constexpr uint32_t ERR_BUCKET_OVERFLOW = 1u << 21;  // a radix bucket with too many distinct keys for its LDS table

// keys[i] = XXH64(subject i, seed 0); keys2 (may be null) = the same with another seed (multi-GPU verification key)
hipError_t launch_rmdup_hash(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const TextTableH& tt,
                             const RmDupParams& P, uint64_t* keys, uint64_t* keys2, hipStream_t st);
// the records of `long_list` (launch_find_long on l_seq with P.hash_long_min), one wave each; after launch_rmdup_hash
hipError_t launch_rmdup_hash_long(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const TextTableH& tt,
                                  const RmDupParams& P, uint64_t* keys, uint64_t* keys2, const uint32_t* long_list,
                                  uint64_t long_count, hipStream_t st);
// table: `cap` slots (power of two) of 16 bytes {key, ~first record}, zeroed by the caller (key_table)
hipError_t launch_rmdup_insert(const uint64_t* keys, uint64_t n, uint64_t base_index, uint64_t* table, uint64_t cap,
                               hipStream_t st);
hipError_t launch_rmdup_resolve(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P,
                                const uint64_t* keys, const uint64_t* table, uint64_t cap, uint32_t* out_len,
                                uint64_t* status, hipStream_t st);

// the same and, in the same pass, keys[i] := first record of i's group, has_dup[first] := 1 (has_dup zeroed by the caller)
hipError_t launch_rmdup_resolve_group(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P,
                                      uint64_t* keys, const uint64_t* table, uint64_t cap, uint32_t* out_len,
                                      uint64_t* status, uint8_t* has_dup, hipStream_t st);

// the (key, record) pairs are bucketed by the low RMDUP_BUCKET_BITS bits of the key: the bit range of the host's sort call
// and the bucket arithmetic of the kernels (ops_rmdup.hip) are this one constant
constexpr uint32_t RMDUP_BUCKET_BITS = 16;
// grouping by radix buckets: skeys / sidx = the (key, record) pairs sorted by the low 16 key bits (launch_sort_pairs_bits),
// bstart: scratch [65 537]; first[] must hold iota and receives, for every duplicate, the lowest record with its key
// k2 != null (second keys by record): a duplicate must agree with the first record of its key in k2 as well; the records
// that do not are listed in ovf[1..] (ovf[0] = their number, zeroed by the caller; entries beyond ovf_cap are dropped)
// `rmdup -s` on FASTQ: output sizes from first[] (a record survives iff first[i] == i) and the byte comparison of every
// duplicate's sequence with its survivor's (-i: case-folded); a difference raises ERR_HASH_COLLISION
hipError_t launch_rmdup_verify_fastq(const uint8_t* buf, const RecordTable& t, const RmDupParams& P, const uint32_t* first,
                                     uint32_t* out_len, uint64_t* status, hipStream_t st);
// round 5: the same comparison AND the output offsets, segment sources and per-tile first segments of the survivors in one
// pass (decoupled look-back over blocks of 2 048 records).  chain: rmdup_place_blocks(n) words, ticket: one word, both
// zeroed by the caller; fin = the control block's FIN words ([0] total, [1] survivors, [2] records >= long_thresh bytes,
// [4] records left to the record-wise emit; [1], [2], [4] zeroed by the caller); first4k as launch_seg_first writes it
uint64_t rmdup_place_blocks(uint64_t n);
hipError_t launch_rmdup_place(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const RmDupParams& P, const uint32_t* first,
                              uint64_t* chain, uint32_t* ticket, uint64_t* out_off, uint64_t* seg_src, uint32_t* first4k, uint64_t* fin,
                              uint32_t long_thresh, uint64_t* status, hipStream_t st);
hipError_t launch_bucket_dedupe(const uint64_t* skeys, const uint32_t* sidx, uint64_t n, uint32_t* bstart, uint32_t* first,
                                uint64_t* status, hipStream_t st, const uint64_t* k2 = nullptr, uint32_t* ovf = nullptr,
                                uint32_t ovf_cap = 0, bool have_bstart = false /* bstart comes from launch_bucket_pass */);
// the pairs (key, record) of keys[0, n) by bucket (key & 0xFFFF), in no particular order inside a bucket: a 16-bit histogram
// (hist: 65 536 words of scratch, zero again afterwards), its prefix sums (bstart: 65 537 words) and one scatter;
// first[i] := i on the way.  n < 2^32.
hipError_t launch_bucket_pass(const uint64_t* keys, uint64_t n, uint32_t* hist, uint32_t* bstart, uint32_t* first, uint64_t* skeys,
                              uint32_t* sidx, hipStream_t st);
// out_len[i] = formatted size of record i if first[i] == i, else 0 (the keys decided; no text is read)
hipError_t launch_rmdup_sizes(const RecordTable& t, const RmDupParams& P, const uint32_t* first, uint32_t* out_len, hipStream_t st);
// out[2 j], out[2 j + 1] = k1, k2 of record list[j];  dst[idx[j]] = val[j]
hipError_t launch_gather_keys(const uint32_t* list, uint32_t m, const uint64_t* k1, const uint64_t* k2, uint64_t* out, hipStream_t st);
// owner side of the multi-GPU exchange: k1[i], k2[i] of tuple i (24-byte rows: k1, k2, global record number);
// keep[i] = tuple i carries the lowest record number of its group (first[i] = one member of the group; gmin: m words scratch)
hipError_t launch_split_tuples(const uint64_t* tuples, uint64_t m, uint64_t* k1, uint64_t* k2, hipStream_t st);
hipError_t launch_keep_lowest(const uint64_t* tuples, const uint32_t* first, uint64_t m, uint64_t* gmin, uint8_t* keep, hipStream_t st,
                              uint64_t* surv = nullptr /* [m]: global index of every tuple's survivor */);
// multi-GPU emit: first_of[] for the byte comparison of the duplicates whose survivor lives in the same shard (*n_local += their number)
hipError_t launch_dist_first(const uint64_t* send, const uint8_t* reply, const uint64_t* surv, uint64_t n, uint64_t base, uint32_t* first_of,
                             uint64_t* n_local, hipStream_t st);
// tests: keys[i] &= mask (forces distinct subjects under one key)
hipError_t launch_mask_keys(uint64_t* keys, uint64_t n, uint64_t mask, hipStream_t st);
hipError_t launch_scatter_u32(const uint32_t* idx, const uint32_t* val, uint32_t m, uint32_t* dst, hipStream_t st);
// out_len from first[] (a duplicate is byte-compared with its survivor); keys_group != null: also keys_group[i] := first[i]
// and has_dup[first] := 1 (the -d / -D side outputs)
hipError_t launch_rmdup_resolve_first(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P,
                                      const uint32_t* first, uint64_t* keys_group, uint32_t* out_len, uint64_t* status,
                                      uint8_t* has_dup, hipStream_t st);

// -d / -D side outputs: keys[i] := survivor index of record i; sizes of the removed records' text and of the
// "<20-digit group>\t<ID>\n" rows of all members of groups of two or more; the rows themselves
hipError_t launch_rmdup_group(uint64_t n, uint64_t* keys, const uint64_t* table, uint64_t cap, uint8_t* has_dup,
                              hipStream_t st);
hipError_t launch_rmdup_side_sizes(const uint8_t* buf, const RecordTable& t, const RmDupParams& P, const uint64_t* group,
                                   const uint8_t* has_dup, uint32_t* dup_len, uint32_t* row_len, hipStream_t st);
hipError_t launch_rmdup_rows(const uint8_t* buf, const RecordTable& t, const RmDupParams& P, const uint64_t* group,
                             const uint32_t* row_len, const uint64_t* row_off, uint8_t* out, hipStream_t st);

// multi-GPU rmdup: tuples (key, second key, global index) routed to owner = key % world (ops_rmdup.hip)
hipError_t launch_rmdup_count_owner(const uint64_t* keys, uint64_t n, uint32_t world, uint64_t* counts, hipStream_t st);
hipError_t launch_rmdup_pack(const uint64_t* keys, const uint64_t* keys2, uint64_t n, uint64_t base, uint32_t world,
                             uint64_t* cursor, uint64_t* send, hipStream_t st);
// table_keys / table_k2 zeroed, table_first 0xFF-filled by the caller; keep[p] = tuple p is the first of its key
hipError_t launch_rmdup_own(const uint64_t* tuples, uint64_t m, uint64_t* table_keys, uint64_t* table_first,
                            uint64_t* table_k2, uint64_t cap, uint8_t* keep, uint64_t* status, hipStream_t st, uint64_t* surv = nullptr);
hipError_t launch_rmdup_apply(const RecordTable& t, const RmDupParams& P, const uint64_t* send, const uint8_t* reply,
                              uint64_t base, uint32_t* out_len, hipStream_t st);

}  // namespace bsk
