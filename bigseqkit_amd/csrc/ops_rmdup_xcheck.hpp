// `rmdup` across ranks, round 6: RmDupCheck's text comparison (/root/reference/bigseqkit-lib/rmdup.go:193-211) for the
// duplicates whose survivor lives on ANOTHER rank.  The owner of a key group answers every tuple with the global index of
// the group's survivor; a duplicate whose survivor is not in its own shard sends its subject text there (a 24-byte request
// + the bytes), the survivor's rank compares and answers one byte.  Records whose text differs from their survivor's
// ("flagged": two subjects under one (k1, k2) -- about N^2 / 2^129 -- or the tests' masked keys) are settled exactly among
// themselves on the host (ops_host_rmdup.cpp).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"
#include "ops_rmdup.hpp"

namespace bsk {

constexpr int XCHECK_MAX_WORLD = 64;

// rank r holds the global records [base[r], base[r + 1])
struct XRanks {
    uint64_t base[XCHECK_MAX_WORLD + 1];
    uint32_t world, rank;
};
// requests / text bytes that arrived from rank p begin at req_start[p] / byte_start[p] of the receive buffers
struct XFrom {
    uint64_t req_start[XCHECK_MAX_WORLD + 1];
    uint64_t byte_start[XCHECK_MAX_WORLD + 1];
    uint32_t world;
};
// a request: {global index of the survivor, offset of the text in the segment of its destination, len | local record << 32}
constexpr uint32_t XREQ_WORDS = 3;

// per destination rank: how many duplicates of this shard have their survivor there, and the bytes of their subjects
// (counts / bytes: u64[XCHECK_MAX_WORLD] each, zeroed by the caller).  send / reply / surv in SEND order (n tuples)
hipError_t launch_x_count(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint64_t* send,
                          const uint8_t* reply, const uint64_t* surv, uint64_t n, const XRanks& R, uint64_t* counts, uint64_t* bytes,
                          hipStream_t st);
// the requests (grouped by destination, cur_req = exclusive prefix of the counts) and where each subject goes
// (cur_bytes zeroed: offsets inside the destination's segment) -- the same traversal as launch_x_count
hipError_t launch_x_place(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint64_t* send,
                          const uint8_t* reply, const uint64_t* surv, uint64_t n, const XRanks& R, uint64_t* cur_req, uint64_t* cur_bytes,
                          uint64_t* req, hipStream_t st);
// the subjects of the m requests into text (seg_base[r] = first byte of destination r's segment): 4 lanes per request
hipError_t launch_x_copy(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint64_t* req,
                         uint64_t m, const XRanks& R, const XFrom& seg, uint8_t* text, hipStream_t st);
// survivor side: verdict[j] = 1 when the text of request j equals the subject of the local record it names, 0 when it differs
// (-i: case-folded), 2 when the request names no record of this shard (a protocol error: status |= ERR_HASH_COLLISION)
hipError_t launch_x_compare(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint64_t* req_in,
                            uint64_t m, const XFrom& from, const uint8_t* text_in, uint64_t base, uint8_t* verdict, uint64_t* status,
                            hipStream_t st);
// sender side: the records whose request came back "differs" go to the flagged list (list[0] = count, entries from list[1];
// entries beyond cap are counted but dropped)
hipError_t launch_x_apply(const uint64_t* req, const uint8_t* verdict, uint64_t m, uint32_t* list, uint32_t cap, hipStream_t st);
// the duplicates whose survivor is in the SAME shard and whose subject differs from it, listed the same way (the slow,
// listing twin of k_rmdup_verify_fastq: it runs only after that kernel raised its flag, or for subjects that kernel does not read)
hipError_t launch_x_local_list(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint64_t* send,
                               const uint8_t* reply, const uint64_t* surv, uint64_t n, uint64_t base, uint32_t* list, uint32_t cap,
                               uint64_t* n_pairs, hipStream_t st);
// single-GPU twin: duplicates (first[i] != i) whose subject differs from record first[i]'s
hipError_t launch_x_first_list(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint32_t* first,
                               uint32_t* list, uint32_t cap, hipStream_t st);
// subjects of listed records: len[j] = bytes of record list[j]'s subject; then their bytes at out + off[j]
hipError_t launch_x_subject_len(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint32_t* list,
                                uint32_t m, uint32_t* len, hipStream_t st);
hipError_t launch_x_subject_copy(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint32_t* list,
                                 const uint64_t* off, uint32_t m, uint8_t* out, hipStream_t st);
// out_len[list[j]] = the formatted size of that record (a flagged record that is the first of its text survives after all)
hipError_t launch_x_resurrect(const RecordTable& t, const RmDupParams& P, const uint32_t* list, uint32_t m, uint32_t* out_len, hipStream_t st);

}  // namespace bsk
