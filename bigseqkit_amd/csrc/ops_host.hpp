#pragma once
#include <hip/hip_runtime_api.h>

#include "../../include/bsk.h"
#include "ctx.hpp"

namespace bsk {
int kernel_error_to_status(bsk_ctx* c, uint64_t flags);
int build_index(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st);
struct FilterDev;
// not an API status: the fused pattern filter gave up (stream_filter.hpp) and the caller must build the full table
constexpr int BSK_ERR_FILTER_FALLBACK = -1000;
// internal: the head of a FASTQ shard shows records wrapped over several lines; the caller rewrites the shard
// (normalize_multiline_fastq) and runs the operator on the 4-line text
constexpr int BSK_ERR_MULTILINE_FASTQ = -1001;
bool fastq_head_multiline(const uint8_t* h, size_t hb);
// head / tail of the call's device shard in the context's pinned sample (c->h_head, c->head_len): one copy per call and shard
int sample_head(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st);
// kernel flags (stream_stats.hpp) with which the strict 4-line reader gives a FASTQ shard up: bad header 1, bad '+' line 2,
// unmatched lengths 4, truncated 8, a range that ends inside a record 16 -- a shard wrapped further down than its head
// raises one of them and is then read by the multi-line reader (c->last_kernel_flags)
constexpr uint64_t STRICT_FASTQ_FLAGS = 1u | 2u | 4u | 8u | 16u;
// d_out == null: c->table := where the records of the text begin (start[] only), *n_out := the byte behind the last one
int normalize_multiline_fastq(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st, const uint8_t** d_out, size_t* n_out);
int build_index_filtered(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, const FilterDev* F);
// the record table and, with hash != null (unfiltered FASTQ), the two keys of every record's sequence in c->d_keys / c->d_keys2
// k2 = false: k1 alone (the caller compares the bytes; c->d_keys2 is not written); group: k1 is the chain-free grouping key of
// hash_dev.hpp instead of XXH64 (only with k2 = false: a key whose value nothing but the grouping sees)
struct HashReq { bool fold; bool k2 = true; bool group = false; int mode() const { return k2 ? 1 : (group ? 2 : 0); } };
int build_index_light(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st);  // FASTA, from the '>' bytes alone (translate)
int build_index_ex(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, const FilterDev* F, const HashReq* hash);
void validate_seq_opts(bsk_ctx* c);
int seq_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
void validate_grep_opts(bsk_ctx* c);
int grep_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
void validate_subseq_opts(bsk_ctx* c);
int subseq_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
void validate_translate_opts(bsk_ctx* c);
int translate_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
void validate_rmdup_opts(bsk_ctx* c);
int rmdup_finish(bsk_ctx* c);
int rmdup_dist_keys(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, uint64_t* n_records);
int rmdup_dist_pack(bsk_ctx* c, uint64_t base, int world, uint64_t* d_send, uint64_t* counts, hipStream_t st);
int rmdup_dist_resolve(bsk_ctx* c, const uint64_t* d_tuples, uint64_t m, uint8_t* d_keep, hipStream_t st, uint64_t* d_surv = nullptr);
int rmdup_dist_emit(bsk_ctx* c, const uint64_t* d_send, const uint8_t* d_reply, uint64_t base, hipStream_t st, bsk_out* out,
                    const uint64_t* d_surv_reply = nullptr);
// round 6: the text comparison of duplicates whose survivor lives on another rank (ops_host_rmdup.cpp, include/bsk.h)
int rmdup_dist_xpack(bsk_ctx* c, const uint64_t* d_send, const uint8_t* d_reply, const uint64_t* d_surv, uint64_t base,
                     const uint64_t* rank_base, int world, uint64_t* req_cnt, uint64_t* byte_cnt, void** d_req, void** d_text, hipStream_t st);
int rmdup_dist_xcompare(bsk_ctx* c, const uint64_t* d_req_in, const uint64_t* req_from, const uint8_t* d_text_in, const uint64_t* bytes_from,
                        int world, uint8_t* d_verdict, hipStream_t st);
int rmdup_dist_xapply(bsk_ctx* c, const uint8_t* d_verdict_back, uint64_t* n_flagged, uint64_t* pairs_compared, hipStream_t st);
int rmdup_dist_flagged_get(bsk_ctx* c, void* buf, size_t cap, size_t* need, hipStream_t st);
int rmdup_dist_flagged_settle(bsk_ctx* c, const void* all, size_t n, hipStream_t st);
int rmdup_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
void validate_locate_opts(bsk_ctx* c);
int locate_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
void validate_records_opts(bsk_ctx* c);
int range_resolve(bsk_ctx* c, int64_t n_records);
int fq2fa_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
int records_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
int rename_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
void validate_sort_opts(bsk_ctx* c);
int sort_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
void validate_faidx_opts(bsk_ctx* c);
int faidx_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
int pair_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, size_t n_first, int format, hipStream_t st, bsk_out* outs);
void validate_common_opts(bsk_ctx* c);
int common_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, const uint64_t* file_ends, uint32_t nfiles, int format,
                      hipStream_t st, bsk_out* out);
int concat_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, size_t n_first, int format, hipStream_t st, bsk_out* out);
int faidx_query_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out);
void store_drainer_free(bsk_ctx* c);  // store.cpp
int ensure_out(bsk_ctx* c, uint64_t bytes);
int ensure_record_scratch(bsk_ctx* c);
Alphabet partition_alphabet(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, int* rc);
}  // namespace bsk
