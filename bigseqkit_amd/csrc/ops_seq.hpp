// `seq` on the record table: size pass -> exclusive scan -> emit pass.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {

struct SeqParams {  // SeqTransform options after Before() (bigseqkit-lib/seq.go:28-79), device-friendly
    int fastq;
    int fasta_out;                          // fq2fa: a FASTQ record is written as FASTA ('>' marker; print_qual is 0)
    int print_name, print_seq, print_qual;  // seq.go:151-163 (constant per partition)
    int qual_only;                          // opts.Qual: no "+\n" before the quality
    int only_id;                            // -i
    int id_mode;                            // 0 default regexp (helper.go:329-357), 1 --id-ncbi
    int reverse;
    int use_lut;                            // complement / dna2rna / rna2dna / case folded into one 256-byte map
    int remove_gaps;
    uint32_t gap_set[8];                    // 256-bit set
    int gap_lt64;                           // every gap letter is below 64 ('-', '.', ' ', '*'): 16 bytes at a time can be ruled out
    int line_width;                         // effective (0 for FASTQ, -s, -q)
    int min_len, max_len;                   // > 0 enables (seq.go:88-89)
    double min_qual, max_qual;              // > 0 enables (seq.go:90-91)
    int qual_base;
    int validate;                           // SeqParser validation (helper.go:304-306)
    int validate_len;
    uint32_t valid_set[8];
    int region_on, region_start, region_end; // Seq.SubSeq(start, end) applied to seq and qual (subseq -r)
    const uint8_t* lut;                     // device, 256 bytes
    const double* qual_err;                 // device, 256 doubles: 10^(-(q-base)/10) indexed by the raw byte
    // FASTA text view (text_dev.hpp): random access into wrapped sequences; null = contiguous records only on the
    // parallel path (others take the sequential per-record walk)
    const uint32_t* text_w;
    const uint64_t* lin_off;
    const uint8_t* lin;
    // subseq --gtf / --bed (bigseqkit-lib/subseq.go:319-526): the record's lower-cased ID selects ONE feature
    // (the first of that name); region, strand and the new header come from it
    int feat_on;
    int feat_query;                         // faidx region queries: names compared as they are unless feat_fold, region by
    int feat_fold;                          // SubLocation (negative positions), records whose region is empty are skipped
    const uint64_t* fset_keys;              // open-addressing set of lower-cased names (fnv1a64), 0 = empty
    const uint32_t* fset_idx;
    uint64_t fset_mask;
    const uint8_t* fname;                   // lower-cased names, for verification
    const uint32_t* fname_off;
    const int64_t* f_s;                     // flank-adjusted 1-based start / end, not yet clamped to the record
    const int64_t* f_e;
    const uint8_t* f_minus;                 // feature on the '-' strand: reverse complement (qualities reversed)
    const uint8_t* fsuffix;                 // "_<start>-<end>:<strand><flank> <label>" appended to the ID
    const uint32_t* fsuffix_off;
    const uint8_t* comp;                    // complement map of the shard's alphabet
    // rename (bigseqkit-lib/rename.go:118-121): ord[i] > 0 => header = ID + "_" + ord + " " + Desc
    const uint32_t* ren_ord;
    const uint8_t* buf_end;                 // one past the shard, or null (the ID search then reads byte by byte)
    // records with a very large output (chromosomes): written by whole blocks, see k_seq_emit<.., LONG>
    const uint32_t* long_list;              // their indices (launch_find_long), or null
    uint64_t long_count, long_max;          // how many, and the largest output size
    uint32_t long_thresh;                   // output bytes from which a record is 'long' (0: none are)
    // records whose output is written by the segmented copy (ops_segcopy.hip): seg_src[i] != 0; null = none
    const uint64_t* seg_src;
};

constexpr uint32_t SEQ_LONG_THRESH = 1u << 20;  // output bytes from which a record is 'long'

constexpr uint32_t ERR_INVALID_LETTER = 128u;

// Seq.SubSeq / SubLocation (shenwei356/bio, not in tree; pinned by the region table at
// /root/reference/bigseqkit-cli/helper.go:348-361): 1-based inclusive, negative = from the end.
// Returns the 0-based half-open [b, e); b == e when empty.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void sub_location(uint32_t length, int start, int end, uint32_t* b, uint32_t* e) {
    *b = *e = 0;
    if (length == 0) return;
    long long L = (long long)length, s = start, t = end;
    if (s < 0) { s = L + s + 1; if (s < 1) s = 1; }
    if (s == 0) s = 1;
    if (t < 0) t = L + t + 1;
    if (t > L) t = L;
    if (s > L || t < 1 || s > t) return;
    *b = (uint32_t)(s - 1);
    *e = (uint32_t)t;
}

hipError_t launch_seq_size(const uint8_t* buf, const RecordTable& t, const SeqParams& P, uint32_t* out_len,
                           uint64_t* status, hipStream_t st);
// total_bytes / records (of the output, when known) pick the lanes per record: 4 for small records, else 16
hipError_t launch_seq_emit(const uint8_t* buf, const RecordTable& t, const SeqParams& P, const uint32_t* out_len,
                           const uint64_t* out_off, uint8_t* out, hipStream_t st, uint64_t total_bytes = 0,
                           uint64_t records = 0);
// list := indices of the records with out_len >= thresh; count_max[0] := their number, [1] := max out_len
// (both zeroed by the caller).  At most 2^32 / SEQ_LONG_THRESH... the list needs one slot per such record.
hipError_t launch_find_long(const uint32_t* out_len, uint64_t n, uint32_t thresh, uint32_t* list, uint64_t* count_max,
                            hipStream_t st);
hipError_t launch_count_nonzero(const uint32_t* v, uint64_t n, uint64_t* counter, hipStream_t st);

}  // namespace bsk
