// Host side of `seq` (SeqTransform.Call, /root/reference/bigseqkit-lib/seq.go:94-190): the names pass and the
// size -> scan -> emit flow.
// (split off ops_host.cpp in round 3; shared helpers: ops_host_internal.hpp)  C-ABI in include/bsk.h.
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>
#include <cerrno>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/bsk.h"
#include "ctx.hpp"
#include "ops_host.hpp"
#include "ops_host_internal.hpp"
#include "ops_concat.hpp"
#include "ops_faidx.hpp"
#include "ops_grep.hpp"
#include "ops_group.hpp"
#include "ops_locate.hpp"
#include "ops_mlfq.hpp"
#include "ops_records.hpp"
#include "ops_rmdup.hpp"
#include "ops_text.hpp"
#include "ops_translate.hpp"
#include "ops_segcopy.hpp"
#include "ops_seq.hpp"
#include "ops_sort.hpp"
#include "stream_fasta_light.hpp"
#include "stream_filter.hpp"
#include "stream_names.hpp"
#include "stream_subseq.hpp"
#include "stream_rmdup.hpp"
#include "stream_stats.hpp"

namespace bsk {

// `seq -n` / `seq -n -i` on FASTQ: the names leave from the streaming pass itself (stream_names.hip) -- per-range slices
// sized from the header density of the shard head, one scan over the ranges, one gather.  BSK_ERR_FILTER_FALLBACK: a
// slice was too small (or the estimate does not fit); the caller takes the record-table path.
static int seq_names_run(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    const int blocks = std::max(1, c->num_cus * names_max_blocks_per_cu(c->use_dpp));
    uint32_t nranges = 0;
    uint64_t chunk = 0;
    int rc = prep_ranges(c, d_buf, n, /*fastq=*/true, blocks, st, &nranges, &chunk);
    if (rc != BSK_OK) return rc;
    rc = sample_head(c, d_buf, n, st);  // (the call's one head sample, pinned)
    if (rc != BSK_OK) return rc;
    const size_t hb = c->head_len;
    const uint8_t* head = c->h_head;
    if (!c->norm_active && fastq_head_multiline(head, hb)) return BSK_ERR_MULTILINE_FASTQ;
    uint64_t hdr = 0, line = 0;
    for_lines(head, hb, [&](size_t s0, size_t e0, bool) {  // (a header cut by the end of the sample counts as far as it goes)
        if ((line & 3) == 0) hdr += e0 - s0;
        ++line;
    });
    double ratio = (double)(hdr + 64) / (double)hb;
    if (const char* sc = c->tune.get("names_scale")) ratio *= atof(sc);  // tests: force the overflow -> fallback route
    uint64_t slice_cap = (uint64_t)((double)chunk * ratio * 1.25) + (c->tune.get("names_scale") ? 16 : 4096);
    slice_cap = (slice_cap + 15) & ~(uint64_t)15;
    if (slice_cap >= (1ull << 32) || slice_cap * nranges > (uint64_t)n + (64ull << 20)) return BSK_ERR_FILTER_FALLBACK;
    rc = grow(c, &c->d_slices, &c->slices_cap, slice_cap * nranges, 256);
    if (rc != BSK_OK) return rc;
    rc = grow(c, &c->d_names_aux, &c->names_aux_cap, 2 * ((uint64_t)nranges + 2), 16);
    if (rc != BSK_OK) return rc;
    NamesDev D;
    D.slices = c->d_slices;
    D.slice_cap = slice_cap;
    D.range_bytes = c->d_names_aux;
    D.range_count = c->d_range_count;
    D.status = c->d_status;
    D.only_id = o.b("OnlyId") ? 1 : 0;
    D.id_mode = id_mode_of(c);
    uint64_t* d_count_base = c->d_names_aux + nranges + 2;
    {
        Timed t(c, "k_names", st);
        HIP_TRYX(c, launch_names(c->use_dpp, blocks, d_buf, n, c->d_anchors,
                                 nranges, reinterpret_cast<uint32_t*>(c->d_anchors + (size_t)nranges + 1), D, st));
    }
    {
        Timed t(c, "k_range_scan", st);
        HIP_TRYX(c, launch_scan_small2(D.range_bytes, c->d_range_base, c->d_fin + bsk_ctx::FIN_AUX0, D.range_count, d_count_base,
                                       c->d_fin + bsk_ctx::FIN_AUX1, nranges, st));
    }
    rc = ctl_readback(c, st);  // bytes, records, status: one copy
    if (rc != BSK_OK) return rc;
    const uint64_t total = c->fin(bsk_ctx::FIN_AUX0), records = c->fin(bsk_ctx::FIN_AUX1);
    uint64_t status = c->status_word();
    if (status & ERR_CAPACITY) {
        status &= ~(uint64_t)ERR_CAPACITY;
        HIP_TRYX(c, hipMemcpyAsync(c->d_status, &status, sizeof status, hipMemcpyHostToDevice, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        if (status == 0) return BSK_ERR_FILTER_FALLBACK;
    }
    if (status) return kernel_error_to_status(c, status);
    c->table.n = 0;  // no record table was built for this shard
    if (total && slices_wanted(c))  // round 6: the per-range slices ARE the result, in order (include/bsk.h bsk_out.d_seg_*)
        return out_as_slices(c, out, D.slices, D.slice_cap, c->d_range_base, nranges, total, records, st);
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    if (total) {
        Timed t(c, "k_names_compact", st);
        HIP_TRYX(c, launch_names_compact(D, c->d_range_base, nranges, c->d_out, st));
    }
    c->table.n = 0;  // no record table was built for this shard
    out->d_data = c->d_out;
    out->len = total;
    out->records = records;
    return BSK_OK;
}

int seq_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    out->d_data = nullptr;
    out->len = 0;
    out->records = 0;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    {
        // names only, nothing that needs the sequence (length / quality filters, gap removal, letter validation): the
        // streaming pass writes them (BSK_NAMES=off keeps the record-table path)
        const char* nm = c->tune.get("names");
        const bool explicit_alphabet = c->alphabet_given();
        if (fastq && n > 0 && o.b("Name") && !o.b("Seq") && !o.b("RemoveGaps") && o.i("MinLen") <= 0 && o.i("MaxLen") <= 0 &&
            !(o.f("MinQual") > 0) && !(o.f("MaxQual") > 0) && !o.b("ValidateSeq") && !explicit_alphabet &&
            (!o.b("OnlyId") || id_mode_of(c) != 2) && !(nm && strcmp(nm, "off") == 0)) {
            const int rcn = seq_names_run(c, d_buf, n, st, out);
            if (rcn != BSK_ERR_FILTER_FALLBACK) return rcn;
        }
    }
    int rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    // ---- per-partition decisions of SeqTransform.Call (seq.go:94-125)
    Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);  // parser.t after the first record
    if (rc != BSK_OK) return rc;
    if (ab == AB_NONE) ab = AB_UNLIMIT;
    SeqParams P;
    memset(&P, 0, sizeof P);
    P.fastq = fastq;
    bool printName = true, printSeq = true, printQual = fastq;
    if (o.b("Name") && o.b("Seq")) { /* both on; printQual as is */ }
    else if (o.b("Name")) { printSeq = false; printQual = false; }
    else if (o.b("Seq")) { printName = false; printQual = false; }
    else if (o.b("Qual")) {
        if (!fastq && c->table.n > 0) {
            c->set_error("FASTA format has no quality. So do not just use flag -q (--qual)");
            return BSK_ERR_FORMAT;
        }
        printName = false; printSeq = false; printQual = true;
    }
    P.print_name = printName; P.print_seq = printSeq; P.print_qual = printQual;
    P.qual_only = o.b("Qual");
    P.only_id = o.b("OnlyId");
    P.buf_end = d_buf + n;
    P.id_mode = id_mode_of(c);
    P.reverse = o.b("Reverse");
    P.remove_gaps = o.b("RemoveGaps");
    set_bits(P.gap_set, o.s("GapLetters"));
    P.gap_lt64 = 1;
    for (char ch : o.s("GapLetters")) if ((uint8_t)ch >= 64) P.gap_lt64 = 0;
    P.line_width = (fastq || o.b("Seq") || o.b("Qual")) ? 0 : (int)o.ci("LineWidth");
    P.min_len = (int)o.i("MinLen"); P.max_len = (int)o.i("MaxLen");
    P.min_qual = o.f("MinQual"); P.max_qual = o.f("MaxQual");
    P.qual_base = (int)o.i("QualAsciiBase");
    bool validate = o.b("ValidateSeq");
    if (!validate && c->alphabet_given()) validate = true;  // seq.go:66-72 (a type that was GIVEN: not a partition's pinned guess)
    const char* letters = alphabet_letters(ab);
    P.validate = validate && letters != nullptr;
    P.validate_len = (int)o.i("ValidateSeqLength");
    if (letters) set_bits(P.valid_set, letters);
    // one byte map for complement -> dna2rna -> rna2dna -> case (seq.go:191-239)
    uint8_t lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)i;
    bool use_lut = false;
    auto apply = [&](const char* from, const char* to) {
        uint8_t m[256];
        for (int i = 0; i < 256; ++i) m[i] = (uint8_t)i;
        for (size_t k = 0; from[k]; ++k) m[(uint8_t)from[k]] = (uint8_t)to[k];
        for (int i = 0; i < 256; ++i) lut[i] = m[lut[i]];
        use_lut = true;
    };
    if (o.b("Complement")) {
        if (ab == AB_DNA || ab == AB_DNAredundant) apply("acgtryswkmbdhvACGTRYSWKMBDHV", "tgcayrswmkvhdbTGCAYRSWMKVHDB");
        else if (ab == AB_RNA || ab == AB_RNAredundant) apply("acguryswkmbdhvACGURYSWKMBDHV", "ugcayrswmkvhdbUGCAYRSWMKVHDB");
    }
    if (o.b("Dna2rna") && !(ab == AB_RNA || ab == AB_RNAredundant)) apply("tT", "uU");
    if (o.b("Rna2dna") && !(ab == AB_DNA || ab == AB_DNAredundant)) apply("uU", "tT");
    if (o.b("LowerCase")) apply("ABCDEFGHIJKLMNOPQRSTUVWXYZ", "abcdefghijklmnopqrstuvwxyz");
    else if (o.b("UpperCase")) apply("abcdefghijklmnopqrstuvwxyz", "ABCDEFGHIJKLMNOPQRSTUVWXYZ");
    P.use_lut = use_lut;
    if (!c->d_lut) HIP_TRYX(c, hipMalloc((void**)&c->d_lut, 256));
    if (!c->d_qual_err) HIP_TRYX(c, hipMalloc((void**)&c->d_qual_err, 256 * sizeof(double)));
    HIP_TRYX(c, hipMemcpyAsync(c->d_lut, lut, 256, hipMemcpyHostToDevice, st));
    double qe[256];
    for (int q = 0; q < 256; ++q) qe[q] = std::pow(10.0, (double)(q - P.qual_base) / -10.0);
    HIP_TRYX(c, hipMemcpyAsync(c->d_qual_err, qe, sizeof qe, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));  // lut / qe live on the host stack
    P.lut = c->d_lut;
    P.qual_err = c->d_qual_err;

    if (c->table.n == 0) {
        uint64_t status = 0;
        HIP_TRYX(c, hipMemcpy(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost));
        return kernel_error_to_status(c, status);
    }
    {  // wrapped FASTA: random access through the text view (with gap removal: for the records without a gap letter)
        TextTableH tt{nullptr, nullptr, nullptr};
        rc = prepare_text(c, d_buf, format, st, &tt);
        if (rc != BSK_OK) return rc;
        P.text_w = tt.text_w; P.lin_off = tt.lin_off; P.lin = tt.lin;
    }
    rc = ensure_record_scratch(c);
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, launch_seq_size(d_buf, c->table, P, c->d_out_len, c->d_status, st));
    uint64_t total = 0, kept = 0;
    rc = finish_sizes(c, st, &total, &kept);
    if (rc != BSK_OK) return rc;
    {
        const int rs = try_records_as_slices(c, d_buf, n, P, total, kept, st, out);
        if (rs < 0) return -rs;
        if (rs == 1) return BSK_OK;
    }
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    apply_long(c, &P);
    { const int rce = emit_records(c, d_buf, n, P, total, kept, st); if (rce != BSK_OK) return rce; }
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    return BSK_OK;
}


}  // namespace bsk
