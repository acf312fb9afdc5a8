// Host side of `subseq` (SubseqTransform, /root/reference/bigseqkit-lib/subseq.go): BED / GTF features, region mode, the
// streaming pass for `-r` on FASTQ.
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

// ---------------------------------------------------------------------------
// subseq --gtf / --bed: feature files (host side of SubseqTransform.Before, subseq.go:100-165)
//   ReadBedFilteredFeatures  bigseqkit-lib/subseq.go:242-310 (in tree)
//   gtf.ReadFilteredFeatures shenwei356/bio featio/gtf (not in tree; PARITY.md GTF)
// Only the FIRST feature of a sequence name is ever used (subseq.go:426, 523 return inside the loop, Q7), so that
// is all the context keeps: name -> (flank-adjusted start, end, strand, header suffix).
// ---------------------------------------------------------------------------
static std::vector<std::string> split_tabs(const std::string& line) {
    std::vector<std::string> items;
    for (size_t i = 0;;) {
        size_t j = line.find('\t', i);
        if (j == std::string::npos) { items.emplace_back(line, i); break; }
        items.emplace_back(line, i, j - i);
        i = j + 1;
    }
    return items;
}

static bool atoi_strict(const std::string& s, long long* v) {
    if (s.empty() || isspace((unsigned char)s[0])) return false;
    char* e = nullptr;
    *v = strtoll(s.c_str(), &e, 10);
    return *e == 0;
}

static std::string lower_str(std::string s) {
    for (auto& ch : s) if (ch >= 'A' && ch <= 'Z') ch += 32;
    return s;
}

static void load_features(bsk_ctx* c) {
    const Options& o = c->opts;
    const bool gtf = !o.s("Gtf").empty();
    const std::vector<std::string>& chrs = o.sl("Chr");
    std::vector<std::string> feats;
    for (auto& f : o.sl("Feature")) feats.push_back(lower_str(f));
    if (!gtf && !feats.empty()) throw OptError("when given flag -b (--bed), flag -f (--feature) is not allowed");
    const int64_t up = o.i("UpStream"), down = o.i("DownStream");
    const bool only = o.b("OnlyFlank");
    std::string flank;
    if (up > 0) {
        if (only) flank = "_usf:" + std::to_string(up);
        else if (down > 0) flank = "_us:" + std::to_string(up) + "_ds:" + std::to_string(down);
        else flank = "_us:" + std::to_string(up);
    } else if (down > 0) {
        flank = only ? "_dsf:" + std::to_string(down) : "_ds:" + std::to_string(down);
    }
    c->features.clear();
    c->features_uploaded = false;
    std::unordered_set<std::string> seen;
    c->info(gtf ? "read GTF file ..." : "read BED file ...", /*unless_quiet=*/true);  // subseq.go:98-100, 131-133
    size_t nloaded = 0;  // len(features) of the reference: every accepted line, before the per-name map keeps the first
    for (const std::string& line : read_pattern_lines(gtf ? o.s("Gtf") : o.s("Bed"))) {
        if (line.empty() || line[0] == '#') continue;
        if (!gtf && ((line.size() > 7 && line.compare(0, 7, "browser") == 0) || (line.size() > 5 && line.compare(0, 5, "track") == 0)))
            continue;
        const auto items = split_tabs(line);
        if (gtf ? items.size() != 9 : items.size() < 3) continue;
        if (!chrs.empty() && std::find(chrs.begin(), chrs.end(), items[0]) == chrs.end()) continue;
        if (gtf && !feats.empty() && std::find(feats.begin(), feats.end(), lower_str(items[2])) == feats.end()) continue;
        long long st, en;
        const std::string &sst = items[gtf ? 3 : 1], &sen = items[gtf ? 4 : 2];
        if (!atoi_strict(sst, &st)) throw OptError(items[0] + ": bad start: " + sst);
        if (!atoi_strict(sen, &en)) throw OptError(items[0] + ": bad end: " + sen);
        std::string strand = ".", label;
        if (gtf) {
            if (st > en) throw OptError(items[0] + ": start (" + std::to_string(st) + ") must be < end (" + std::to_string(en) + ")");
            if (items[6] != "+" && items[6] != "-" && items[6] != ".") throw OptError("bad strand: " + items[6]);
            strand = items[6];
            const std::string& at = items[8];  // tag "value"; tag "value";
            for (size_t i = 0; i < at.size();) {
                size_t j = at.find(';', i);
                if (j == std::string::npos) j = at.size();
                std::string item(at, i, j - i);
                i = j + 1;
                const size_t a0 = item.find_first_not_of(' ');
                if (a0 == std::string::npos) continue;
                item.erase(0, a0);
                const size_t sp = item.find(' ');
                if (sp == std::string::npos) continue;
                std::string v(item, sp + 1);
                while (!v.empty() && v.back() == ' ') v.pop_back();
                if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);
                if (item.compare(0, sp, o.s("GtfTag")) == 0 && sp == o.s("GtfTag").size()) { label = v; break; }
            }
        } else {
            if (st >= en) throw OptError(items[0] + ": start (" + std::to_string(st) + ") must be <= end (" + std::to_string(en) + ")");
            st += 1;  // BED start is 0-based (subseq.go:294)
            if (items.size() >= 4) label = items[3];
            if (items.size() >= 6) {
                if (items[5] != "+" && items[5] != "-" && items[5] != ".") throw OptError("bad strand: " + items[5]);
                strand = items[5];
            }
        }
        ++nloaded;
        const std::string key = lower_str(items[0]);
        if (!seen.insert(key).second) continue;  // a later feature of the same name is never reached
        bsk_ctx::Feature f;
        f.name_lower = key;
        f.minus = strand == "-";
        if (f.minus) {  // subseq.go:340-352
            if (only) { if (up > 0) { f.s = en + 1; f.e = en + up; } else { f.s = st - down; f.e = st - 1; } }
            else { f.s = st - down; f.e = en + up; }
        } else {        // subseq.go:359-371
            if (only) { if (up > 0) { f.s = st - up; f.e = st - 1; } else { f.s = en + 1; f.e = en + down; } }
            else { f.s = st - up; f.e = en + down; }
        }
        f.suffix = "_" + std::to_string(st) + "-" + std::to_string(en) + ":" + strand + flank + " " + label;
        c->features.push_back(f);
    }
    c->info(std::to_string(nloaded) + (gtf ? " GTF" : " BED") + " features loaded", true);  // subseq.go:127-129, 157-159
}

// ---------------------------------------------------------------------------
// subseq by region  (SubseqTransform, bigseqkit-lib/subseq.go:36-165, 314-317)
// ---------------------------------------------------------------------------
void validate_subseq_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    if (o.b("OnlyFlank")) {
        if (o.i("UpStream") > 0 && o.i("DownStream") > 0)
            throw OptError("when flag -f (--only-flank) given, only one of flags -u (--up-stream) and -d (--down-stream) is allowed");
        else if (o.i("UpStream") == 0 && o.i("DownStream") == 0)
            throw OptError("when flag -f (--only-flank) given, one of flags -u (--up-stream) and -d (--down-stream) should be given");
    }
    if (!o.s("Region").empty()) {
        if (o.i("UpStream") > 0 || o.i("DownStream") > 0 || o.b("OnlyFlank"))
            throw OptError("when flag -r (--region) given, any of flags -u (--up-stream), -d (--down-stream) and -f (--only-flank) is not allowed");
        c->region_on = true;
        parse_region_opt(o.s("Region"), "subseq", &c->region_start, &c->region_end);
    } else if (!o.s("Gtf").empty() || !o.s("Bed").empty()) {
        load_features(c);
    } else {
        throw OptError("one of the options needed: -r/--region, --bed, --gtf");
    }
}

static int upload_features(bsk_ctx* c, hipStream_t st) {
    const size_t nf = c->features.size();
    uint64_t slots = 16;
    while (slots < 2 * nf) slots <<= 1;
    std::vector<uint64_t> keys(slots, 0);
    std::vector<uint32_t> idx(slots, 0), name_off{0}, suf_off{0};
    std::vector<int64_t> fs(nf), fe(nf);
    std::vector<uint8_t> minus(nf), names, sufs;
    for (size_t k = 0; k < nf; ++k) {
        const auto& f = c->features[k];
        uint64_t h = 1469598103934665603ull;
        for (unsigned char ch : f.name_lower) h = (h ^ ch) * 1099511628211ull;
        if (!h) h = 1;
        uint64_t s = h & (slots - 1);
        while (keys[s]) s = (s + 1) & (slots - 1);
        keys[s] = h;
        idx[s] = (uint32_t)k;
        names.insert(names.end(), f.name_lower.begin(), f.name_lower.end());
        name_off.push_back((uint32_t)names.size());
        sufs.insert(sufs.end(), f.suffix.begin(), f.suffix.end());
        suf_off.push_back((uint32_t)sufs.size());
        fs[k] = f.s; fe[k] = f.e; minus[k] = f.minus;
    }
    // one allocation, every array 16-byte aligned
    const void* src[8] = {keys.data(), idx.data(), name_off.data(), suf_off.data(), fs.data(), fe.data(), minus.data(), nullptr};
    const uint64_t bytes[8] = {slots * 8, slots * 4, name_off.size() * 4, suf_off.size() * 4, nf * 8, nf * 8, nf, 0};
    uint64_t off = 0;
    for (int a = 0; a < 7; ++a) { c->feat_off[a] = off; off += (bytes[a] + 15) & ~15ull; }
    const uint64_t names_at = off;
    off += (names.size() + 15) & ~15ull;
    const uint64_t sufs_at = off;
    off += (sufs.size() + 15) & ~15ull;
    c->feat_off[7] = names_at;
    int rc = grow(c, &c->d_feat, &c->feat_cap, off + 16);
    if (rc != BSK_OK) return rc;
    for (int a = 0; a < 7; ++a)
        if (bytes[a]) HIP_TRYX(c, hipMemcpyAsync(c->d_feat + c->feat_off[a], src[a], bytes[a], hipMemcpyHostToDevice, st));
    if (!names.empty()) HIP_TRYX(c, hipMemcpyAsync(c->d_feat + names_at, names.data(), names.size(), hipMemcpyHostToDevice, st));
    if (!sufs.empty()) HIP_TRYX(c, hipMemcpyAsync(c->d_feat + sufs_at, sufs.data(), sufs.size(), hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    c->feat_slots = slots;
    c->feat_off[8] = sufs_at;
    c->features_uploaded = true;
    return BSK_OK;
}

// the feature set of the context (subseq --gtf / --bed, faidx region queries) on the device, bound to P
int bind_features(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, SeqParams* P) {
    int rc = BSK_OK;
    if (!c->features_uploaded) {
        rc = upload_features(c, st);
        if (rc != BSK_OK) return rc;
    }
    Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);
    if (rc != BSK_OK) return rc;
    uint8_t comp[256];
    complement_table(ab, comp);
    if (!c->d_lut) HIP_TRYX(c, hipMalloc((void**)&c->d_lut, 256));
    HIP_TRYX(c, hipMemcpyAsync(c->d_lut, comp, 256, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));  // comp lives on the host stack
    const uint8_t* base = c->d_feat;
    P->feat_on = 1;
    P->fset_keys = (const uint64_t*)(base + c->feat_off[0]);
    P->fset_idx = (const uint32_t*)(base + c->feat_off[1]);
    P->fset_mask = c->feat_slots - 1;
    P->fname_off = (const uint32_t*)(base + c->feat_off[2]);
    P->fsuffix_off = (const uint32_t*)(base + c->feat_off[3]);
    P->f_s = (const int64_t*)(base + c->feat_off[4]);
    P->f_e = (const int64_t*)(base + c->feat_off[5]);
    P->f_minus = base + c->feat_off[6];
    P->fname = base + c->feat_off[7];
    P->fsuffix = base + c->feat_off[8];
    P->comp = c->d_lut;
    return BSK_OK;
}

// `subseq -r a:b` on FASTQ: the records leave from the streaming pass itself (stream_subseq.hip) -- per-range slices sized
// from the shard head, one scan over the ranges, one gather.  BSK_ERR_FILTER_FALLBACK: not this path's input (long lines,
// a slice too small); the caller takes the record-table path.
static int subseq_stream_run(bsk_ctx* c, const uint8_t* d_buf, size_t n, hipStream_t st, bsk_out* out) {
    const int blocks = std::max(1, c->num_cus * subseq_stream_max_blocks_per_cu(c->use_dpp));
    uint32_t nranges = 0;
    uint64_t chunk = 0;
    int rc = prep_ranges(c, d_buf, n, /*fastq=*/true, blocks, st, &nranges, &chunk);
    if (rc != BSK_OK) return rc;
    rc = sample_head(c, d_buf, n, st);  // (the call's one head sample, pinned)
    if (rc != BSK_OK) return rc;
    const size_t hb = c->head_len;
    const uint8_t* head = c->h_head;
    if (!c->norm_active && fastq_head_multiline(head, hb)) return BSK_ERR_MULTILINE_FASTQ;
    // output bytes per input byte over the complete records of the sample
    uint64_t in_b = 0, out_b = 0, line = 0, rec_out = 0, max_line = 0;
    for_lines(head, hb, [&](size_t s0, size_t e0, bool terminated) {
        const uint64_t ll = e0 - s0;
        max_line = std::max(max_line, ll);
        if (!terminated) return;
        const uint32_t role = (uint32_t)(line & 3);
        if (role == 0) rec_out = ll + 1;
        else if (role == 2) rec_out += 2;
        else {
            uint32_t b, e;
            sub_location((uint32_t)ll, c->region_start, c->region_end, &b, &e);
            rec_out += (uint64_t)(e - b) + 1;
        }
        if (role == 3) { out_b += rec_out; in_b = e0 + 1; }
        ++line;
    });
    // a lane copies its piece alone: lines of kilobytes (long reads) stay with the record-table kernels
    if (max_line > 2048 || in_b == 0) return BSK_ERR_FILTER_FALLBACK;
    double ratio = (double)(out_b + 64) / (double)in_b;
    if (const char* sc = c->tune.get("subseq_scale")) ratio *= atof(sc);  // tests: force the overflow -> fallback route
    uint64_t slice_cap = (uint64_t)((double)chunk * ratio * 1.25) + (c->tune.get("subseq_scale") ? 16 : 4096);
    slice_cap = (slice_cap + 15) & ~(uint64_t)15;
    if (slice_cap >= (1ull << 32) || slice_cap * nranges > 2 * (uint64_t)n + (64ull << 20)) return BSK_ERR_FILTER_FALLBACK;
    rc = grow(c, &c->d_slices, &c->slices_cap, slice_cap * nranges, 256);
    if (rc != BSK_OK) return rc;
    rc = grow(c, &c->d_names_aux, &c->names_aux_cap, 2 * ((uint64_t)nranges + 2), 16);
    if (rc != BSK_OK) return rc;
    SubseqDev D;
    D.slices = c->d_slices;
    D.slice_cap = slice_cap;
    D.range_bytes = c->d_names_aux;
    D.range_count = c->d_range_count;
    D.status = c->d_status;
    D.region_start = c->region_start;
    D.region_end = c->region_end;
    uint64_t* d_count_base = c->d_names_aux + nranges + 2;
    {
        Timed t(c, "k_subseq_stream", st);
        HIP_TRYX(c, launch_subseq_stream(c->use_dpp, blocks, d_buf, n, c->d_anchors, nranges,
                                         reinterpret_cast<uint32_t*>(c->d_anchors + (size_t)nranges + 1), D, st));
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
    if (total && slices_wanted(c)) {  // round 6: the per-range slices ARE the result, in order (include/bsk.h bsk_out.d_seg_*)
        c->table.n = 0;
        return out_as_slices(c, out, D.slices, D.slice_cap, c->d_range_base, nranges, total, records, st);
    }
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    if (total) {
        Timed t(c, "k_subseq_compact", st);
        NamesDev G;
        memset(&G, 0, sizeof G);
        G.slices = D.slices;
        G.slice_cap = D.slice_cap;
        G.range_bytes = D.range_bytes;
        HIP_TRYX(c, launch_names_compact(G, c->d_range_base, nranges, c->d_out, st));
    }
    c->table.n = 0;  // no record table was built for this shard
    out->d_data = c->d_out;
    out->len = total;
    out->records = records;
    return BSK_OK;
}

int subseq_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const bool fastq = format == BSK_FORMAT_FASTQ;
    if (fastq && n > 0 && c->region_on && !c->tune.is("subseq", "table")) {
        const int rcs = subseq_stream_run(c, d_buf, n, st, out);
        if (rcs != BSK_ERR_FILTER_FALLBACK) return rcs;
    }
    int rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    SeqParams P = format_params(c, fastq);
    P.buf_end = d_buf + n;
    if (c->region_on) {
        P.region_on = 1;
        P.region_start = c->region_start;
        P.region_end = c->region_end;
    } else {
        if (c->features.empty()) return empty_result(c, out);  // no record can have a feature
        rc = bind_features(c, d_buf, n, format, st, &P);
        if (rc != BSK_OK) return rc;
    }
    {   // wrapped FASTA: random access through the text view instead of the sequential per-record walk
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
