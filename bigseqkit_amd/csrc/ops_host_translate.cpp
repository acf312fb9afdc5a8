// Host side of `translate` (Translate, /root/reference/bigseqkit-lib/translate.go): genetic codes, codon tables, the
// light FASTA record table and the kernel sequence.
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
// translate  (Translate.Before, bigseqkit-lib/translate.go:33-64)
// ---------------------------------------------------------------------------
#include "genetic_codes.inc"

// names of the tables as the reference lists them (bigseqkit-cli/translate.go:55-78); `translate -l 0` prints "ID\tName"
static const struct { int id; const char* name; } kCodeNames[] = {
    {1, "The Standard Code"},
    {2, "The Vertebrate Mitochondrial Code"},
    {3, "The Yeast Mitochondrial Code"},
    {4, "The Mold, Protozoan, and Coelenterate Mitochondrial Code and the Mycoplasma/Spiroplasma Code"},
    {5, "The Invertebrate Mitochondrial Code"},
    {6, "The Ciliate, Dasycladacean and Hexamita Nuclear Code"},
    {9, "The Echinoderm and Flatworm Mitochondrial Code"},
    {10, "The Euplotid Nuclear Code"},
    {11, "The Bacterial, Archaeal and Plant Plastid Code"},
    {12, "The Alternative Yeast Nuclear Code"},
    {13, "The Ascidian Mitochondrial Code"},
    {14, "The Alternative Flatworm Mitochondrial Code"},
    {16, "Chlorophycean Mitochondrial Code"},
    {21, "Trematode Mitochondrial Code"},
    {22, "Scenedesmus obliquus Mitochondrial Code"},
    {23, "Thraustochytrium Mitochondrial Code"},
    {24, "Pterobranchia Mitochondrial Code"},
    {25, "Candidate Division SR1 and Gracilibacteria Code"},
    {26, "Pachysolen tannophilus Nuclear Code"},
    {27, "Karyorelict Nuclear"},
    {28, "Condylostoma Nuclear"},
    {29, "Mesodinium Nuclear"},
    {30, "Peritrich Nuclear"},
    {31, "Blastocrithidia Nuclear"},
};

static const GeneticCode* find_code(int id) {
    for (auto& g : kGeneticCodes)
        if (g.id == id) return &g;
    return nullptr;
}

void validate_translate_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    if (!find_code((int)o.i("TranslTable"))) throw OptError("invalid translate table: " + std::to_string(o.i("TranslTable")));
    c->frames.clear();
    for (auto& f : o.sl("Frame")) {
        char* endp = nullptr;
        const long v = strtol(f.c_str(), &endp, 10);
        if (f.empty() || *endp)
            throw OptError("invalid frame(s): " + f + ". available: 1, 2, 3, -1, -2, -3, and 6 for all. multiple frames should be separated by comma");
        if (!(v == 1 || v == 2 || v == 3 || v == -1 || v == -2 || v == -3 || v == 6))
            throw OptError("invalid frame: " + std::to_string(v) + ". available: 1, 2, 3, -1, -2, -3, and 6 for all");
        if (v == 6) { c->frames = {1, 2, 3, -1, -2, -3}; break; }
        c->frames.push_back((int)v);
    }
    if (c->frames.size() > 6) throw OptError("libbsk: at most 6 frames per call");
    // translate.go:75-101: -l 0 / -L 0 list the tables; -l N / -L N print bio's CodonTable.String() /
    // StringWithAmbiguousCodons(), whose layout lives in shenwei356/bio (not in tree) -- refused, see PARITY.md
    if (o.i("ListTranslTable") > 0 || o.i("ListTranslTableWithAmbCodons") > 0)
        throw OptError("libbsk: translate -l N / -L N (the codon listing of one table) is not provided; -l 0 lists the tables");
}

// 4096-entry tables over 4-bit IUPAC codes (A=1 C=2 G=4 T=8): amino acid common to all
// expansions of the codon ('X' when they disagree), and the exact start codons
static void build_codon_tables(const GeneticCode& g, uint8_t* aa, uint8_t* start) {
    static const int tcag[4] = {8, 2, 1, 4};  // code of T, C, A, G
    auto idx64 = [&](int b1, int b2, int b3) {
        int i[3] = {b1, b2, b3}, r = 0;
        for (int k = 0; k < 3; ++k) {
            int j = 0;
            while (tcag[j] != i[k]) ++j;
            r = r * 4 + j;
        }
        return r;
    };
    memset(aa, 0, 4096);
    memset(start, 0, 4096);
    for (int c1 = 1; c1 < 16; ++c1)
        for (int c2 = 1; c2 < 16; ++c2)
            for (int c3 = 1; c3 < 16; ++c3) {
                char r = 0;
                for (int b1 = 1; b1 <= 8; b1 <<= 1) {
                    if (!(c1 & b1)) continue;
                    for (int b2 = 1; b2 <= 8; b2 <<= 1) {
                        if (!(c2 & b2)) continue;
                        for (int b3 = 1; b3 <= 8; b3 <<= 1) {
                            if (!(c3 & b3)) continue;
                            const char a = g.aa[idx64(b1, b2, b3)];
                            if (r == 0) r = a;
                            else if (r != a) r = 'X';
                        }
                    }
                }
                aa[(c1 << 8) | (c2 << 4) | c3] = (uint8_t)r;
            }
    for (int b1 = 1; b1 <= 8; b1 <<= 1)
        for (int b2 = 1; b2 <= 8; b2 <<= 1)
            for (int b3 = 1; b3 <= 8; b3 <<= 1)
                if (g.starts[idx64(b1, b2, b3)] == 'M') start[(b1 << 8) | (b2 << 4) | b3] = 1;
}

// translate.go:78-89: with -l 0 or -L 0 every Call returns the list of tables ("ID\tName", ascending ids) and reads no record
static int translate_list_tables(bsk_ctx* c, bsk_out* out) {
    std::string txt;
    uint64_t rows = 0;
    for (auto& e : kCodeNames) { txt += std::to_string(e.id) + "\t" + e.name + "\n"; ++rows; }
    int rc = ensure_out(c, txt.size());
    if (rc != BSK_OK) return rc;
    HIP_TRYX(c, hipMemcpy(c->d_out, txt.data(), txt.size(), hipMemcpyHostToDevice));
    out->d_data = c->d_out;
    out->len = txt.size();
    out->records = rows;
    return BSK_OK;
}

// The layout of a FASTA shard whose records all look alike, proposed from the head sample: header length, line width,
// bases and stride of record 0, confirmed on every complete record of the sample (the kernel verifies ALL records byte by
// byte; this probe only decides whether the attempt is worth a launch).  false: not such a shard.
static bool translate_uniform_probe(bsk_ctx* c, size_t n, const TranslateParams& P, UniformLayout* U) {
    const uint8_t* h = c->h_head;
    const size_t hb = c->head_len;
    if (hb < 64 || h[0] != '>') return false;
    // record 0: header line, sequence lines up to the next '>' at a line start
    std::vector<size_t> nl;  // positions of its line breaks
    size_t S = 0;
    {
        size_t p = 0;
        while (p < hb) {
            const void* q = memchr(h + p, '\n', hb - p);
            if (!q) return false;  // (record 0 does not end inside the sample)
            const size_t e = (size_t)((const uint8_t*)q - h);
            nl.push_back(e);
            p = e + 1;
            if (p < hb && h[p] == '>') { S = p; break; }
            if (p >= hb) return false;
        }
    }
    if (S == 0 || nl.size() < 2) return false;
    const size_t H = nl[0];                       // header line length, marker included
    const size_t nlines = nl.size() - 1;          // sequence lines
    const size_t W0 = nl[1] - nl[0] - 1;          // first line
    size_t L = 0;
    for (size_t k = 1; k < nl.size(); ++k) {
        const size_t w = nl[k] - nl[k - 1] - 1;
        if (k + 1 < nl.size() ? w != W0 : (w == 0 || w > W0)) return false;
        L += w;
    }
    if (L == 0 || H < 1 || H > 65535) return false;
    const uint32_t W = nlines > 1 ? (uint32_t)W0 : 0u;
    if (W && W < 50u) return false;               // (k_translate_wide's windows hold at most one line break)
    const uint32_t lw = P.line_width > 0 ? (uint32_t)P.line_width : 0u;
    if (lw && lw < 16u) return false;
    if (L >= (1u << 20)) return false;            // chromosome-sized records have their own launches
    // the file is a whole number of such records (the last one may lack its '\n')
    if (!(n % S == 0 || (n + 1) % S == 0)) return false;
    const uint64_t R = (n + 1) / S;
    if (R < 2) return false;
    // every complete record of the sample: '>' at k * S and the same line breaks
    for (size_t k = 1; (k + 1) * S <= hb; ++k) {
        const uint8_t* r = h + k * S;
        if (r[0] != '>') return false;
        for (size_t e : nl) if (r[e] != '\n') return false;
        size_t cnt = 0;
        for (const uint8_t* q = r; (q = (const uint8_t*)memchr(q, '\n', (size_t)(r + S - q))) != nullptr; ++q) ++cnt;
        if (cnt != nl.size()) return false;
    }
    memset(U, 0, sizeof *U);
    U->on = 1; U->H = (uint32_t)H; U->L = (uint32_t)L; U->W = W; U->S = S; U->n = R;
    uint64_t off = 0;
    for (int k = 0; k < P.nframes; ++k) {
        const uint32_t f = (uint32_t)(P.frames[k] < 0 ? -P.frames[k] : P.frames[k]);
        const uint32_t naa = L >= f - 1u ? (uint32_t)((L - (f - 1u)) / 3u) : 0u;  // num_aa(L, frame)
        uint32_t len = 1u + ((uint32_t)H - 1u) + 1u;                              // '>' name '\n'
        len += naa + ((lw && naa) ? (naa - 1u) / lw : 0u) + 1u;                    // residues, their line breaks, the final '\n'
        U->len[k] = len;
        U->off[k] = (uint32_t)off;
        off += len;
    }
    U->out_S = off;
    return off < (1ull << 32);
}

int translate_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    if (o.i("ListTranslTable") == 0 || o.i("ListTranslTableWithAmbCodons") == 0) return translate_list_tables(c, out);
    int rc = BSK_OK;
    TranslateParams P;
    memset(&P, 0, sizeof P);
    P.fastq = format == BSK_FORMAT_FASTQ;
    P.nframes = (int)c->frames.size();
    for (int k = 0; k < P.nframes; ++k) P.frames[k] = c->frames[k];
    P.trim = o.b("Trim"); P.clean = o.b("Clean"); P.allow_unknown = o.b("AllowUnknownCodon");
    P.init_m = o.b("InitCodonAsM"); P.append_frame = o.b("AppendFrame");
    P.line_width = (int)o.ci("LineWidth");
    P.id_mode = id_mode_of(c);
    if (!c->codon_ready) {  // the tables follow from the options: built and uploaded once per context (0.3 ms of host time per call before)
        if (!c->d_codon) HIP_TRYX(c, hipMalloc((void**)&c->d_codon, 6 * 4096 + 256 + 16384));
        std::vector<uint8_t> tab(6 * 4096 + 256 + 16384);
        uint8_t *fw = tab.data(), *stt = fw + 4096, *rcw = fw + 8192, *rcs = fw + 12288, *iu = fw + 16384;
        build_codon_tables(*find_code((int)o.i("TranslTable")), fw, stt);
        auto comp = [](int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); };
        for (int c0 = 0; c0 < 16; ++c0)
            for (int c1 = 0; c1 < 16; ++c1)
                for (int c2 = 0; c2 < 16; ++c2) {
                    const int i = (c0 << 8) | (c1 << 4) | c2, r = (comp(c2) << 8) | (comp(c1) << 4) | comp(c0);
                    rcw[i] = fw[r];
                    rcs[i] = stt[r];
                }
        memset(iu, 0, 256);
        const char* letters = "acgturyswkmbdhvn";
        const int codes[] = {1, 2, 4, 8, 8, 5, 10, 6, 9, 12, 3, 14, 13, 11, 7, 15};
        for (int k = 0; letters[k]; ++k) { iu[(uint8_t)letters[k]] = (uint8_t)codes[k]; iu[(uint8_t)(letters[k] - 32)] = (uint8_t)codes[k]; }
        for (int i = 0; i < 8192; ++i) {  // tables as the frames kernel wants them: -x and --clean folded in
            uint8_t a = i < 4096 ? fw[i] : rcw[i - 4096];
            if (P.allow_unknown && a == 0) a = 'X';
            if (P.clean && a == '*') a = 'X';
            tab[16384 + 256 + i] = a;
        }
        {   // pairs of plain-letter codons for k_translate_wide (TranslateParams::pair)
            const uint8_t* baked = tab.data() + 16384 + 256;
            const int iu4[4] = {1, 2, 8, 4};  // IUPAC code of the 2-bit codes A C T G
            auto full = [&](int j) { return (iu4[j & 3] << 8) | (iu4[(j >> 2) & 3] << 4) | iu4[(j >> 4) & 3]; };
            uint8_t* pair = tab.data() + 6 * 4096 + 256;
            for (int i = 0; i < 4096; ++i) {
                const int lo = i & 63, hi = i >> 6;
                pair[2 * i] = baked[full(lo)];
                pair[2 * i + 1] = baked[full(hi)];
                pair[8192 + 2 * i] = baked[4096 + full(hi)];
                pair[8192 + 2 * i + 1] = baked[4096 + full(lo)];
            }
        }
        HIP_TRYX(c, hipMemcpyAsync(c->d_codon, tab.data(), tab.size(), hipMemcpyHostToDevice, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        c->codon_ready = true;
    }
    P.pair = c->d_codon + 6 * 4096 + 256;
    P.baked = c->d_codon + 16384 + 256;
    P.codon = c->d_codon;
    P.start = c->d_codon + 4096;
    P.codon_rc = c->d_codon + 8192;
    P.start_rc = c->d_codon + 12288;
    P.iupac = c->d_codon + 16384;

    // ---- FASTA whose records all look alike: no table at all (UniformLayout, ops_translate.hpp).  The host proposes the
    // layout from the head sample, k_translate_wide<G, true> verifies every record against it; one record that differs and
    // the call starts over below, and the context does not try again.  translate_index = full / light skip the attempt.
    if (format == BSK_FORMAT_FASTA && n > 0 && c->translate_uniform_ok && !c->tune.get("translate_index") && !c->tune.get("translate") &&
        !P.init_m && !P.trim && !P.append_frame && !c->id_custom) {
        rc = sample_head(c, d_buf, n, st);
        if (rc != BSK_OK) return rc;
        UniformLayout U;
        if (translate_uniform_probe(c, n, P, &U)) {
            Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);
            if (rc != BSK_OK) return rc;
            if (!(ab == AB_DNA || ab == AB_DNAredundant || ab == AB_RNA || ab == AB_RNAredundant)) {  // translate.go:116-122
                c->set_error("command 'seqkit translate' only apply to DNA/RNA sequences");
                return BSK_ERR_FORMAT;
            }
            const uint64_t total = U.n * U.out_S;
            rc = ensure_out(c, total);
            if (rc != BSK_OK) return rc;
            P.uni = U;
            const int forced = (int)c->tune.num("tr_lanes");
            const uint64_t avg = U.S;
            const int wide_lanes = forced == 4 || forced == 16 || forced == 64 ? forced : (avg >= 3000 ? 64 : (avg < 500 ? 4 : 16));
            uint64_t* d_redo_count = c->d_fin + bsk_ctx::FIN_AUX0;
            HIP_TRYX(c, hipMemsetAsync(d_redo_count, 0, sizeof(uint64_t), st));
            {
                Timed t(c, "k_translate_uniform", st);
                HIP_TRYX(c, launch_translate_uniform(wide_lanes, d_buf, n, P, c->d_out, d_redo_count, c->d_status, st));
            }
            rc = ctl_readback(c, st);  // records that did not verify + the status word: the call's one read-back
            if (rc != BSK_OK) return rc;
            if (c->fin(bsk_ctx::FIN_AUX0) == 0) {
                rc = kernel_error_to_status(c, c->status_word());
                if (rc != BSK_OK) return rc;
                c->table.n = 0;  // no record table was built for this shard
                out->d_data = c->d_out;
                out->len = total;
                out->records = U.n * (uint64_t)P.nframes;
                return BSK_OK;
            }
            c->translate_uniform_ok = false;  // a record that is not like record 0: the table paths, from now on
            P.uni.on = 0;
        }
    }

    // ---- FASTA of long records that do NOT all look alike (round 5): one pass -- record starts, sizes, output offsets and the
    // translation itself (k_translate_stream), every window verified; anything that does not fit and the call goes on below
    // with the tables, and the context stays with them.  translate_stream = off skips it, = force takes it for short records too.
    if (c->translate_stream_skip) --c->translate_stream_skip;  // (backing off after a misfit: this call takes the tables)
    else if (format == BSK_FORMAT_FASTA && n > 0 && c->translate_stream_ok && !c->tune.get("translate_index") && !c->tune.get("translate") &&
        !c->tune.is("translate_stream", "off") && !P.init_m && !P.trim && !P.append_frame && !c->id_custom && (P.line_width == 0 || P.line_width >= 16)) {
        rc = sample_head(c, d_buf, n, st);
        if (rc != BSK_OK) return rc;
        uint64_t heads = 1;
        for (size_t i = 0; i + 1 < c->head_len; ++i) heads += (c->h_head[i] == '\n' && c->h_head[i + 1] == '>');
        const uint64_t avg = c->head_len / heads;
        if (avg >= 1500 || c->tune.is("translate_stream", "force")) {   // (a wave per record: from ~3 k bases; shorter records take 4 / 16 lanes)
            Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);
            if (rc != BSK_OK) return rc;
            if (!(ab == AB_DNA || ab == AB_DNAredundant || ab == AB_RNA || ab == AB_RNAredundant)) {  // translate.go:116-122
                c->set_error("command 'seqkit translate' only apply to DNA/RNA sequences");
                return BSK_ERR_FORMAT;
            }
            const int blocks = std::max(1, c->num_cus * translate_stream_max_blocks_per_cu());
            uint32_t nranges = 0;
            uint64_t chunk = 0;
            // ranges of ~200 records (the lists of a block hold 512), between 256 KiB and 2 MiB: the per-range phases of the pass
            // cost what they cost per RANGE (64 KiB: 76 ms at C4, 512 KiB: 55, 1 MiB: 51 -- scripts/history/r05_trstream.sh);
            // min_range_bytes pins the size (tests: ranges of 4 KiB)
            uint64_t want_chunk = std::min<uint64_t>(2ull << 20, std::max<uint64_t>(256ull << 10, avg * 200));
            if (c->tune.num("min_range_bytes", 0) > 0) want_chunk = (uint64_t)c->tune.num("min_range_bytes", 0);
            want_chunk = std::max<uint64_t>(16, want_chunk & ~15ull);
            rc = prep_ranges(c, d_buf, n, false, blocks, st, &nranges, &chunk, want_chunk);
            if (rc != BSK_OK) return rc;
            uint32_t* queue = reinterpret_cast<uint32_t*>(c->d_anchors + (size_t)nranges + 1);
            // six frames of L bases are 2 L residues + their line breaks + six headers: the reserved room is checked per record
            const uint64_t out_cap = (uint64_t)n * (uint64_t)std::max(1, P.nframes) * 3 / 8 + (64ull << 20);
            rc = ensure_out(c, out_cap);
            if (rc != BSK_OK) return rc;
            rc = grow(c, &c->d_scan_tmp, &c->scan_tmp_cap, (uint64_t)nranges + 8, 64);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_scan_tmp, 0, ((size_t)nranges + 1) * sizeof(uint64_t), st));
            HIP_TRYX(c, hipMemsetAsync(c->d_fin, 0, 2 * sizeof(uint64_t), st));
            uint64_t* d_redo_count = c->d_fin + bsk_ctx::FIN_AUX0;
            HIP_TRYX(c, hipMemsetAsync(d_redo_count, 0, sizeof(uint64_t), st));
            {
                const char* e = c->tune.get("long_bytes");
                P.long_thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
            }
            {
                Timed t(c, "k_translate_stream", st);
                // (round 6: the record starts of a range are predicted from the starts at its head and verified in 128-byte
                // windows; translate_probe = off searches every range in full, as round 5 did)
                const uint32_t span_hint = c->tune.is("translate_probe", "off") ? 0u : (uint32_t)std::min<uint64_t>(avg, 1u << 20);
                HIP_TRYX(c, launch_translate_stream(blocks, d_buf, n, c->d_anchors, nranges, queue, P, c->d_out, out_cap, c->d_scan_tmp, c->d_fin,
                                                    d_redo_count, c->d_status, st, span_hint));
            }
            P.long_thresh = 0;
            rc = ctl_readback(c, st);
            if (rc != BSK_OK) return rc;
            if (c->fin(bsk_ctx::FIN_AUX0) == 0) {
                rc = kernel_error_to_status(c, c->status_word());
                if (rc != BSK_OK) return rc;
                c->table.n = 0;  // no record table was built for this shard
                out->d_data = c->d_out;
                out->len = c->fin(bsk_ctx::FIN_TOTAL);
                out->records = c->fin(bsk_ctx::FIN_KEPT) * (uint64_t)P.nframes;
                c->translate_stream_backoff = 8;
                return BSK_OK;
            }
            // something did not fit: this call and the next few take the table paths, then the pass is tried again
            ++c->translate_stream_fallbacks;
            c->translate_stream_skip = c->translate_stream_backoff;
            c->translate_stream_backoff = std::min<uint32_t>(1024, c->translate_stream_backoff * 2);
            HIP_TRYX(c, hipMemsetAsync(c->d_status, 0, 8 * sizeof(uint64_t), st));
        }
    }

    // FASTA: first with the record table from the '>' bytes alone (stream_fasta_light.hip) -- k_translate_wide validates the
    // whole text against the layout that table assumes; whatever does not fit (a record flagged by the wide kernel, a
    // chromosome-sized one, ...) sends the call through the full index pass below, and the context remembers it
    // (... which is why the light table is not used where the wide kernel does not run: a shard below its minimum size)
    bool light = format == BSK_FORMAT_FASTA && c->translate_light_ok && !c->tune.is("translate_index", "full") &&
                 !c->tune.get("translate") && !o.b("InitCodonAsM") && (uint64_t)n >= TRANSLATE_WIDE_MIN_BYTES;
    rc = BSK_ERR_FILTER_FALLBACK;
    if (light) {
        rc = build_index_light(c, d_buf, n, st);
        if (rc != BSK_OK && rc != BSK_ERR_FILTER_FALLBACK) return rc;
    }
    if (rc == BSK_ERR_FILTER_FALLBACK) {
        light = false;
        rc = build_index(c, d_buf, n, format, st);
    }
    if (rc != BSK_OK) return rc;
    if (c->table.n == 0) return empty_result(c, out);
    Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);
    if (rc != BSK_OK) return rc;
    if (!(ab == AB_DNA || ab == AB_DNAredundant || ab == AB_RNA || ab == AB_RNAredundant)) {  // translate.go:116-122
        c->set_error("command 'seqkit translate' only apply to DNA/RNA sequences");
        return BSK_ERR_FORMAT;
    }
    TextTableH tt;
    rc = prepare_text(c, d_buf, format, st, &tt);
    if (rc != BSK_OK) return rc;
    // per-element scratch: nframes elements per record
    const uint64_t ne = c->table.n * (uint64_t)P.nframes;
    const uint64_t saved_n = c->table.n;
    c->table.n = ne;  // size the scratch for elements
    rc = ensure_record_scratch(c);
    c->table.n = saved_n;
    if (rc != BSK_OK) return rc;
    {
        Timed t(c, "k_translate_size+scan", st);
        HIP_TRYX(c, launch_translate_size(d_buf, c->table, tt, P, c->d_out_len, c->d_status, st));
        HIP_TRYX(c, launch_scan_u32(c->d_out_len, c->d_out_off, ne, c->d_scan_tmp, st));
    }
    uint64_t total = 0;
    HIP_TRYX(c, hipMemcpyAsync(&total, c->d_out_off + ne, sizeof total, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    // chromosome-sized records leave the per-record kernels (one wave would translate 10^8 bases alone)
    uint64_t long_max = 0;
    {
        const char* e = c->tune.get("long_bytes");
        const uint32_t thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
        const char* mode = c->tune.get("translate");
        if (!(mode && strcmp(mode, "legacy") == 0)) {
            rc = grow(c, &c->d_long_list, &c->long_list_cap, c->table.n, c->table.n / 8 + 16);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
            HIP_TRYX(c, launch_find_long(c->table.l_seq, c->table.n, thresh, c->d_long_list, c->d_counter + 2, st));
            uint64_t lc[2] = {0, 0};
            HIP_TRYX(c, hipMemcpyAsync(lc, c->d_counter + 2, sizeof lc, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            if (light && lc[0]) {  // chromosome-sized records are translated from positions, nothing validates their layout
                c->translate_light_ok = false;
                return translate_run_device(c, d_buf, n, format, st, out);
            }
            if (lc[0] && lc[0] * (uint64_t)P.nframes <= 65535) {  // (grid.y; more long records than that stay per record)
                P.long_list = c->d_long_list;
                P.long_count = lc[0];
                P.long_thresh = thresh;
                long_max = lc[1];
            }
        }
    }
    {
        // wave per record for long sequences, 16 lanes per record for reads; BSK_TRANSLATE=legacy keeps
        // the per-(record, frame) kernel (used by tests to cross-check the two implementations)
        const char* mode = c->tune.get("translate");
        if (mode && strcmp(mode, "legacy") == 0) {
            HIP_TRYX(c, launch_translate_emit(d_buf, c->table, tt, P, c->d_out_len, c->d_out_off, c->d_out, c->d_status, st));
        } else {
            const uint64_t avg = n / std::max<uint64_t>(1, c->table.n);
            const int forced = (int)c->tune.num("tr_lanes");  // measurement knob
            // one flag byte per record for the records k_translate_wide leaves to k_translate_frames4
            rc = grow(c, &c->d_redo, &c->redo_cap, c->table.n, c->table.n / 8 + 64);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_redo, 0, c->table.n, st));
            if (!c->d_counter) HIP_TRYX(c, hipMalloc((void**)&c->d_counter, 4 * sizeof(uint64_t)));
            HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, sizeof(uint64_t), st));
            // wide kernel: a wave per record from ~3 k bases (a step of 64 lanes covers 3072), 16 lanes per record below
            // (reads: a step of 4 lanes covers 192 bases; 16 lanes per 150-base read left 12 of them idle)
            const int wide_lanes = forced == 4 || forced == 16 || forced == 64 ? forced : (avg >= 3000 ? 64 : (avg < 500 ? 4 : 16));
            uint64_t redo_left = 0;
            {
                Timed t(c, "k_translate", st);
                HIP_TRYX(c, launch_translate_frames(avg >= 1024 ? 64 : 16, d_buf, c->table, tt, P, c->d_out_len, c->d_out_off,
                                                    c->d_out, c->d_status, st, n, c->d_redo, wide_lanes, c->d_counter,
                                                    c->tune.is("translate", "v3") ? 1 : (c->tune.is("translate", "frames4") ? 2 : 0),
                                                    light ? &redo_left : nullptr));
            }
            if (light && redo_left) {
                // a record did not fit the layout the light table assumed (or holds letters beyond ACGT): its l_seq cannot
                // be trusted -- the whole call again with the full index pass; this context stays with it
                c->translate_light_ok = false;
                return translate_run_device(c, d_buf, n, format, st, out);
            }
            HIP_TRYX(c, launch_translate_long(d_buf, c->table, tt, P, c->d_out_len, c->d_out_off, c->d_out, c->d_status,
                                              long_max, st));
        }
    }
    uint64_t status = 0;
    HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    if (status & ERR_UNKNOWN_CODON) {
        c->set_error("seq: unknown codon (use flag -x/--allow-unknown-codon to translate it to 'X')");
        return BSK_ERR_FORMAT;
    }
    rc = kernel_error_to_status(c, status);
    if (rc != BSK_OK) return rc;
    out->d_data = c->d_out;
    out->len = total;
    out->records = ne;
    return BSK_OK;
}


}  // namespace bsk
