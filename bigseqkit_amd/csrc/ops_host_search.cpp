// Host side of `grep` and `locate` (Grep / Locate, /root/reference/bigseqkit-lib/grep.go, locate.go): pattern sets, option
// validation, the fused filter of the streaming pass and the per-record search kernels' orchestration.
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
// pattern helpers shared by grep and locate
// ---------------------------------------------------------------------------
using ByteSet = std::array<uint32_t, 8>;
static inline void set_add(ByteSet& s, uint8_t b) { s[b >> 5] |= 1u << (b & 31); }
static inline bool set_has(const ByteSet& s, uint8_t b) { return (s[b >> 5] >> (b & 31)) & 1u; }

// Seq.Degenerate2Regexp [shenwei356/bio v0.7.0, not in tree; PARITY.md DEG]: the letters a degenerate
// base / residue stands for; nullptr = the byte stays a literal of the regular expression
static std::string degenerate_letters(char c, bool protein) {
    const bool low = c >= 'a' && c <= 'z';
    const char u = low ? (char)(c - 32) : c;
    std::string r;
    if (!protein) {
        switch (u) {
            case 'A': case 'C': case 'G': case 'T': case 'U': r = std::string(1, u); break;
            case 'R': r = "AG"; break; case 'Y': r = "CT"; break; case 'M': r = "AC"; break; case 'K': r = "GT"; break;
            case 'S': r = "CG"; break; case 'W': r = "AT"; break; case 'H': r = "ACT"; break; case 'B': r = "CGT"; break;
            case 'V': r = "ACG"; break; case 'D': r = "AGT"; break; case 'N': r = "ACGT"; break;
            default: return "";
        }
    } else {
        if (u < 'A' || u > 'Z') return "";
        switch (u) {
            case 'B': r = "DN"; break; case 'Z': r = "EQ"; break; case 'J': r = "IL"; break;
            case 'X': r = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"; break;
            default: r = std::string(1, u);
        }
    }
    if (low) for (auto& ch : r) ch = (char)(ch + 32);
    return r;
}

static std::vector<ByteSet> class_sets(const std::string& p, bool degenerate, bool protein, bool icase) {
    std::vector<ByteSet> out;
    for (char ch : p) {
        ByteSet s{};
        std::string letters = degenerate ? degenerate_letters(ch, protein) : std::string();
        if (letters.empty()) {
            if (degenerate && !((ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z')))
                throw OptError("libbsk: with -d the HIP path takes patterns made of letters only (regular-expression "
                               "syntax is not supported): " + p);
            letters = std::string(1, ch);
        }
        for (char l : letters) {
            set_add(s, (uint8_t)l);
            if (icase && l >= 'A' && l <= 'Z') set_add(s, (uint8_t)(l + 32));
            if (icase && l >= 'a' && l <= 'z') set_add(s, (uint8_t)(l - 32));
        }
        out.push_back(s);
    }
    return out;
}

void complement_table(Alphabet ab, uint8_t m[256]) {
    for (int i = 0; i < 256; ++i) m[i] = (uint8_t)i;
    const char *from = nullptr, *to = nullptr;
    if (ab == AB_DNA || ab == AB_DNAredundant) { from = "acgtryswkmbdhvACGTRYSWKMBDHV"; to = "tgcayrswmkvhdbTGCAYRSWMKVHDB"; }
    else if (ab == AB_RNA || ab == AB_RNAredundant) { from = "acguryswkmbdhvACGURYSWKMBDHV"; to = "ugcayrswmkvhdbUGCAYRSWMKVHDB"; }
    if (from) for (size_t k = 0; from[k]; ++k) m[(uint8_t)from[k]] = (uint8_t)to[k];
}

// class pattern that matches on the forward text exactly where the original matches on RevCom(text)
static std::vector<ByteSet> revcom_sets(const std::vector<ByteSet>& s, Alphabet ab) {
    uint8_t comp[256];
    complement_table(ab, comp);
    std::vector<ByteSet> out(s.size());
    for (size_t q = 0; q < s.size(); ++q)
        for (int b = 0; b < 256; ++b)
            if (set_has(s[s.size() - 1 - q], comp[b])) set_add(out[q], (uint8_t)b);
    return out;
}

static std::string read_whole_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw OptError("open " + path + ": no such file or directory");
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

// breader.NewDefaultBufferedReader: one pattern per line, line ends trimmed (grep.go:126-140)
std::vector<std::string> read_pattern_lines(const std::string& path) {
    std::vector<std::string> out;
    const std::string s = read_whole_file(path);
    for (size_t i = 0; i < s.size();) {
        size_t j = s.find('\n', i);
        if (j == std::string::npos) j = s.size();
        size_t e = j;
        while (e > i && (s[e - 1] == '\r' || s[e - 1] == '\n')) --e;
        out.emplace_back(s, i, e - i);
        i = j + 1;
    }
    return out;
}

// fastx.GetSeqsMap(file, seq.Unlimit, ...) (locate.go:86): full name -> sequence, file order (PARITY.md Q11);
// a repeated name keeps the later sequence, like the Go map assignment
static std::vector<std::pair<std::string, std::string>> read_pattern_fasta(const std::string& path) {
    std::vector<std::pair<std::string, std::string>> out;
    const std::string s = read_whole_file(path);
    bool have = false;
    for (size_t i = 0; i < s.size();) {
        size_t j = s.find('\n', i);
        if (j == std::string::npos) j = s.size();
        size_t e = j;
        while (e > i && s[e - 1] == '\r') --e;
        if (e > i && s[i] == '>') {
            const std::string name(s, i + 1, e - i - 1);
            have = true;
            size_t k = 0;
            for (; k < out.size(); ++k) if (out[k].first == name) break;
            if (k < out.size()) out.erase(out.begin() + (long)k);
            out.emplace_back(name, "");
        } else if (have) {
            out.back().second.append(s, i, e - i);
        }
        i = j + 1;
    }
    return out;
}

// ---------------------------------------------------------------------------
// grep  (Grep.Before, bigseqkit-lib/grep.go:41-253)
// ---------------------------------------------------------------------------
void validate_grep_opts(bsk_ctx* c) {
    Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    bool any = !o.s("PatternFile").empty();
    for (auto& p : o.sl("Pattern")) if (!p.empty()) any = true;
    // PARITY.md Q17: the default Pattern [""] must not defeat this guard (grep.go:53)
    if (!any) throw OptError("one of flags -p (--pattern) and -f (--pattern-file) needed");
    // the log lines of Grep.Before (grep.go:57-98), in its order
    for (auto& p : o.sl("Pattern"))
        if (has_unquoted_comma(p)) { c->warn(HELP_UNQUOTED_COMMA); break; }
    if (o.b("Degenerate") && !o.b("BySeq")) c->info("when flag -d (--degenerate) given, flag -s (--by-seq) is automatically on");
    if (o.b("Degenerate")) o.mut("BySeq").b = true;
    if (o.i("MaxMismatch") > 0) {
        if (o.b("UseRegexp") || o.b("Degenerate"))
            throw OptError("flag -r (--use-regexp) or -d (--degenerate) not allowed when giving flag -m (--max-mismatch)");
        if (!o.b("BySeq")) c->info("when value of flag -m (--max-mismatch) > 0, flag -s (--by-seq) is automatically on");
        o.mut("BySeq").b = true;
        if (o.i("MaxMismatch") > 4) c->warn("large value flag -m/--max-mismatch will slow down the search");
    }
    if (o.b("UseRegexp") && o.b("Degenerate"))
        throw OptError("could not give both flags -d (--degenerate) and -r (--use-regexp)");
    c->region_on = false;
    if (!o.s("Region").empty()) {
        c->region_on = true;
        if (!o.b("BySeq")) c->info("when flag -R (--region) given, flag -s (--by-seq) is automatically on");
        o.mut("BySeq").b = true;
        parse_region_opt(o.s("Region"), "grep", &c->region_start, &c->region_end);
    }
    c->patterns.clear();
    c->regexes.clear();
    c->grep_vm = false;
    c->pattern_cls.clear();
    c->max_mm = (int)o.i("MaxMismatch");
    c->general = o.b("Degenerate") || c->max_mm > 0;
    c->patterns_uploaded = false;
    // grep.go:122-252: the pattern file replaces -p when given
    const std::vector<std::string> given = !o.s("PatternFile").empty() ? read_pattern_lines(o.s("PatternFile")) : o.sl("Pattern");
    std::unordered_set<std::string> seen;
    const bool default_id_re = o.cs("IDRegexp") == "^(\\S+)\\s?" && !o.cb("IDNCBI");
    for (std::string p : given) {
        if (p.empty()) continue;
        // grep.go:140-147, 199-207 (unless --quiet)
        if (p[0] == '>') c->warn("symbol \">\" detected, it should not be a part of the sequence ID/name: " + p, true);
        else if (p[0] == '@') c->warn("symbol \"@\" detected, it should not be a part of the sequence ID/name. " + p, true);
        else if (!o.b("ByName") && default_id_re && p.find_first_of("\t ") != std::string::npos)
            c->warn("space found in pattern, you may need use -n/--by-name: " + p, true);
        if (o.b("UseRegexp")) {  // grep.go:148-153, 211-225: "(?i)" + p with -i, then regexp.Compile
            if (o.b("IgnoreCase")) p = "(?i)" + p;
            if (!seen.insert(p).second) continue;
            try {
                c->regexes.push_back(compile_regex(p));
            } catch (const OptError& e) {
                if (std::string(e.what()).rfind("libbsk:", 0) != 0) throw;  // a syntax error is one in any engine
                c->grep_vm = true;  // \b, more than 64 positions: the thread-list matcher takes what the automaton does not
            }
            c->patterns.push_back(p);
            continue;
        }
        if (o.b("Degenerate")) {
            // Degenerate2Regexp with the alphabet of -t (nil for auto => nucleotide map), "(?i)" with -i
            if (!seen.insert(p).second) continue;
            c->pattern_cls.push_back(class_sets(p, true, c->alphabet == AB_PROTEIN, o.b("IgnoreCase")));
            c->patterns.push_back(p);
            continue;
        }
        if (o.b("BySeq")) {
            if (c->max_mm > 0 && c->max_mm > (int)p.size()) throw OptError("mismatch should be <= length of sequence: " + p);
            const uint8_t* b = (const uint8_t*)p.data();
            if (!(alphabet_valid_letters(AB_DNAredundant, b, p.size()) || alphabet_valid_letters(AB_RNAredundant, b, p.size()) ||
                  alphabet_valid_letters(AB_PROTEIN, b, p.size())))
                throw OptError("illegal DNA/RNA/Protein sequence: " + p);
        }
        if (o.b("IgnoreCase"))
            for (auto& ch : p) if (ch >= 'A' && ch <= 'Z') ch += 32;
        if (!seen.insert(p).second) continue;
        if (c->general) c->pattern_cls.push_back(class_sets(p, false, false, o.b("IgnoreCase")));
        c->patterns.push_back(p);
    }
    if (c->grep_vm) {  // one engine for all expressions of the call (compile_vm throws what it cannot take either)
        c->regexes.clear();
        c->vm_progs.clear();
        for (auto& p : c->patterns) c->vm_progs.push_back(compile_vm(p));
    }
    if (!o.s("PatternFile").empty()) {  // grep.go:191-197 (unless --quiet; a warning when the file held none)
        const size_t np = c->patterns.size();
        const std::string m = std::to_string(np) + " patterns loaded from file";
        if (np == 0) c->warn(m, true); else c->info(m, true);
    }
    if (o.b("DeleteMatched") && !o.b("InvertMatch")) {  // PARITY.md DEL
        // with -m the reference takes grepBySeqMismatches (grep.go:255-365), which never deletes a pattern, and the driver
        // returns its records as they are (bigseqkit/grep.go:141-143): --delete-matched is a no-op there
        if (o.b("BySeq") && c->max_mm > 0) o.mut("DeleteMatched").b = false;
        const size_t np = c->patterns.size();
        if ((o.b("BySeq") || o.b("UseRegexp")) && np > 255)  // (15 per hit-bit array, 17 arrays; round 2 stopped at 15)
            throw OptError("libbsk: --delete-matched with more than 255 sequence / regexp patterns is not provided");
    }
}

static std::string revcom_pattern(const std::string& p, Alphabet ab) {
    uint8_t m[256];
    complement_table(ab, m);
    std::string r(p.rbegin(), p.rend());
    for (auto& ch : r) ch = (char)m[(uint8_t)ch];
    return r;
}

// class sets of all patterns (forward, then reverse-complemented when `rc`), 8 dwords per position
static int upload_classes(bsk_ctx* c, bool rc, Alphabet ab, hipStream_t st) {
    std::vector<uint32_t> flat;
    for (int pass = 0; pass < (rc ? 2 : 1); ++pass)
        for (auto& sets : c->pattern_cls) {
            const std::vector<ByteSet> use = pass ? revcom_sets(sets, ab) : sets;
            for (auto& s : use) flat.insert(flat.end(), s.begin(), s.end());
        }
    int r = grow(c, &c->d_cls, &c->cls_cap, flat.size() + 8);
    if (r != BSK_OK) return r;
    if (!flat.empty()) HIP_TRYX(c, hipMemcpyAsync(c->d_cls, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    return BSK_OK;
}

// open-addressing set of the ID / name patterns keyed by fnv1a64 (pattern_match_dev.hpp)
static int upload_pattern_set(bsk_ctx* c, hipStream_t st) {
    uint64_t slots = 16;
    while (slots < 2 * c->patterns.size()) slots <<= 1;
    std::vector<uint64_t> keys(slots, 0);
    std::vector<uint32_t> idx(slots, 0);
    for (size_t k = 0; k < c->patterns.size(); ++k) {
        const std::string& p = c->patterns[k];
        uint64_t h = 1469598103934665603ull;
        for (unsigned char ch : p) h = (h ^ ch) * 1099511628211ull;  // patterns are already lower-cased with -i
        if (!h) h = 1;
        uint64_t s = h & (slots - 1);
        while (keys[s]) s = (s + 1) & (slots - 1);
        keys[s] = h;
        idx[s] = (uint32_t)k;
    }
    int r = grow(c, &c->d_set_keys, &c->set_keys_cap, slots);
    if (r != BSK_OK) return r;
    r = grow(c, &c->d_set_idx, &c->set_idx_cap, slots);
    if (r != BSK_OK) return r;
    HIP_TRYX(c, hipMemcpyAsync(c->d_set_keys, keys.data(), slots * 8, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipMemcpyAsync(c->d_set_idx, idx.data(), slots * 4, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    c->set_slots = slots;
    return BSK_OK;
}

static std::string pattern_signature(const std::vector<std::string>& all) {
    std::string sig;
    for (auto& p : all) { sig += p; sig.push_back('\x01'); }
    return sig;
}

static int upload_patterns(bsk_ctx* c, const std::vector<std::string>& all, hipStream_t st) {
    // the patterns of a context change only with the alphabet of the shard (the reverse strand): a call that needs what
    // the device holds already uploads nothing (two copies and a synchronisation per call, twice per call in `grep -s`)
    const std::string sig = pattern_signature(all);
    if (c->d_pat && c->d_pat_off && sig == c->pat_sig) return BSK_OK;
    c->pat_sig.clear();
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> off{0};
    for (auto& p : all) {
        bytes.insert(bytes.end(), p.begin(), p.end());
        off.push_back((uint32_t)bytes.size());
    }
    int rc = grow(c, &c->d_pat, &c->pat_cap, bytes.size() + 16);
    if (rc != BSK_OK) return rc;
    rc = grow(c, &c->d_pat_off, &c->pat_off_cap, off.size());
    if (rc != BSK_OK) return rc;
    if (!bytes.empty()) HIP_TRYX(c, hipMemcpyAsync(c->d_pat, bytes.data(), bytes.size(), hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipMemcpyAsync(c->d_pat_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRYX(c, hipStreamSynchronize(st));
    c->pat_sig = sig;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// the fused pattern filter (stream_filter.hip): host side
// ---------------------------------------------------------------------------
// `all` = the pattern strings as uploaded to c->d_pat (forward, then reverse-complemented); the first `nuse` of them
// are searched.  Builds the collision-free pair-hash table and uploads it.  false: not applicable (pattern lengths,
// too many patterns, no collision-free table found, BSK_FILTER=off) -- the caller keeps the record-table path.
static bool make_filter(bsk_ctx* c, const std::vector<std::string>& all, size_t nuse, bool invert, bool icase, hipStream_t st,
                        FilterDev* F, int* rc) {
    *rc = BSK_OK;
    const char* env = c->tune.get("filter");
    if (env && strcmp(env, "off") == 0) return false;
    if (nuse == 0 || nuse * 4 > FILTER_MAX_ENTRIES) return false;
    for (size_t k = 0; k < nuse; ++k)
        if (all[k].size() < FILTER_MIN_LEN || all[k].size() > FILTER_MAX_LEN) return false;
    const size_t o_ent = 512 * 4, o_pat = o_ent + FILTER_MAX_ENTRIES * 2;
    auto bind = [&]() {
        F->t1 = reinterpret_cast<const uint32_t*>(c->d_ftab);
        F->ent = reinterpret_cast<const uint16_t*>(c->d_ftab + o_ent);
        F->pat_padded = reinterpret_cast<const uint32_t*>(c->d_ftab + o_pat);
        F->ignore_case = icase ? 1 : 0;
        F->invert = invert ? 1 : 0;
    };
    const std::string sig = pattern_signature(all) + "#" + std::to_string(nuse);
    if (c->d_ftab && sig == c->ftab_sig) { bind(); return true; }  // (the table of the last call: nothing to upload)
    c->ftab_sig.clear();
    std::vector<uint32_t> tab(512, 0u);  // T1 ++ T2
    std::vector<uint16_t> ent(FILTER_MAX_ENTRIES, 0);
    std::vector<uint8_t> padded(FILTER_MAX_PATTERNS * FILTER_MAX_LEN, 0);
    uint32_t e = 0;
    for (size_t k = 0; k < nuse; ++k) {
        memcpy(padded.data() + k * FILTER_MAX_LEN, all[k].data(), all[k].size());
        for (uint32_t j = 0; j < 4; ++j, ++e) {
            uint32_t first, second;
            memcpy(&first, all[k].data() + j, 4);  // little-endian dwords, as the kernel loads the text
            memcpy(&second, all[k].data() + j + 4, 4);
            tab[filter_code(first)] |= 1u << e;
            tab[256 + filter_code(second)] |= 1u << e;
            ent[e] = (uint16_t)(k | (j << 5) | (all[k].size() << 8));  // (FILTER_MAX_LEN = 64 fits the high byte)
        }
    }
    int r = grow(c, &c->d_ftab, &c->ftab_cap, o_pat + padded.size() + 64);
    if (r != BSK_OK) { *rc = r; return false; }
    hipError_t he = hipMemcpyAsync(c->d_ftab, tab.data(), 512 * 4, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(c->d_ftab + o_ent, ent.data(), FILTER_MAX_ENTRIES * 2, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(c->d_ftab + o_pat, padded.data(), padded.size(), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);  // the vectors live in this frame
    if (he != hipSuccess) { c->set_error(std::string("hipMemcpy: ") + hipGetErrorString(he)); *rc = BSK_ERR_HIP; return false; }
    c->ftab_sig = sig;
    bind();
    return true;
}

int grep_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    const bool fastq = format == BSK_FORMAT_FASTQ;
    c->last_count = 0;
    int rc = BSK_OK;
    // exact sequence patterns on FASTQ: the streaming pass itself selects the records (stream_filter.hip) and the
    // per-record kernels below run on the selected ones only.  Everything else (and any shard on which the filter gives
    // up) goes through the table of all records.
    bool filtered = false;
    if (fastq && n > 0 && o.b("BySeq") && !c->general && c->regexes.empty() && !c->grep_vm && !c->region_on && !o.b("Circular") &&
        !o.b("DeleteMatched") && !c->patterns.empty()) {
        Alphabet fab = partition_alphabet(c, d_buf, n, format, st, &rc);
        if (rc != BSK_OK) return rc;
        if (fab == AB_NONE) fab = AB_UNLIMIT;
        const bool both = !(o.b("OnlyPositiveStrand") || fab == AB_UNLIMIT || fab == AB_PROTEIN);
        std::vector<std::string> all = c->patterns;
        if (both)
            for (auto& p : c->patterns) all.push_back(revcom_pattern(p, fab));
        rc = upload_patterns(c, all, st);
        if (rc != BSK_OK) return rc;
        FilterDev F;
        if (make_filter(c, all, all.size(), o.b("InvertMatch"), o.b("IgnoreCase"), st, &F, &rc)) {
            rc = build_index_filtered(c, d_buf, n, format, st, &F);
            if (rc == BSK_OK) filtered = true;
            else if (rc != BSK_ERR_FILTER_FALLBACK) return rc;
        } else if (rc != BSK_OK) return rc;
    }
    if (!filtered) rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    uint64_t total = 0, kept = 0;
    TextTableH tt{nullptr, nullptr, nullptr};
    if (c->table.n > 0) {
        Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);
        if (rc != BSK_OK) return rc;
        if (ab == AB_NONE) ab = AB_UNLIMIT;
        GrepParams G;
        memset(&G, 0, sizeof G);
        G.fastq = fastq;
        G.by_seq = o.b("BySeq");
        G.by_name = o.b("ByName");
        G.invert = o.b("InvertMatch");
        G.ignore_case = o.b("IgnoreCase");
        G.circular = o.b("Circular") && !c->region_on;  // the region branch wins (grep.go:447-456)
        G.region_on = c->region_on;
        G.region_start = c->region_start;
        G.region_end = c->region_end;
        // grep.go:404-409: protein / unlimit sequences are searched on the '+' strand only
        const bool only_pos = o.b("OnlyPositiveStrand") || ab == AB_UNLIMIT || ab == AB_PROTEIN;
        G.both_strands = G.by_seq && !only_pos;
        G.id_mode = id_mode_of(c);
        G.line_width = fastq ? 0 : (int)o.ci("LineWidth");
        G.npat = (int)c->patterns.size();
        // (uses d_out_len as scratch: before the match kernel.)  A search in the sequences reads them many times at
        // arbitrary offsets: wrapped FASTA records get a linear copy first (the emit below goes back to the views)
        const bool flat_text = !fastq && G.by_seq;
        rc = prepare_text(c, d_buf, format, st, &tt, flat_text, false, n);
        if (rc != BSK_OK) return rc;
        if (!c->regexes.empty() || c->grep_vm) {
            if (!c->patterns_uploaded && c->grep_vm) {
                rc = grow(c, &c->d_vm_progs, &c->vm_progs_cap, c->vm_progs.size());
                if (rc != BSK_OK) return rc;
                HIP_TRYX(c, hipMemcpyAsync(c->d_vm_progs, c->vm_progs.data(), c->vm_progs.size() * sizeof(VmProgram), hipMemcpyHostToDevice, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                c->patterns_uploaded = true;
            } else if (!c->patterns_uploaded) {
                rc = grow(c, &c->d_regex, &c->regex_cap, c->regexes.size());
                if (rc != BSK_OK) return rc;
                HIP_TRYX(c, hipMemcpyAsync(c->d_regex, c->regexes.data(), c->regexes.size() * sizeof(RegexProgram),
                                           hipMemcpyHostToDevice, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                c->patterns_uploaded = true;
            }
            uint8_t comp[256];
            complement_table(ab, comp);
            if (!c->d_lut) HIP_TRYX(c, hipMalloc((void**)&c->d_lut, 256));
            HIP_TRYX(c, hipMemcpyAsync(c->d_lut, comp, 256, hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));  // comp lives on the host stack
            if (c->grep_vm) G.vm = c->d_vm_progs; else G.regex = c->d_regex;
            G.comp = c->d_lut;
        } else if (!G.by_seq) {
            // ID / name: the patterns do not depend on the shard, upload once per context
            if (!c->patterns_uploaded) {
                rc = upload_patterns(c, c->patterns, st);
                if (rc != BSK_OK) return rc;
                if (c->patterns.size() > 8) {
                    rc = upload_pattern_set(c, st);
                    if (rc != BSK_OK) return rc;
                } else c->set_slots = 0;
                c->patterns_uploaded = true;
            }
            if (c->set_slots) { G.set_keys = c->d_set_keys; G.set_idx = c->d_set_idx; G.set_mask = c->set_slots - 1; }
        } else {
            std::vector<std::string> all = c->patterns;
            if (G.both_strands)
                for (auto& p : c->patterns) all.push_back(revcom_pattern(p, ab));
            rc = upload_patterns(c, all, st);
            if (rc != BSK_OK) return rc;
            if (c->general) {
                rc = upload_classes(c, G.both_strands, ab, st);
                if (rc != BSK_OK) return rc;
                G.general = 1;
                G.max_mm = c->max_mm;
                G.cls = c->d_cls;
                size_t longest = 0;
                for (auto& p : all) longest = std::max(longest, p.size());
                G.sa_ok = longest <= 64 && all.size() <= 8 && c->max_mm <= 3 && !G.circular && !c->tune.is("grep_shiftand", "off");
            }
        }
        G.pat = c->d_pat;
        G.pat_off = c->d_pat_off;
        rc = ensure_record_scratch(c);
        if (rc != BSK_OK) return rc;
        if (filtered) {
            // the table holds exactly the records the command prints (the streaming pass verified every occurrence,
            // stream_filter.hip): no second search, only their formatted sizes
            SeqParams FP = format_params(c, fastq);
            FP.buf_end = d_buf + n;
            Timed t(c, "k_seq_size", st);
            HIP_TRYX(c, launch_seq_size(d_buf, c->table, FP, c->d_out_len, c->d_status, st));
        } else if (G.by_seq && G.general && G.sa_ok) {
            // one lane per record (k_grep_shiftand): not for chromosomes
            const char* e = c->tune.get("long_bytes");
            const uint32_t thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
            rc = grow(c, &c->d_long_list, &c->long_list_cap, c->table.n, c->table.n / 8 + 16);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
            HIP_TRYX(c, launch_find_long(c->table.l_seq, c->table.n, thresh, c->d_long_list, c->d_counter + 2, st));
            uint64_t lc[2] = {0, 0};
            HIP_TRYX(c, hipMemcpyAsync(lc, c->d_counter + 2, sizeof lc, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            if (lc[0]) G.sa_ok = 0;
        } else if (G.by_seq && !G.general && !G.regex && !G.vm) {
            // chromosome-sized sequences are searched by whole blocks (k_grep_seq<.., LONG>): list them
            const char* e = c->tune.get("long_bytes");
            const uint32_t thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
            rc = grow(c, &c->d_long_list, &c->long_list_cap, c->table.n, c->table.n / 8 + 16);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
            HIP_TRYX(c, launch_find_long(c->table.l_seq, c->table.n, thresh, c->d_long_list, c->d_counter + 2, st));
            uint64_t lc[2] = {0, 0};
            HIP_TRYX(c, hipMemcpyAsync(lc, c->d_counter + 2, sizeof lc, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            if (lc[0]) {
                rc = grow(c, &c->d_hit_list, &c->hit_list_cap, lc[0], 64);
                if (rc != BSK_OK) return rc;
                HIP_TRYX(c, hipMemsetAsync(c->d_hit_list, 0, lc[0] * sizeof(uint32_t), st));
                G.long_list = c->d_long_list;
                G.long_hit = c->d_hit_list;
                G.long_count = lc[0];
                G.long_max = lc[1];
                G.long_thresh = thresh;
            }
        }
        if (!filtered) HIP_TRYX(c, launch_grep_match(d_buf, n, c->table, &tt, G, c->d_out_len, st, c->avg_record_bytes));
        if (o.b("DeleteMatched") && !G.invert) {
            // grep.go:463-511 + bigseqkit/grep.go:144-156: a pattern is dropped at its first hit and the driver keeps
            // the lowest partition per pattern, so every pattern selects its FIRST record in file order (PARITY.md DEL)
            const uint64_t N = c->table.n;
            const bool exact_key = !G.by_seq && !o.b("UseRegexp");
            if (exact_key) {
                // all records with the ID / name of a hit are hits: "first per pattern" = hit AND first of its key group
                RmDupParams R;
                memset(&R, 0, sizeof R);
                R.fastq = fastq;
                R.by_name = G.by_name;
                R.ignore_case = G.ignore_case;
                R.id_mode = G.id_mode;
                R.line_width = G.line_width;
                R.buf_end = d_buf + n;
                uint64_t cap = 0;
                uint64_t* tk = nullptr;
                rc = key_table(c, N, &cap, &tk, st);
                if (rc != BSK_OK) return rc;
                Arena A;
                const uint64_t o_first = A.take(N * 4);
                rc = arena_reserve(c, &A);
                if (rc != BSK_OK) return rc;
                uint32_t* d_firsts = A.at<uint32_t>(o_first);
                HIP_TRYX(c, launch_rmdup_hash(d_buf, n, c->table, tt, R, c->d_keys, nullptr, st));
                HIP_TRYX(c, launch_rmdup_insert(c->d_keys, N, 0, tk, cap, st));
                HIP_TRYX(c, launch_rmdup_resolve(d_buf, c->table, tt, R, c->d_keys, tk, cap, d_firsts, c->d_status, st));
                HIP_TRYX(c, launch_mask_u32(c->d_out_len, d_firsts, N, st));
            } else if (G.npat == 1) {
                // one pattern: only its first hit survives
                HIP_TRYX(c, hipMemsetAsync(c->d_counter + 3, 0xFF, 8, st));
                HIP_TRYX(c, launch_first_nonzero(c->d_out_len, N, c->d_counter + 3, st));
                HIP_TRYX(c, launch_keep_only(c->d_out_len, N, c->d_counter + 3, st));
            } else {
                // several sequence / regexp patterns (grep.go:463-511): the records are visited in file order, a record
                // is a hit when one of the REMAINING patterns matches it, and that pattern -- the first one in the order
                // the patterns were given (PARITY.md Q11; the reference walks a Go map) -- is dropped.  At most one
                // record per pattern is selected, so the walk is: hit bits of every pattern (one match launch each),
                // then <= npat rounds of "first record after the last selected one that still matches something".
                const int np = G.npat;
                // hit bits in arrays of 15 patterns each (bit k = pattern on '+', bit 16 + k = on '-', bit 31 of array 0 =
                // selected); round 2 had one array and refused more than 15 patterns
                const int nblk = (np + 14) / 15;
                Arena A;
                const uint64_t o_masks = A.take((uint64_t)nblk * N * 4), o_hit = A.take(N * 4);
                rc = arena_reserve(c, &A);
                if (rc != BSK_OK) return rc;
                uint32_t* d_masks = A.at<uint32_t>(o_masks);
                uint32_t* d_hit = A.at<uint32_t>(o_hit);
                HIP_TRYX(c, hipMemsetAsync(d_masks, 0, (uint64_t)nblk * N * 4, st));
                const std::vector<std::string> all_patterns = c->patterns;
                const auto all_cls = c->pattern_cls;
                // the reference asks the '+' strand about every remaining pattern before it turns to the '-' strand
                // (grep.go:420-433)
                const int nstrands = G.both_strands ? 2 : 1;
                for (int k = 0; k < np; ++k) {
                    GrepParams G1 = G;
                    G1.npat = 1;
                    if (G.regex) {
                        G1.regex = c->d_regex + k;
                    } else if (G.vm) {
                        G1.vm = c->d_vm_progs + k;
                    } else {
                        c->patterns.assign(1, all_patterns[k]);
                        if (c->general) c->pattern_cls.assign(1, all_cls[k]);
                        std::vector<std::string> one = c->patterns;
                        if (G.both_strands) one.push_back(revcom_pattern(all_patterns[k], ab));
                        rc = upload_patterns(c, one, st);
                        if (rc == BSK_OK && c->general) rc = upload_classes(c, G.both_strands, ab, st);
                        c->patterns = all_patterns;
                        c->pattern_cls = all_cls;
                        if (rc != BSK_OK) return rc;
                    }
                    for (int sd = 0; sd < nstrands; ++sd) {
                        G1.strand_only = sd + 1;
                        if (G1.long_hit) HIP_TRYX(c, hipMemsetAsync(c->d_hit_list, 0, G.long_count * sizeof(uint32_t), st));
                        HIP_TRYX(c, launch_grep_match(d_buf, n, c->table, &tt, G1, d_hit, st, c->avg_record_bytes));
                        HIP_TRYX(c, launch_or_bit(d_masks + (uint64_t)(k / 15) * N, d_hit, N, 1u << (16 * sd + k % 15), st));
                    }
                }
                std::vector<uint32_t> remaining(nblk);
                for (int b = 0; b < nblk; ++b) remaining[b] = (1u << std::min(15, np - 15 * b)) - 1u;
                uint64_t from = 0;
                auto any_left = [&] { for (uint32_t r : remaining) if (r) return true; return false; };
                while (any_left() && from < N) {
                    // the first record at or after `from` that one of the remaining patterns matches: per array, then the lowest
                    uint64_t idx = ~0ull;
                    std::vector<uint64_t> first(nblk, ~0ull);
                    HIP_TRYX(c, hipMemsetAsync(c->d_counter + 3, 0xFF, 8, st));
                    if (nblk == 1) {
                        HIP_TRYX(c, launch_first_masked(d_masks, N, remaining[0] | (remaining[0] << 16), from, c->d_counter + 3, st));
                        HIP_TRYX(c, hipMemcpyAsync(&idx, c->d_counter + 3, 8, hipMemcpyDeviceToHost, st));
                        HIP_TRYX(c, hipStreamSynchronize(st));
                    } else {
                        for (int b = 0; b < nblk; ++b) {
                            if (!remaining[b]) continue;
                            HIP_TRYX(c, hipMemsetAsync(c->d_counter + 3, 0xFF, 8, st));
                            HIP_TRYX(c, launch_first_masked(d_masks + (uint64_t)b * N, N, remaining[b] | (remaining[b] << 16), from, c->d_counter + 3, st));
                            HIP_TRYX(c, hipMemcpyAsync(&first[b], c->d_counter + 3, 8, hipMemcpyDeviceToHost, st));
                            HIP_TRYX(c, hipStreamSynchronize(st));
                            idx = std::min(idx, first[b]);
                        }
                    }
                    if (idx == ~0ull) break;
                    // the first remaining pattern (in the order given) that matched, '+' strand before '-'
                    std::vector<uint32_t> m(nblk);
                    for (int b = 0; b < nblk; ++b) HIP_TRYX(c, hipMemcpy(&m[b], d_masks + (uint64_t)b * N + idx, 4, hipMemcpyDeviceToHost));
                    int drop = -1;
                    for (int pass = 0; pass < 2 && drop < 0; ++pass)
                        for (int b = 0; b < nblk && drop < 0; ++b) {
                            const uint32_t hit = (pass ? (m[b] >> 16) : m[b]) & remaining[b];
                            if (hit) drop = 15 * b + (__builtin_ffs((int)hit) - 1);
                        }
                    if (drop < 0) break;  // (cannot happen: idx matched something)
                    remaining[drop / 15] &= ~(1u << (drop % 15));
                    m[0] |= 0x80000000u;  // bit 31 of array 0: selected
                    HIP_TRYX(c, hipMemcpy(d_masks + idx, &m[0], 4, hipMemcpyHostToDevice));
                    from = idx + 1;
                }
                HIP_TRYX(c, launch_keep_selected(c->d_out_len, d_masks, N, st));
            }
        }
        rc = finish_sizes(c, st, &total, &kept);  // (ERR_HASH_COLLISION -> BSK_ERR_UNSUPPORTED: kernel_error_to_status)
        if (rc != BSK_OK) return rc;
    } else {
        rc = empty_result(c, out);
        if (rc != BSK_OK) return rc;
    }
    c->last_count = kept;
    if (o.b("Count")) {  // grep.go:526-540: one element holding the decimal count
        const std::string txt = std::to_string(kept) + "\n";
        rc = ensure_out(c, txt.size());
        if (rc != BSK_OK) return rc;
        HIP_TRYX(c, hipMemcpy(c->d_out, txt.data(), txt.size(), hipMemcpyHostToDevice));
        out->d_data = c->d_out;
        out->len = txt.size();
        out->records = 1;
        return BSK_OK;
    }
    out->d_data = nullptr;
    out->len = 0;
    out->records = 0;
    if (total == 0) return BSK_OK;
    SeqParams P = format_params(c, fastq);
    if (fastq) {
        const int rs = try_records_as_slices(c, d_buf, n, P, total, kept, st, out);
        if (rs < 0) return -rs;
        if (rs == 1) return BSK_OK;
    }
    rc = ensure_out(c, total);
    if (rc != BSK_OK) return rc;
    if (!fastq && tt.text_w == c->d_text_w) {  // the search ran on linear copies: the emit reads the wrapped text in place
        rc = prepare_text(c, d_buf, format, st, &tt, false, /*keep_out_len=*/true);
        if (rc != BSK_OK) return rc;
    }
    P.text_w = tt.text_w; P.lin_off = tt.lin_off; P.lin = tt.lin;
    apply_long(c, &P);
    { const int rce = emit_records(c, d_buf, n, P, total, kept, st); if (rce != BSK_OK) return rce; }
    out->d_data = c->d_out;
    out->len = total;
    out->records = kept;
    return BSK_OK;
}

// ---------------------------------------------------------------------------
// locate  (Locate.Before, bigseqkit-lib/locate.go:33-193; exact patterns)
// ---------------------------------------------------------------------------
void validate_locate_opts(bsk_ctx* c) {
    const Options& o = c->opts;
    c->alphabet = alphabet_from_seqtype(o.cs("SeqType"));
    check_id_regexp(c);
    bool any = !o.s("PatternFile").empty();
    for (auto& p : o.sl("Pattern")) if (!p.empty()) any = true;
    if (!any) throw OptError("one of flags -p (--pattern) and -f (--pattern-file) needed");  // PARITY.md Q17
    for (auto& p : o.sl("Pattern"))  // locate.go:50-59
        if (has_unquoted_comma(p)) { c->warn(HELP_UNQUOTED_COMMA); break; }
    if (o.i("MaxMismatch") > 0) {
        if (o.b("Degenerate")) throw OptError("flag -d (--degenerate) not allowed when giving flag -m (--max-mismatch)");
        if (o.b("UseRegexp")) throw OptError("flag -r (--use-regexp) not allowed when giving flag -m (--use-regexp)");
        if (o.b("NonGreedy")) c->info("flag -G (--non-greedy) ignored when giving flag -m (--max-mismatch)", true);  // :68-70
    }
    if (o.b("UseFmi")) {
        if (o.b("Degenerate")) throw OptError("flag -d (--degenerate) ignored when giving flag -F (--use-fmi)");
        if (o.b("UseRegexp")) throw OptError("flag -r (--use-regexp) ignored when giving flag -F (--use-fmi)");
    }
    c->patterns.clear();
    c->pattern_names.clear();
    c->pattern_disp.clear();
    c->pattern_cls.clear();
    c->max_mm = (int)o.i("MaxMismatch");
    c->fmi_order = c->max_mm > 0 || o.b("UseFmi");
    c->general = o.b("Degenerate") || o.b("UseRegexp") || c->fmi_order;
    c->locate_vm = false;
    std::vector<std::pair<std::string, std::string>> given;  // (name, sequence)
    const bool from_file = !o.s("PatternFile").empty();
    if (from_file) {
        given = read_pattern_fasta(o.s("PatternFile"));
        if (given.empty()) throw OptError("no FASTA sequences found in pattern file: " + o.s("PatternFile"));
    } else {
        for (const std::string& p : o.sl("Pattern")) if (!p.empty()) given.emplace_back(p, p);
    }
    // locate.go:96-98 (a pattern file: bytes.Contains(seq, "\t ") -- the two bytes in a row, as written), :143-145 (-p: any)
    for (auto& g : given) {
        if (from_file) { if (g.second.find("\t ") != std::string::npos) c->warn("space found in sequence: " + g.first, true); }
        else if (g.second.find_first_of(" \t") != std::string::npos) c->warn("space found in sequence: '" + g.first + "'", true);
    }
    if (o.b("UseRegexp")) {
        // locate.go:102-121, 153-172: the regexp branch shares the search loop of -d (FindSubmatchIndex from a moving
        // offset).  Expressions that are a fixed-length chain of literals, '.', classes and escapes become class patterns (leftmost-
        // first matching has nothing to choose there, and 16 start positions are tested per step); as soon as one
        // expression has quantifiers, alternation, groups with choices or anchors, ALL of them run on the position-
        // reporting matcher instead (regex_vm.hpp: Go's leftmost-first priorities, matches of any length).
        c->locate_vm = false;
        c->vm_progs.clear();
        std::vector<std::pair<std::string, std::string>> uniq;
        for (auto& g : given) {
            bool seen = false;
            for (auto& u : uniq) seen |= u.first == g.first;
            if (!seen) uniq.push_back(g);
        }
        std::vector<std::vector<ByteSet>> chains;
        for (auto& g : uniq) {
            const std::string expr = o.b("IgnoreCase") ? "(?i)" + g.second : g.second;  // :104-106
            bool chain = false;
            RegexProgram pr;
            try {
                pr = compile_regex(expr);
                chain = pr.npos > 0 && !pr.nullable && pr.first == 1ull && pr.last == (1ull << (pr.npos - 1)) &&
                        pr.accept[RE_SYM_BEGIN] == 0 && pr.accept[RE_SYM_END] == 0;
                for (uint32_t q = 0; chain && q < pr.npos; ++q)
                    chain = pr.follow[q >> 3][1u << (q & 7)] == (q + 1 < pr.npos ? (1ull << (q + 1)) : 0ull);
            } catch (const OptError& e) {
                if (std::string(e.what()).rfind("libbsk:", 0) != 0) throw;  // a syntax error is one in any engine
            }
            if (chain) {
                std::vector<ByteSet> sets(pr.npos);
                for (uint32_t q = 0; q < pr.npos; ++q) {
                    sets[q].fill(0);
                    for (int b = 0; b < 256; ++b)
                        if ((pr.accept[b] >> q) & 1ull) set_add(sets[q], (uint8_t)b);
                }
                chains.push_back(sets);
            } else {
                c->locate_vm = true;
            }
        }
        c->locate_pre.clear();
        if (c->locate_vm) {
            // the position-reporting matcher costs ~35 ns per base and lane; most records hold no match at all, and WHETHER
            // one exists is what the boolean automaton of grep -r answers ten times faster: it goes first (expressions it
            // does not take -- more than 64 positions -- leave the matcher alone with every record)
            try {
                for (auto& g : uniq) c->locate_pre.push_back(compile_regex(o.b("IgnoreCase") ? "(?i)" + g.second : g.second));
            } catch (const OptError&) {
                c->locate_pre.clear();
            }
        }
        for (size_t k = 0; k < uniq.size(); ++k) {
            auto& g = uniq[k];
            c->pattern_names.push_back(g.first);
            c->pattern_disp.push_back(g.second);
            if (c->locate_vm) {
                c->vm_progs.push_back(compile_vm(o.b("IgnoreCase") ? "(?i)" + g.second : g.second));
                c->patterns.push_back("N");  // (the match length comes from the matcher)
                c->pattern_cls.push_back(std::vector<ByteSet>(1, ByteSet{}));
            } else {
                c->pattern_cls.push_back(chains[k]);
                c->patterns.push_back(std::string(chains[k].size(), 'N'));  // carries the match length only
            }
        }
        return;
    }
    for (auto& g : given) {  // locate.go:86-190
        std::string eff = g.second;
        if (!o.b("Degenerate") && o.b("IgnoreCase"))
            for (auto& ch : eff) if (ch >= 'A' && ch <= 'Z') ch += 32;
        const uint8_t* b = (const uint8_t*)eff.data();
        const bool legal = alphabet_valid_letters(AB_DNAredundant, b, eff.size()) || alphabet_valid_letters(AB_RNAredundant, b, eff.size()) ||
                           alphabet_valid_letters(AB_PROTEIN, b, eff.size());
        if (c->max_mm > 0) {
            if (c->max_mm > (int)eff.size()) throw OptError("mismatch should be <= length of sequence: " + g.second);
            if (!legal) throw OptError("illegal DNA/RNA/Protein sequence: " + g.first);
        } else if (!o.b("Degenerate") && (eff.find('.') != std::string::npos || !legal)) {
            throw OptError("illegal DNA/RNA/Protein sequence: " + g.first + ", you may switch on -d/--degenerate or -r/--use-regexp");
        }
        if (std::find(c->pattern_names.begin(), c->pattern_names.end(), g.first) != c->pattern_names.end()) continue;
        if (c->general)
            // -d: Degenerate2Regexp with the alphabet of -t (records of a pattern file are seq.Unlimit => nucleotide map)
            c->pattern_cls.push_back(class_sets(eff, o.b("Degenerate"), !from_file && c->alphabet == AB_PROTEIN,
                                                o.b("IgnoreCase")));
        c->pattern_names.push_back(g.first);
        c->patterns.push_back(eff);
    }
}

int locate_run_device(bsk_ctx* c, const uint8_t* d_buf, size_t n, int format, hipStream_t st, bsk_out* out) {
    const Options& o = c->opts;
    int rc = BSK_OK;
    // exact patterns on FASTQ: only the records that hold an occurrence produce rows, and the streaming filter finds them
    // (stream_filter.hip); k_locate then computes the rows of those records exactly as before
    bool filtered = false;
    if (format == BSK_FORMAT_FASTQ && n > 0 && !c->general && !o.b("UseRegexp") && !o.b("Circular") && !c->patterns.empty() &&
        c->pattern_disp.empty()) {
        Alphabet fab = partition_alphabet(c, d_buf, n, format, st, &rc);
        if (rc != BSK_OK) return rc;
        if (fab == AB_NONE) fab = AB_UNLIMIT;
        std::vector<std::string> all = c->patterns;
        for (auto& p : c->patterns) all.push_back(revcom_pattern(p, fab));
        rc = upload_patterns(c, all, st);
        if (rc != BSK_OK) return rc;
        FilterDev F;
        const size_t nuse = o.b("OnlyPositiveStrand") ? c->patterns.size() : all.size();  // locate.go:669 tests the option only
        if (make_filter(c, all, nuse, false, o.b("IgnoreCase"), st, &F, &rc)) {
            rc = build_index_filtered(c, d_buf, n, format, st, &F);
            if (rc == BSK_OK) filtered = true;
            else if (rc != BSK_ERR_FILTER_FALLBACK) return rc;
        } else if (rc != BSK_OK) return rc;
    }
    if (!filtered) rc = build_index(c, d_buf, n, format, st);
    if (rc != BSK_OK) return rc;
    // header row of partition 0 (locate.go:198-204)
    std::string header;
    const bool tsv = !(o.b("Gtf") || o.b("Bed"));
    if (tsv && c->cur_pid == 0)
        header = o.b("HideMatched") ? "seqID\tpatternName\tpattern\tstrand\tstart\tend\n"
                                    : "seqID\tpatternName\tpattern\tstrand\tstart\tend\tmatched\n";
    uint64_t total = 0, nrows = 0;
    LocateParams P;
    memset(&P, 0, sizeof P);
    TextTableH tt{nullptr, nullptr, nullptr};
    if (c->table.n > 0) {
        Alphabet ab = partition_alphabet(c, d_buf, n, format, st, &rc);
        if (rc != BSK_OK) return rc;
        if (ab == AB_NONE) ab = AB_UNLIMIT;
        rc = prepare_text(c, d_buf, format, st, &tt, /*flatten=*/format != BSK_FORMAT_FASTQ, false, n);  // (see grep)
        if (rc != BSK_OK) return rc;
        P.fastq = format == BSK_FORMAT_FASTQ;
        P.ignore_case = o.b("IgnoreCase");
        P.circular = o.b("Circular");
        P.non_greedy = o.b("NonGreedy");
        P.both_strands = !o.b("OnlyPositiveStrand");  // sic: locate.go:669 tests the option, not the alphabet
        if (c->fmi_order) {  // the FM-index branch does consult the alphabet (locate.go:222-227, 308-310)
            P.both_strands = !(o.b("OnlyPositiveStrand") || ab == AB_UNLIMIT || ab == AB_PROTEIN);
            P.non_greedy = 0;  // "flag -G (--non-greedy) ignored when giving flag -m" (locate.go:67-69)
        }
        P.format = o.b("Gtf") ? 2 : (o.b("Bed") ? 3 : (o.b("HideMatched") ? 1 : 0));
        P.id_mode = id_mode_of(c);
        P.npat = (int)c->patterns.size();
        std::vector<std::string> all = c->patterns;
        for (auto& p : c->patterns) all.push_back(revcom_pattern(p, ab));
        rc = upload_patterns(c, all, st);
        if (rc != BSK_OK) return rc;
        P.pat = c->d_pat;
        P.pat_off = c->d_pat_off;
        if (c->general) {
            rc = upload_classes(c, true, ab, st);
            if (rc != BSK_OK) return rc;
            uint8_t comp[256];
            complement_table(ab, comp);
            if (!c->d_lut) HIP_TRYX(c, hipMalloc((void**)&c->d_lut, 256));
            HIP_TRYX(c, hipMemcpyAsync(c->d_lut, comp, 256, hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));  // comp lives on the host stack
            P.general = 1;
            P.max_mm = c->max_mm;
            P.cls = c->d_cls;
            P.fmi_order = c->fmi_order;
            P.matched_lower = !o.b("Degenerate") && !o.b("UseRegexp") && o.b("IgnoreCase");  // locate.go:430-432 lower-cases the text
            P.comp = c->d_lut;
        }
        {
            std::vector<uint8_t> bytes;
            std::vector<uint32_t> off{0};
            for (auto& p : c->pattern_names) {
                bytes.insert(bytes.end(), p.begin(), p.end());
                off.push_back((uint32_t)bytes.size());
            }
            for (auto& p : c->pattern_disp) {  // after the names: off[npat + k] .. off[npat + k + 1]
                bytes.insert(bytes.end(), p.begin(), p.end());
                off.push_back((uint32_t)bytes.size());
            }
            rc = grow(c, &c->d_names, &c->names_cap, bytes.size() + 16);
            if (rc != BSK_OK) return rc;
            rc = grow(c, &c->d_names_off, &c->names_off_cap, off.size());
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemcpyAsync(c->d_names, bytes.data(), bytes.size(), hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipMemcpyAsync(c->d_names_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
        }
        P.name = c->d_names;
        P.name_off = c->d_names_off;
        if (!c->pattern_disp.empty()) { P.disp = c->d_names; P.disp_off = c->d_names_off + c->pattern_names.size(); }
        rc = ensure_record_scratch(c);
        if (rc != BSK_OK) return rc;
        rc = grow(c, &c->d_hit_list, &c->hit_list_cap, c->table.n, c->table.n / 8 + 16);
        if (rc != BSK_OK) return rc;
        P.hit_list = c->d_hit_list;
        P.hit_count = c->d_counter;
        // chromosome-sized sequences: one wave per (pattern, strand, chunk) cell instead of one group per record
        // (not with --non-greedy, whose search position depends on the previous match)
        uint64_t ncells_total = 0;
        const uint64_t per_cells = (uint64_t)P.npat * (P.both_strands ? 2 : 1);
        bool long_checked = false;
        if (!P.non_greedy && per_cells < 32768 && !c->locate_vm) {  // (the matcher of variable-length -r walks every record itself)  // (cells of one record are counted in 32 bits: chunks <= 2^17)
            const char* e = c->tune.get("long_bytes");
            const uint32_t thresh = e && atoll(e) > 0 ? (uint32_t)atoll(e) : SEQ_LONG_THRESH;
            rc = grow(c, &c->d_long_list, &c->long_list_cap, c->table.n, c->table.n / 8 + 16);
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 4 * sizeof(uint64_t), st));
            HIP_TRYX(c, launch_find_long(c->table.l_seq, c->table.n, thresh, c->d_long_list, c->d_counter + 2, st));
            uint64_t lc[2] = {0, 0};
            HIP_TRYX(c, hipMemcpyAsync(lc, c->d_counter + 2, sizeof lc, hipMemcpyDeviceToHost, st));
            HIP_TRYX(c, hipStreamSynchronize(st));
            long_checked = true;
            if (lc[0]) {
                const uint64_t nl = lc[0];
                auto al = [](uint64_t b) { return (b + 15) & ~15ull; };
                // cell counts -> cellbase (record order does not matter: every record has its own rows)
                const uint64_t o_nc = 0, o_cb = al(nl * 4), meta = o_cb + al((nl + 1) * 8);
                rc = grow(c, &c->d_cellmeta, &c->cellmeta_cap, meta, meta / 8 + 64);
                if (rc != BSK_OK) return rc;
                rc = grow(c, &c->d_scan_tmp, &c->scan_tmp_cap, 2 * ((nl + 2047) / 2048) + 4, 16);
                if (rc != BSK_OK) return rc;
                P.long_list = c->d_long_list;
                P.long_count = nl;
                P.cellbase = (const uint64_t*)(c->d_cellmeta + o_cb);
                HIP_TRYX(c, launch_locate_long_cells(c->table, P, (uint32_t*)(c->d_cellmeta + o_nc), st));
                HIP_TRYX(c, launch_scan_u32((const uint32_t*)(c->d_cellmeta + o_nc), const_cast<uint64_t*>(P.cellbase), nl,
                                            c->d_scan_tmp, st));
                HIP_TRYX(c, hipMemcpyAsync(&ncells_total, P.cellbase + nl, 8, hipMemcpyDeviceToHost, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                const uint64_t o_off = al(ncells_total * 4), need = o_off + al((ncells_total + 1) * 8);
                rc = grow(c, &c->d_cells, &c->cells_cap, need, need / 8 + 64);
                if (rc != BSK_OK) return rc;
                rc = grow(c, &c->d_scan_tmp, &c->scan_tmp_cap, 2 * ((ncells_total + 2047) / 2048) + 4, 16);
                if (rc != BSK_OK) return rc;
                P.cell_bytes = (uint32_t*)c->d_cells;
                P.cell_off = (const uint64_t*)(c->d_cells + o_off);
                P.long_cells = ncells_total;
                P.long_thresh = thresh;
            }
        }
        HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 2 * sizeof(uint64_t), st));
        if (c->locate_vm) {
            rc = grow(c, &c->d_vm_progs, &c->vm_progs_cap, c->vm_progs.size());
            if (rc != BSK_OK) return rc;
            HIP_TRYX(c, hipMemcpyAsync(c->d_vm_progs, c->vm_progs.data(), c->vm_progs.size() * sizeof(VmProgram), hipMemcpyHostToDevice, st));
            bool pre = false;
            // (--circular: an occurrence across the origin is invisible to the boolean pass over the plain text)
            if (!c->locate_pre.empty() && !c->tune.get("locate_nopre") && !P.circular) {
                rc = grow(c, &c->d_regex, &c->regex_cap, c->locate_pre.size());
                if (rc != BSK_OK) return rc;
                HIP_TRYX(c, hipMemcpyAsync(c->d_regex, c->locate_pre.data(), c->locate_pre.size() * sizeof(RegexProgram), hipMemcpyHostToDevice, st));
                GrepParams G;
                memset(&G, 0, sizeof G);
                G.fastq = P.fastq;
                G.by_seq = 1;
                G.both_strands = P.both_strands;
                G.npat = (int)c->locate_pre.size();
                G.regex = c->d_regex;
                G.comp = P.comp;
                HIP_TRYX(c, launch_grep_match(d_buf, n, c->table, &tt, G, c->d_out_len, st, c->avg_record_bytes));  // != 0: some match exists
                pre = true;
                P.pre_regex = c->d_regex;
            }
            uint64_t ncand = 0;
            if (pre) {  // the candidates as a list: the matcher then runs with every lane busy
                HIP_TRYX(c, launch_compact_hits(c->d_out_len, c->table.n, c->d_hit_list, c->d_counter, st));
                HIP_TRYX(c, hipMemcpyAsync(&ncand, c->d_counter, sizeof ncand, hipMemcpyDeviceToHost, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 2 * sizeof(uint64_t), st));
            }
            HIP_TRYX(c, launch_locate_vm(false, d_buf, n, c->table, tt, P, c->d_vm_progs, c->d_out_len, nullptr, nullptr, c->d_counter + 1, st,
                                         pre ? c->d_hit_list : nullptr, ncand));
        } else {
            // -d / -m: whether a record holds an occurrence at all is what grep's Shift-And answers at one table lookup per
            // base; the position-reporting search (one class test per start position and pattern byte) then runs on the
            // few records that do.  Long records have their own cell launches and are not prefiltered.
            size_t longest = 0;
            for (auto& p : all) longest = std::max(longest, p.size());
            if (c->general && long_checked && !P.long_count && !P.circular && longest <= 64 && all.size() <= 8 && c->max_mm <= 3 &&
                !c->tune.get("locate_nopre")) {
                GrepParams G;
                memset(&G, 0, sizeof G);
                G.fastq = P.fastq;
                G.by_seq = 1;
                G.both_strands = P.both_strands;
                G.npat = P.npat;
                G.pat = P.pat;
                G.pat_off = P.pat_off;
                G.general = 1;
                G.max_mm = P.max_mm;
                G.cls = P.cls;
                G.sa_ok = 1;
                HIP_TRYX(c, launch_grep_match(d_buf, n, c->table, &tt, G, c->d_out_len, st, c->avg_record_bytes));
                HIP_TRYX(c, launch_compact_hits(c->d_out_len, c->table.n, c->d_hit_list, c->d_counter, st));
                HIP_TRYX(c, hipMemcpyAsync(&P.ncand, c->d_counter, sizeof P.ncand, hipMemcpyDeviceToHost, st));
                HIP_TRYX(c, hipStreamSynchronize(st));
                HIP_TRYX(c, hipMemsetAsync(c->d_counter, 0, 2 * sizeof(uint64_t), st));
                P.cand = c->d_hit_list;
            }
            if (!P.cand || P.ncand)
                HIP_TRYX(c, launch_locate(false, d_buf, n, c->table, tt, P, c->d_out_len, nullptr, nullptr, c->d_counter + 1, st, c->avg_record_bytes));
            P.cand = nullptr;
        }
        if (P.long_count) {
            // place every cell inside its record's rows, then the record sizes
            HIP_TRYX(c, launch_scan_u32(P.cell_bytes, const_cast<uint64_t*>(P.cell_off), ncells_total, c->d_scan_tmp, st));
            HIP_TRYX(c, launch_locate_long_sizes(P, c->d_out_len, st));
        }
        HIP_TRYX(c, launch_compact_hits(c->d_out_len, c->table.n, c->d_hit_list, c->d_counter, st));
        HIP_TRYX(c, launch_scan_u32(c->d_out_len, c->d_out_off, c->table.n, c->d_scan_tmp, st));
        uint64_t status = 0;
        HIP_TRYX(c, hipMemcpyAsync(&total, c->d_out_off + c->table.n, sizeof total, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipMemcpyAsync(&P.nhit, c->d_counter, sizeof P.nhit, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipMemcpyAsync(&status, c->d_status, sizeof status, hipMemcpyDeviceToHost, st));
        HIP_TRYX(c, hipStreamSynchronize(st));
        rc = kernel_error_to_status(c, status);
        if (rc != BSK_OK) return rc;
    } else {
        rc = empty_result(c, out);
        if (rc != BSK_OK) return rc;
    }
    out->d_data = nullptr;
    out->len = 0;
    out->records = 0;
    if (total + header.size() == 0) return BSK_OK;
    rc = ensure_out(c, total + header.size());
    if (rc != BSK_OK) return rc;
    if (!header.empty()) HIP_TRYX(c, hipMemcpyAsync(c->d_out, header.data(), header.size(), hipMemcpyHostToDevice, st));
    if (total && c->locate_vm)
        HIP_TRYX(c, launch_locate_vm(true, d_buf, n, c->table, tt, P, c->d_vm_progs, c->d_out_len, c->d_out_off, c->d_out + header.size(),
                                     c->d_counter + 1, st, P.hit_list, P.nhit));
    else if (total)
        HIP_TRYX(c, launch_locate(true, d_buf, n, c->table, tt, P, c->d_out_len, c->d_out_off, c->d_out + header.size(),
                                  c->d_counter + 1, st, c->avg_record_bytes));
    if (total) HIP_TRYX(c, hipMemcpyAsync(&nrows, c->d_counter + 1, sizeof nrows, hipMemcpyDeviceToHost, st));  // counted by the emit pass
    HIP_TRYX(c, hipStreamSynchronize(st));  // header lives on the host stack
    out->d_data = c->d_out;
    out->len = total + header.size();
    out->records = nrows + (header.empty() ? 0 : 1);
    return BSK_OK;
}


}  // namespace bsk
