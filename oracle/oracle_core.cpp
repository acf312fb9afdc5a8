// TEST INFRASTRUCTURE -- see the header of oracle_core.hpp ("parity unpinned").
// CPU restatement of bigseqkit-lib parser + stats.  Citations: /root/reference/.
#include "oracle_core.hpp"

#include <regex>
#include <array>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

#include <algorithm>
#include <set>
#include <cmath>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

namespace orc {

// ---------------------------------------------------------------------------
// Alphabets  [upstream-memory: shenwei356/bio v0.7.0 seq/alphabet.go]
// ---------------------------------------------------------------------------
namespace {
struct LetterSet {
    bool has[256];
    explicit LetterSet(const char* s) {
        memset(has, 0, sizeof(has));
        for (; *s; ++s) has[(unsigned char)*s] = true;
    }
    bool subset(std::string_view v) const {
        for (unsigned char c : v)
            if (!has[c]) return false;
        return true;
    }
};
// AllLetters() = letters + gap + ambiguous
const LetterSet kDNA("acgtACGT -.nN");
const LetterSet kRNA("acguACGU -.nN");
const LetterSet kDNAr("acgtryswkmbdhvACGTRYSWKMBDHV -.nN");
const LetterSet kRNAr("acguryswkmbdhvACGURYSWKMBDHV -.nN");
const LetterSet kProt("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ -xX*_.");
}  // namespace

const char* alphabet_name(Alphabet a) {
    switch (a) {
        case AB_DNA: return "DNA";
        case AB_DNAredundant: return "DNAredundant";
        case AB_RNA: return "RNA";
        case AB_RNAredundant: return "RNAredundant";
        case AB_PROTEIN: return "Protein";
        case AB_UNLIMIT: return "Unlimit";
        default: return "";
    }
}

static Alphabet guess_alphabet(std::string_view s, int thr) {
    if (s.empty()) return AB_UNLIMIT;
    if (thr != 0 && (int64_t)s.size() > thr) s = s.substr(0, thr);
    if (kDNA.subset(s)) return AB_DNA;
    if (kRNA.subset(s)) return AB_RNA;
    if (kDNAr.subset(s)) return AB_DNAredundant;
    if (kRNAr.subset(s)) return AB_RNAredundant;
    if (kProt.subset(s)) return AB_PROTEIN;
    return AB_UNLIMIT;
}

Alphabet guess_alphabet_less_conservatively(std::string_view s, int thr) {
    Alphabet ab = guess_alphabet(s, thr);
    if (ab == AB_DNA) return AB_DNAredundant;
    if (ab == AB_RNA) return AB_RNAredundant;
    return ab;
}

static std::string lower(std::string s) {
    for (auto& c : s)
        if (c >= 'A' && c <= 'Z') c += 32;
    return s;
}

// bigseqkit/helper.go:68-84
Alphabet alphabet_from_seqtype(const std::string& t) {
    std::string v = lower(t);
    if (v == "dna") return AB_DNAredundant;
    if (v == "rna") return AB_RNAredundant;
    if (v == "protein") return AB_PROTEIN;
    if (v == "unlimit") return AB_UNLIMIT;
    if (v == "auto") return AB_NONE;
    throw Error("invalid sequence type: " + t + ", available value: dna|rna|protein|unlimit|auto");
}

bool alphabet_is_valid(Alphabet a, std::string_view s) {
    switch (a) {
        case AB_DNA: return kDNA.subset(s);
        case AB_RNA: return kRNA.subset(s);
        case AB_DNAredundant: return kDNAr.subset(s);
        case AB_RNAredundant: return kRNAr.subset(s);
        case AB_PROTEIN: return kProt.subset(s);
        default: return true;
    }
}

// ---------------------------------------------------------------------------
// Record splitting.  PARITY.md "SPLIT": FASTA records start at a '>' that is
// the first byte of a line; FASTQ records are read sequentially the way seqkit's
// own reader does (header '@', sequence lines until a non-empty line starting
// with '+', then quality lines -- at least one -- until len(qual) >= len(seq)).
// Then ReadFixer (bigseqkit-lib/helper.go:41-66): drop empty elements, strip
// ONE trailing '\n', make sure the element starts with the marker.
// ---------------------------------------------------------------------------
std::vector<std::string_view> split_records(std::string_view buf, bool fastq) {
    std::vector<std::string_view> out;
    const size_t n = buf.size();
    size_t p = 0;
    // leading blank lines are not records
    while (p < n && buf[p] == '\n') ++p;
    if (!fastq) {
        if (p < n && buf[p] != '>') throw Error("invalid FASTA: data before the first '>'");
        while (p < n) {
            size_t q = p + 1;
            // next '>' at line start
            for (;;) {
                const void* f = memchr(buf.data() + q, '>', n - q);
                if (!f) { q = n; break; }
                q = (const char*)f - buf.data();
                if (buf[q - 1] == '\n') break;
                ++q;
            }
            std::string_view e = buf.substr(p, q - p);
            if (!e.empty() && e.back() == '\n') e.remove_suffix(1);  // helper.go:51-56
            if (!e.empty()) out.push_back(e);
            p = q;
        }
        return out;
    }
    auto line_end = [&](size_t s) {  // index of '\n' or n
        const void* f = s < n ? memchr(buf.data() + s, '\n', n - s) : nullptr;
        return f ? (size_t)((const char*)f - buf.data()) : n;
    };
    while (p < n) {
        // trailing blank lines at EOF are ignored
        size_t t = p;
        while (t < n && buf[t] == '\n') ++t;
        if (t == n) break;
        if (buf[p] != '@') throw Error("invalid FASTQ: record does not start with '@'");
        size_t s = p;
        size_t e = line_end(s);  // header
        size_t cur = e < n ? e + 1 : n;
        size_t seqlen = 0, quallen = 0;
        bool isQual = false, any_qual_line = false;
        size_t rec_end = n;
        while (cur < n) {
            size_t le = line_end(cur);
            size_t k = le - cur;
            if (!isQual) {
                if (k > 0 && buf[cur] == '+') isQual = true;  // helper.go:255
                else seqlen += k;
            } else {
                quallen += k;
                any_qual_line = true;
            }
            cur = le < n ? le + 1 : n;
            if (isQual && any_qual_line && quallen >= seqlen) { rec_end = cur; break; }
            // quality shorter than the sequence and the next line looks like a header:
            // the reference's "\n@" split ends the record here and SeqParser then
            // reports the length mismatch (bigseqkit-lib/helper.go:308-311)
            if (isQual && any_qual_line && cur < n && buf[cur] == '@') { rec_end = cur; break; }
        }
        std::string_view el = buf.substr(s, rec_end - s);
        if (!el.empty() && el.back() == '\n') el.remove_suffix(1);
        if (!el.empty()) out.push_back(el);
        p = rec_end;
    }
    return out;
}

// Strict 4-line FASTQ == what the HIP path accepts (PARITY.md "STRICT4"):
// every record is exactly: '@' line, sequence line not starting with '+' unless
// empty... (a non-empty sequence line starting with '+' is rejected), '+' line,
// quality line with len == len(seq).  Optional missing final '\n'.  Trailing
// empty lines at EOF are tolerated.
bool is_strict_4line_fastq(std::string_view buf) {
    const size_t n = buf.size();
    size_t p = 0;
    auto line_end = [&](size_t s) {
        const void* f = s < n ? memchr(buf.data() + s, '\n', n - s) : nullptr;
        return f ? (size_t)((const char*)f - buf.data()) : n;
    };
    if (n > 0 && buf[0] == '\n') return false;  // leading blank lines unsupported
    while (p < n) {
        size_t t = p;
        while (t < n && buf[t] == '\n') ++t;
        if (t == n) return true;
        if (buf[p] != '@') return false;
        size_t h = line_end(p);
        if (h >= n) return false;
        size_t s0 = h + 1, s1 = line_end(s0);
        if (s1 >= n) return false;
        if (s1 > s0 && buf[s0] == '+') return false;
        size_t p0 = s1 + 1, p1 = line_end(p0);
        if (p1 >= n) return false;
        if (p1 == p0 || buf[p0] != '+') return false;
        size_t q0 = p1 + 1, q1 = line_end(q0);
        if (q1 - q0 != s1 - s0) return false;
        p = q1 < n ? q1 + 1 : n;
    }
    return true;
}

// ---------------------------------------------------------------------------
// SeqParser  bigseqkit-lib/helper.go:179-325
// ---------------------------------------------------------------------------
SeqParser::SeqParser(Alphabet t_, const std::vector<std::string_view>* it_, const std::string& idRegexp, int thr)
    : it(it_), t(t_), guess_thr(thr), id_regexp(idRegexp) {
    // helper.go:182-197
    default_id_regexp = idRegexp.empty() || idRegexp == "^(\\S+)\\s?";
    if (!default_id_regexp && idRegexp.find('(') == std::string::npos)
        throw Error("fastx: regular expression must contain \"(\" and \")\" to capture matched ID. default: ^(\\S+)\\s?");
}

// bigseqkit-lib/helper.go:329-369 ; custom regexps: only the NCBI one
// (bigseqkit/helper.go:97-100, `\|([^\|]+)\| `) is restated natively.
void parse_head_id_desc(const std::string& head, bool default_re, const std::string& re, std::string& id,
                        std::string& desc) {
    id.clear();
    desc.clear();
    if (default_re) {
        for (char sep : {' ', '\t'}) {
            size_t i = head.find(sep);
            if (i != std::string::npos && i > 0) {
                size_t e = head.size();
                size_t j = i + 1;
                for (; j < e; j++) {
                    if (head[j] == ' ' || head[j] == '\t') j++;  // sic: skips two per iteration (helper.go:335-339)
                    else break;
                }
                id = head.substr(0, i);
                if (j >= e) return;
                desc = head.substr(j);
                return;
            }
        }
        id = head;
        return;
    }
    if (re == "\\|([^\\|]+)\\| ") {
        // leftmost match of  '|' [^|]+ '|' ' '
        for (size_t a = head.find('|'); a != std::string::npos; a = head.find('|', a + 1)) {
            size_t b = head.find('|', a + 1);
            if (b == std::string::npos) break;
            if (b > a + 1 && b + 1 < head.size() && head[b + 1] == ' ') {
                id = head.substr(a + 1, b - a - 1);
                return;
            }
        }
        id = head;  // not match -> whole head (helper.go:365-367)
        return;
    }
    // any other expression: found = idRegexp.FindSubmatch(head); nil -> head; else found[1] (helper.go:362-368).
    // Go's regexp is not in this image: std::regex (ECMAScript, leftmost with the same alternation / greediness
    // priorities) stands in for it on the syntax the two share -- a different engine from the product's Pike VM.
    if (re.find('(') == std::string::npos || re.rfind(')') == std::string::npos || re.rfind(')') < re.find('(') + 2)
        throw Error("fastx: regular expression must contain \"(\" and \")\" to capture matched ID. default: ^(\\S+)\\s?");
    static std::string cached_text;
    static std::regex cached;
    if (cached_text != re) {
        try { cached = std::regex(re, std::regex::ECMAScript); }
        catch (const std::regex_error&) { throw Error("fastx: fail to compile regexp: " + re); }
        cached_text = re;
    }
    std::smatch m;
    if (!std::regex_search(head, m, cached)) { id = head; return; }
    if (m.size() < 2) throw Error("fastx: regular expression must contain \"(\" and \")\" to capture matched ID. default: ^(\\S+)\\s?");
    id = m[1].matched ? m[1].str() : std::string();
}

bool SeqParser::Read() {
    if (pos >= it->size()) return false;  // io.EOF
    std::string_view p = (*it)[pos++];
    if (firstseq) IsFastq = !p.empty() && p[0] == '@';  // helper.go:228-230
    std::string head;
    rec.seq.clear();
    rec.qual.clear();
    size_t j = p.find('\n');
    if (j != std::string_view::npos && j > 0) {
        head = std::string(p.substr(0, j));
        size_t r = j + 1;
        if (!IsFastq) {  // helper.go:240-250
            for (;;) {
                size_t k = p.find('\n', r);
                if (k != std::string_view::npos) {
                    rec.seq.append(p.substr(r, k - r));
                    r = k + 1;
                    continue;
                }
                rec.seq.append(p.substr(r));
                break;
            }
        } else {  // helper.go:251-273
            bool isQual = false;
            for (;;) {
                size_t kk = p.find('\n', r);
                if (kk != std::string_view::npos) {
                    size_t k = kk - r;
                    if (k > 0 && p[r] == '+' && !isQual) isQual = true;
                    else if (isQual) rec.qual.append(p.substr(r, k));
                    else rec.seq.append(p.substr(r, k));
                    r = kk + 1;
                    continue;
                }
                if (isQual) rec.qual.append(p.substr(r));
                break;  // sic: an unterminated non-quality last line is dropped
            }
        }
    } else {  // helper.go:275-283
        if (!p.empty() && p.back() == '\n') head = std::string(p.substr(0, p.size() - 1));
        else head = std::string(p);
    }
    if (firstseq) {  // helper.go:286-291
        if (t == AB_NONE) t = guess_alphabet_less_conservatively(rec.seq, guess_thr);
        firstseq = false;
    }
    if (head.empty() && rec.seq.empty()) return false;  // helper.go:293-295
    // PARITY.md Q1: the marker byte is not part of the name.
    if (!head.empty() && (head[0] == '>' || head[0] == '@')) head.erase(0, 1);
    rec.name = head;
    parse_head_id_desc(head, default_id_regexp, id_regexp, rec.id, rec.desc);
    if (IsFastq && rec.seq.size() != rec.qual.size()) {  // helper.go:308-311
        char b[256];
        snprintf(b, sizeof b, "seq('%s'): unmatched length of sequence (%zu) and quality (%zu)", rec.name.c_str(),
                 rec.seq.size(), rec.qual.size());
        throw Error(b);
    }
    return true;
}

// bigseqkit-lib/helper.go:81-117
std::string wrap_byte_slice(std::string_view s, int width) {
    if (width < 1) return std::string(s);
    size_t l = s.size();
    if (l == 0) return std::string();
    size_t w = (size_t)width;
    size_t lines = (l % w == 0) ? l / w - 1 : l / w;
    std::string out;
    out.reserve(l + lines);
    for (size_t i = 0; i <= lines; i++) {
        size_t start = i * w, end = std::min((i + 1) * w, l);
        out.append(s.substr(start, end - start));
        if (i < lines) out.push_back('\n');
    }
    return out;
}

std::string record_format(const Record& r, bool fastq, int width) {
    std::string o;
    o.push_back(fastq ? '@' : '>');
    o += r.name;
    o.push_back('\n');
    o += wrap_byte_slice(r.seq, width);
    o.push_back('\n');
    if (fastq) {
        o += "+\n";
        o += wrap_byte_slice(r.qual, width);
        o.push_back('\n');
    }
    return o;
}

// bigseqkit-lib/helper.go:119-136 + bio QualityEncoding.Offset [upstream-memory]
int quality_offset(const std::string& enc) {
    std::string s = lower(enc);
    if (s == "sanger" || s == "illumina-1.8+") return 33;
    if (s == "solexa" || s == "illumina-1.3+" || s == "illumina-1.5+") return 64;
    if (s == "") return 0;
    throw Error("unsupported quality encoding: " + enc +
                ". available values: 'sanger', 'solexa', 'illumina-1.3+', 'illumina-1.5+', 'illumina-1.8+'");
}

// ---------------------------------------------------------------------------
// Stats  bigseqkit-lib/stats.go
// ---------------------------------------------------------------------------
static void check_gap_letters(const std::string& g) {  // stats.go:36-43
    if (g.empty()) throw Error("value of flag -G (--gap-letters) should not be empty");
    for (unsigned char c : g)
        if (c > 127) throw Error("value of -G (--gap-letters) contains non-ASCII characters");
}

std::map<int64_t, int64_t> stats_call(const std::vector<std::string_view>& part, const StatsOptions& o) {
    Alphabet ab = alphabet_from_seqtype(o.Config.SeqType);
    check_gap_letters(o.GapLetters);
    SeqParser rd(ab, &part, o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
    bool gap[256] = {false};
    for (unsigned char c : o.GapLetters) gap[c] = true;
    const int off = quality_offset(o.FqEncoding);
    std::string seqFormat;
    std::map<int64_t, int64_t> result;
    const int64_t Q20 = -1, Q30 = -2, GAP_SUM = -3, T = -4;
    while (rd.Read()) {
        const Record& r = rd.rec;
        if (seqFormat.empty()) seqFormat = r.qual.empty() ? "FASTA" : "FASTQ";  // stats.go:80-86
        result[(int64_t)r.seq.size()]++;                                        // stats.go:88
        if (o.All) {
            if (rd.IsFastq) {
                for (unsigned char q : r.qual) {  // stats.go:91-100
                    if ((int)q - off >= 20) {
                        result[Q20]++;
                        if ((int)q - off >= 30) result[Q30]++;
                    }
                }
            }
            int64_t g = 0;
            for (unsigned char c : r.seq) g += gap[c];
            result[GAP_SUM] += g;  // stats.go:102
        }
    }
    Alphabet fa = rd.GetAlphabet();  // stats.go:106-114
    if (fa == AB_DNAredundant) result[T] = 'D';
    else if (fa == AB_RNAredundant) result[T] = 'R';
    else if (seqFormat.empty() && fa == AB_UNLIMIT) result[T] = 'U';
    else result[T] = 'F';
    return result;
}

// stats.go:128-137 restated with PARITY.md Q2: counts are summed; the type key
// (-4) keeps the value of the lower-index partition.
std::map<int64_t, int64_t> stats_reduce(const std::map<int64_t, int64_t>& a, const std::map<int64_t, int64_t>& b) {
    std::map<int64_t, int64_t> r = a;
    for (auto& kv : b) {
        if (kv.first == -4) {
            if (!r.count(-4) || r[-4] == 'U') r[-4] = kv.second;
        } else {
            r[kv.first] += kv.second;
        }
    }
    return r;
}

// ---------------------------------------------------------------------------
// SeqTransform  bigseqkit-lib/seq.go
// ---------------------------------------------------------------------------
// seq.Seq.ComplementInplace [upstream-memory]: PairLetter per byte, letters that
// are not in the alphabet stay as they are; Unlimit/Protein are identities.
void complement_inplace(std::string& s, Alphabet a) {
    const char *from, *to;
    if (a == AB_DNA || a == AB_DNAredundant) {
        from = "acgtryswkmbdhvACGTRYSWKMBDHV";
        to = "tgcayrswmkvhdbTGCAYRSWMKVHDB";
    } else if (a == AB_RNA || a == AB_RNAredundant) {
        from = "acguryswkmbdhvACGURYSWKMBDHV";
        to = "ugcayrswmkvhdbUGCAYRSWMKVHDB";
    } else {
        return;
    }
    unsigned char lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = (unsigned char)i;
    for (size_t i = 0; from[i]; ++i) lut[(unsigned char)from[i]] = (unsigned char)to[i];
    for (auto& c : s) c = (char)lut[(unsigned char)c];
}

// seq.Seq.AvgQual [upstream-memory]: mean error probability, sequential float64 sum
double avg_qual(const std::string& qual, int base) {
    if (qual.empty()) return 0;
    double sum = 0;
    for (unsigned char q : qual) sum += std::pow(10.0, (double)((int)q - base) / -10.0);
    return -10.0 * std::log10(sum / (double)qual.size());
}

std::vector<std::string> seq_call(const std::vector<std::string_view>& part, const SeqOptions& oin) {
    SeqOptions o = oin;
    // ---- Before (seq.go:28-79)
    Alphabet ab = alphabet_from_seqtype(o.Config.SeqType);
    if (o.GapLetters.empty()) throw Error("value of flag -G (--gap-letters) should not be empty");
    for (unsigned char c : o.GapLetters)
        if (c > 127) throw Error("value of -G (--gap-letters) contains non-ASCII characters");
    if (o.MinLen >= 0 && o.MaxLen >= 0 && o.MinLen > o.MaxLen)
        throw Error("value of flag -m (--min-len) should be >= value of flag -M (--max-len)");
    if (o.MinQual >= 0 && o.MaxQual >= 0 && o.MinQual > o.MaxQual)
        throw Error("value of flag -Q (--min-qual) should be <= value of flag -R (--max-qual)");
    bool validate = o.ValidateSeq;
    if (!validate && !(ab == AB_NONE || ab == AB_UNLIMIT)) validate = true;  // seq.go:66-72
    if (o.LowerCase && o.UpperCase) throw Error("could not give both flags -l (--lower-case) and -u (--upper-case)");
    // ---- Call (seq.go:81-269)
    SeqParser rd(ab, &part, o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
    const bool filterMinLen = o.MinLen > 0, filterMaxLen = o.MaxLen > 0;
    const bool filterMinQual = o.MinQual > 0, filterMaxQual = o.MaxQual > 0;
    bool isFastq = false, printName, printSeq, printQual = false, checkSeqType = true;
    int lineWidth = o.Config.LineWidth;
    if (o.Seq || o.Qual) lineWidth = 0;  // seq.go:106-108
    bool gapset[256] = {false};
    for (unsigned char c : o.GapLetters) gapset[c] = true;
    std::vector<std::string> result;
    while (rd.Read()) {
        Record& r = rd.rec;
        if (validate) {  // SeqParser validates with parser.t (helper.go:304-306,318-320)
            std::string_view v = r.seq;
            if (o.ValidateSeqLength > 0 && (int64_t)v.size() > o.ValidateSeqLength) v = v.substr(0, o.ValidateSeqLength);
            if (!alphabet_is_valid(rd.GetAlphabet(), v)) throw Error("seq: invalid " + std::string(alphabet_name(rd.GetAlphabet())) + " letter");
        }
        if (checkSeqType) {
            isFastq = rd.IsFastq;
            if (isFastq) { lineWidth = 0; printQual = true; }
            checkSeqType = false;
        }
        if (o.RemoveGaps) {  // Seq.RemoveGapsInplace [upstream-memory]: seq and qual together
            std::string s2, q2;
            for (size_t i = 0; i < r.seq.size(); ++i)
                if (!gapset[(unsigned char)r.seq[i]]) {
                    s2.push_back(r.seq[i]);
                    if (i < r.qual.size()) q2.push_back(r.qual[i]);
                }
            r.seq.swap(s2);
            if (!r.qual.empty()) r.qual.swap(q2);
        }
        if (filterMinLen && (int64_t)r.seq.size() < o.MinLen) continue;
        if (filterMaxLen && (int64_t)r.seq.size() > o.MaxLen) continue;
        if (filterMinQual || filterMaxQual) {
            double aq = avg_qual(r.qual, o.QualAsciiBase);
            if (filterMinQual && aq < o.MinQual) continue;
            if (filterMaxQual && aq >= o.MaxQual) continue;
        }
        printName = true; printSeq = true;
        if (o.Name && o.Seq) { printName = true; printSeq = true; }
        else if (o.Name) { printName = true; printSeq = false; printQual = false; }
        else if (o.Seq) { printName = false; printSeq = true; printQual = false; }
        else if (o.Qual) {
            if (!isFastq) throw Error("FASTA format has no quality. So do not just use flag -q (--qual)");
            printName = false; printSeq = false; printQual = true;
        }
        std::string out;
        if (printName) {
            const std::string& head = o.OnlyId ? r.id : r.name;
            if (printSeq) out.push_back(isFastq ? '@' : '>');
            out += head;
            out.push_back('\n');
        }
        if (o.Reverse) {  // ReverseInplace: sequence and quality
            std::reverse(r.seq.begin(), r.seq.end());
            std::reverse(r.qual.begin(), r.qual.end());
        }
        if (o.Complement) complement_inplace(r.seq, rd.GetAlphabet());
        if (printSeq) {
            Alphabet fa = rd.GetAlphabet();
            if (o.Dna2rna && !(fa == AB_RNA || fa == AB_RNAredundant))
                for (auto& c : r.seq) { if (c == 't') c = 'u'; else if (c == 'T') c = 'U'; }
            if (o.Rna2dna && !(fa == AB_DNA || fa == AB_DNAredundant))
                for (auto& c : r.seq) { if (c == 'u') c = 't'; else if (c == 'U') c = 'T'; }
            if (o.LowerCase) { for (auto& c : r.seq) if (c >= 'A' && c <= 'Z') c += 32; }
            else if (o.UpperCase) { for (auto& c : r.seq) if (c >= 'a' && c <= 'z') c -= 32; }
            if (isFastq) out += r.seq;
            else out += wrap_byte_slice(r.seq, lineWidth);
            out.push_back('\n');
        }
        if (printQual) {
            if (!o.Qual) out += "+\n";
            out += r.qual;
            out.push_back('\n');
        }
        if (!out.empty() && out.back() == '\n') out.pop_back();  // seq.go:261-265
        result.push_back(out);
    }
    return result;
}

// ---------------------------------------------------------------------------
// regions  (Seq.SubSeq / SubLocation [upstream-memory]; KATs bigseqkit-cli/helper.go:348-361)
// ---------------------------------------------------------------------------
void sub_location(size_t length, int start, int end, size_t* b, size_t* e) {
    *b = *e = 0;
    if (length == 0) return;
    long L = (long)length, s = start, t = end;
    if (s < 0) { s = L + s + 1; if (s < 1) s = 1; }
    if (s == 0) s = 1;
    if (t < 0) t = L + t + 1;
    if (t > L) t = L;
    if (s > L || t < 1 || s > t) return;
    *b = (size_t)(s - 1);
    *e = (size_t)t;
}

// reRegion `\-?\d+:\-?\d+` + the checks at bigseqkit-lib/grep.go:103-118, subseq.go:83-97
void parse_region(const std::string& region, const char* cmd, int* start, int* end) {
    // leftmost match of -?digits:-?digits anywhere in the string (MatchString), then Split(":")
    bool ok = false;
    for (size_t i = 0; i < region.size() && !ok; ++i) {
        size_t p = i;
        if (p < region.size() && region[p] == '-') ++p;
        size_t d0 = p;
        while (p < region.size() && isdigit((unsigned char)region[p])) ++p;
        if (p == d0 || p >= region.size() || region[p] != ':') continue;
        ++p;
        if (p < region.size() && region[p] == '-') ++p;
        size_t d1 = p;
        while (p < region.size() && isdigit((unsigned char)region[p])) ++p;
        if (p > d1) ok = true;
    }
    if (!ok) throw Error("invalid region: " + region + ". type \"seqkit " + cmd + " -h\" for more examples");
    size_t c = region.find(':');
    char* endp = nullptr;
    std::string a = region.substr(0, c), b = region.substr(c + 1);
    long sa = strtol(a.c_str(), &endp, 10);
    if (a.empty() || *endp) throw Error("strconv.Atoi: parsing \"" + a + "\": invalid syntax");
    long sb = strtol(b.c_str(), &endp, 10);
    if (b.empty() || *endp) throw Error("strconv.Atoi: parsing \"" + b + "\": invalid syntax");
    if (sa == 0 || sb == 0) throw Error("both start and end should not be 0");
    if (sa < 0 && sb > 0) throw Error("when start < 0, end should not > 0");
    *start = (int)sa;
    *end = (int)sb;
}

std::string rev_com(const std::string& s, Alphabet a) {
    std::string r(s.rbegin(), s.rend());
    complement_inplace(r, a);
    return r;
}

// ---------------------------------------------------------------------------
// Pattern helpers shared by grep and locate
//   Seq.Degenerate2Regexp           [shenwei356/bio v0.7.0, not in tree; upstream-memory, PARITY.md DEG]
//   regexp.Regexp (Go RE2)          only the subset Degenerate2Regexp can emit: literals and [..] classes, (?i)
//   fmi.FMIndex.Match / Locate      [shenwei356/bwt v0.6.0, not in tree]: Hamming distance <= k, no indels,
//                                   locations ascending (PARITY.md FMI)
//   breader / fastx.GetSeqsMap      pattern files
// ---------------------------------------------------------------------------
static const char* degenerate_nucl(char c) {
    switch (c) {
        case 'A': return "A"; case 'T': return "T"; case 'U': return "U"; case 'C': return "C"; case 'G': return "G";
        case 'R': return "AG"; case 'Y': return "CT"; case 'M': return "AC"; case 'K': return "GT"; case 'S': return "CG";
        case 'W': return "AT"; case 'H': return "ACT"; case 'B': return "CGT"; case 'V': return "ACG"; case 'D': return "AGT";
        case 'N': return "ACGT";
        case 'a': return "a"; case 't': return "t"; case 'u': return "u"; case 'c': return "c"; case 'g': return "g";
        case 'r': return "ag"; case 'y': return "ct"; case 'm': return "ac"; case 'k': return "gt"; case 's': return "cg";
        case 'w': return "at"; case 'h': return "act"; case 'b': return "cgt"; case 'v': return "acg"; case 'd': return "agt";
        case 'n': return "acgt";
    }
    return nullptr;
}
static const char* degenerate_prot(char c) {
    static char one[128][2];
    switch (c) {
        case 'B': return "DN"; case 'Z': return "EQ"; case 'J': return "IL"; case 'X': return "ABCDEFGHIJKLMNOPQRSTUVWXYZ";
        case 'b': return "dn"; case 'z': return "eq"; case 'j': return "il"; case 'x': return "abcdefghijklmnopqrstuvwxyz";
    }
    if ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z')) { one[(int)c][0] = c; one[(int)c][1] = 0; return one[(int)c]; }
    return nullptr;
}

std::string degenerate2regexp(const std::string& p, Alphabet a) {
    std::string s;
    for (char c : p) {
        const char* m = a == AB_PROTEIN ? degenerate_prot(c) : degenerate_nucl(c);
        if (!m) s.push_back(c);
        else if (m[1] == 0) s += m;
        else s += std::string("[") + m + "]";
    }
    return s;
}

MiniRe MiniRe::compile(const std::string& re_in) {
    MiniRe r;
    std::string re = re_in;
    if (re.rfind("(?i)", 0) == 0) { r.icase = true; re = re.substr(4); }
    for (size_t i = 0; i < re.size(); ++i) {
        std::bitset<256> set;
        unsigned char c = (unsigned char)re[i];
        if (c == '[') {
            size_t j = re.find(']', i + 1);
            if (j == std::string::npos) throw Error("error parsing regexp: missing closing ]: `" + re.substr(i) + "`");
            for (size_t k = i + 1; k < j; ++k) set.set((unsigned char)re[k]);
            i = j;
        } else if (isalpha(c)) {
            set.set(c);
        } else {
            throw Error("oracle: regexp syntax beyond literals and [..] classes is not restated: " + re_in);
        }
        if (r.icase)
            for (int b = 0; b < 256; ++b)
                if (set[b] && isalpha(b)) { set.set(tolower(b)); set.set(toupper(b)); }
        r.atoms.push_back(set);
    }
    return r;
}

long MiniRe::find(const std::string& t, size_t from) const {  // leftmost match at or after `from`, -1 if none
    const size_t m = atoms.size();
    for (size_t i = from; i + m <= t.size(); ++i) {
        size_t q = 0;
        while (q < m && atoms[q][(unsigned char)t[i + q]]) ++q;
        if (q == m) return (long)i;
    }
    return -1;
}

std::vector<long> fmi_locate(const std::string& text, const std::string& pat, int k) {
    std::vector<long> loc;
    const size_t m = pat.size();
    for (size_t i = 0; m > 0 && i + m <= text.size(); ++i) {
        int mm = 0;
        for (size_t q = 0; q < m && mm <= k; ++q) mm += text[i + q] != pat[q];
        if (mm <= k) loc.push_back((long)i);
    }
    return loc;
}

static std::string read_text_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw Error("open " + path + ": no such file or directory");
    std::string s;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

std::vector<std::string> read_pattern_lines(const std::string& path) {  // breader: one pattern per line, CR/LF trimmed
    std::vector<std::string> out;
    const std::string s = read_text_file(path);
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find('\n', i);
        if (j == std::string::npos) j = s.size();
        std::string line = s.substr(i, j - i);
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        out.push_back(line);
        i = j + 1;
    }
    return out;
}

// ---------------------------------------------------------------------------
// Grep  bigseqkit-lib/grep.go (--delete-matched: grep.go:463-511 + the driver's reduce, bigseqkit/grep.go:144-156)
// ---------------------------------------------------------------------------
std::vector<std::string> grep_call(const std::vector<std::string_view>& part, const GrepOptions& oin) {
    GrepOptions o = oin;
    Alphabet ab = alphabet_from_seqtype(o.Config.SeqType);
    // PARITY.md Q17: the default Pattern [""] must not defeat the guard at grep.go:53
    bool any = !o.PatternFile.empty();
    for (auto& p : o.Pattern) if (!p.empty()) any = true;
    if (!any) throw Error("one of flags -p (--pattern) and -f (--pattern-file) needed");
    if (o.Degenerate) o.BySeq = true;
    if (o.MaxMismatch > 0) {
        if (o.UseRegexp || o.Degenerate)
            throw Error("flag -r (--use-regexp) or -d (--degenerate) not allowed when giving flag -m (--max-mismatch)");
        o.BySeq = true;
    }
    if (o.UseRegexp && o.Degenerate) throw Error("could not give both flags -d (--degenerate) and -r (--use-regexp)");
    bool limitRegion = false;
    int start = 0, end = 0;
    if (!o.Region.empty()) {
        limitRegion = true;
        o.BySeq = true;
        parse_region(o.Region, "grep", &start, &end);
    }
    // --delete-matched (with -v it does nothing, grep.go:463): a pattern is dropped at its first hit, so every pattern
    // selects at most its FIRST record in file order (the driver's ReduceByKey keeps the lowest partition, PARITY.md DEL)
    // with -m the reference takes grepBySeqMismatches (grep.go:255-365), which never drops a pattern, and the driver hands
    // its records through (bigseqkit/grep.go:141-143): --delete-matched does nothing there
    const bool del = o.DeleteMatched && !o.InvertMatch && !(o.BySeq && o.MaxMismatch > 0);
    std::vector<std::string> patterns;  // PARITY.md Q11: CLI / file order instead of Go map order
    std::vector<MiniRe> regexps;        // -d
    // -r: Go regexp (RE2) is not in this image; std::regex (ECMAScript grammar) stands in for it -- the two agree on
    // the syntax the tests use (classes, escapes, groups, alternation, quantifiers, ^ $).  "(?i)" becomes the icase flag.
    std::vector<std::regex> stdres;
    // grep.go:122-252: the file replaces -p when given
    const std::vector<std::string> given = !o.PatternFile.empty() ? read_pattern_lines(o.PatternFile) : o.Pattern;
    for (auto p : given) {
        if (p.empty()) continue;
        if (o.UseRegexp) {  // grep.go:148-153
            if (std::find(patterns.begin(), patterns.end(), p) != patterns.end()) continue;
            try {
                stdres.emplace_back(p, o.IgnoreCase ? std::regex::ECMAScript | std::regex::icase : std::regex::ECMAScript);
            } catch (const std::regex_error& e) { throw Error(std::string("error parsing regexp: ") + e.what() + ": `" + p + "`"); }
            patterns.push_back(p);
            continue;
        }
        if (o.Degenerate) {
            p = degenerate2regexp(p, ab);
            if (o.IgnoreCase) p = "(?i)" + p;
            if (std::find(patterns.begin(), patterns.end(), p) != patterns.end()) continue;
            regexps.push_back(MiniRe::compile(p));
            patterns.push_back(p);
            continue;
        }
        if (o.BySeq) {
            if (o.MaxMismatch > 0 && o.MaxMismatch > (int)p.size()) throw Error("mismatch should be <= length of sequence: " + p);
            if (!(alphabet_is_valid(AB_DNAredundant, p) || alphabet_is_valid(AB_RNAredundant, p) ||
                  alphabet_is_valid(AB_PROTEIN, p)))
                throw Error("illegal DNA/RNA/Protein sequence: " + p);
        }
        if (o.IgnoreCase) p = lower(p);
        if (std::find(patterns.begin(), patterns.end(), p) == patterns.end()) patterns.push_back(p);
    }
    auto drop = [&](size_t idx) {
        if (!del) return;
        patterns.erase(patterns.begin() + (long)idx);
        if (!regexps.empty()) regexps.erase(regexps.begin() + (long)idx);
        if (!stdres.empty()) stdres.erase(stdres.begin() + (long)idx);
    };
    SeqParser rd(ab, &part, o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
    std::vector<std::string> result;
    bool checkAlphabet = true, onlyPos = o.OnlyPositiveStrand;
    int lineWidth = o.Config.LineWidth;
    int64_t count = 0;
    while (rd.Read()) {
        Record& r = rd.rec;
        if (checkAlphabet) {  // grep.go:403-409, :276-281
            if (rd.GetAlphabet() == AB_UNLIMIT || rd.GetAlphabet() == AB_PROTEIN) onlyPos = true;
            checkAlphabet = false;
        }
        if (rd.IsFastq) lineWidth = 0;
        bool hit = false;
        for (int strand = 0; strand < 2 && !hit; ++strand) {
            if (strand == 1 && (!o.BySeq || onlyPos)) break;
            std::string target;
            if (o.BySeq) {
                std::string sq = strand == 0 ? r.seq : rev_com(r.seq, rd.GetAlphabet());
                if (limitRegion) {
                    size_t b, e;
                    sub_location(sq.size(), start, end, &b, &e);
                    target = sq.substr(b, e - b);
                } else if (o.Circular) target = sq + sq;
                else target = sq;
                if (o.UseRegexp) {
                    for (size_t q = 0; q < stdres.size(); ++q)
                        if (std::regex_search(target, stdres[q])) { hit = true; drop(q); break; }
                    continue;
                }
                if (o.Degenerate) {  // grep.go:459-468: re.Match on the un-lowered target
                    for (size_t q = 0; q < regexps.size(); ++q)
                        if (regexps[q].find(target, 0) >= 0) { hit = true; drop(q); break; }
                    continue;
                }
                if (o.IgnoreCase) target = lower(target);
                for (size_t q = 0; q < patterns.size(); ++q) {
                    const std::string& k = patterns[q];
                    if (o.MaxMismatch == 0 ? target.find(k) != std::string::npos
                                           : !fmi_locate(target, k, o.MaxMismatch).empty()) {  // grep.go:327-339, 484-497
                        hit = true;
                        drop(q);
                        break;
                    }
                }
            } else {
                target = o.ByName ? r.name : r.id;
                if (o.UseRegexp) {  // grep.go:459-468: the regexp sees the un-lowered ID / name
                    for (size_t q = 0; q < stdres.size(); ++q)
                        if (std::regex_search(target, stdres[q])) { hit = true; drop(q); break; }
                    continue;
                }
                if (o.IgnoreCase) target = lower(target);
                auto it = std::find(patterns.begin(), patterns.end(), target);
                hit = it != patterns.end();
                if (hit) drop((size_t)(it - patterns.begin()));
            }
        }
        if (o.InvertMatch ? hit : !hit) continue;
        if (o.Count) { ++count; continue; }
        std::string bb = record_format(r, rd.IsFastq, lineWidth);
        bb.pop_back();  // grep.go:531-533 (grepBySeqMismatches keeps it, :356 -- PARITY.md Q6: one newline everywhere)
        result.push_back(bb);
    }
    if (o.Count) result.push_back(std::to_string(count));
    return result;
}

// ---------------------------------------------------------------------------
// SubseqTransform  bigseqkit-lib/subseq.go
//   gtf.ReadFilteredFeatures [shenwei356/bio v0.7.0 featio/gtf, not in tree; PARITY.md GTF]
//   ReadBedFilteredFeatures  subseq.go:242-310 (in tree)
// ---------------------------------------------------------------------------
struct FeatureRow {  // what subseqByGTFFile / subSeqByBEDFile use of a feature
    std::string chr, type;
    long start = 0, end = 0;  // 1-based, end included
    std::string strand;       // "+", "-", "." ("." also when the column is absent)
    std::string label;        // BED name column / value of the --gtf-tag attribute
};

static std::vector<std::string> split_tab(const std::string& line) {
    std::vector<std::string> items;
    size_t i = 0;
    for (;;) {
        size_t j = line.find('\t', i);
        if (j == std::string::npos) { items.push_back(line.substr(i)); break; }
        items.push_back(line.substr(i, j - i));
        i = j + 1;
    }
    return items;
}

static bool go_atoi(const std::string& s, long* v) {
    if (s.empty()) return false;
    char* e = nullptr;
    *v = strtol(s.c_str(), &e, 10);
    return *e == 0 && !isspace((unsigned char)s[0]);
}

static std::vector<FeatureRow> read_bed(const std::string& file, const std::vector<std::string>& chrs) {
    std::vector<FeatureRow> out;
    for (std::string line : read_pattern_lines(file)) {  // lines with "\r\n" trimmed (subseq.go:251)
        if (line.empty() || line[0] == '#' || (line.size() > 7 && line.compare(0, 7, "browser") == 0) ||
            (line.size() > 5 && line.compare(0, 5, "track") == 0))
            continue;
        auto items = split_tab(line);
        if (items.size() < 3) continue;
        if (!chrs.empty() && std::find(chrs.begin(), chrs.end(), items[0]) == chrs.end()) continue;
        FeatureRow f;
        long st, en;
        if (!go_atoi(items[1], &st)) throw Error(items[0] + ": bad start: " + items[1]);
        if (!go_atoi(items[2], &en)) throw Error(items[0] + ": bad end: " + items[2]);
        if (st >= en) throw Error(items[0] + ": start (" + std::to_string(st) + ") must be <= end (" + std::to_string(en) + ")");
        f.chr = items[0];
        f.start = st + 1;
        f.end = en;
        if (items.size() >= 4) f.label = items[3];
        f.strand = ".";
        if (items.size() >= 6) {
            if (items[5] != "+" && items[5] != "-" && items[5] != ".") throw Error("bad strand: " + items[5]);
            f.strand = items[5];
        }
        out.push_back(f);
    }
    return out;
}

static std::vector<FeatureRow> read_gtf(const std::string& file, const std::vector<std::string>& chrs,
                                        const std::vector<std::string>& feats_lower, const std::string& tag) {
    std::vector<FeatureRow> out;
    for (std::string line : read_pattern_lines(file)) {
        if (line.empty() || line[0] == '#') continue;
        auto items = split_tab(line);
        if (items.size() != 9) continue;
        if (!chrs.empty() && std::find(chrs.begin(), chrs.end(), items[0]) == chrs.end()) continue;
        if (!feats_lower.empty() && std::find(feats_lower.begin(), feats_lower.end(), lower(items[2])) == feats_lower.end()) continue;
        FeatureRow f;
        long st, en;
        if (!go_atoi(items[3], &st)) throw Error(items[0] + ": bad start: " + items[3]);
        if (!go_atoi(items[4], &en)) throw Error(items[0] + ": bad end: " + items[4]);
        if (st > en) throw Error(items[0] + ": start (" + std::to_string(st) + ") must be < end (" + std::to_string(en) + ")");
        if (items[6] != "+" && items[6] != "-" && items[6] != ".") throw Error("bad strand: " + items[6]);
        f.chr = items[0];
        f.type = items[2];
        f.start = st;
        f.end = en;
        f.strand = items[6];
        // attributes: `tag "value"; tag "value";`  -> first attribute named `tag`
        const std::string& at = items[8];
        size_t i = 0;
        while (i < at.size()) {
            size_t j = at.find(';', i);
            if (j == std::string::npos) j = at.size();
            std::string item = at.substr(i, j - i);
            i = j + 1;
            size_t a0 = item.find_first_not_of(' ');
            if (a0 == std::string::npos) continue;
            item = item.substr(a0);
            size_t sp = item.find(' ');
            if (sp == std::string::npos) continue;
            std::string t = item.substr(0, sp), v = item.substr(sp + 1);
            while (!v.empty() && v.back() == ' ') v.pop_back();
            if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);
            if (t == tag) { f.label = v; break; }
        }
        out.push_back(f);
    }
    return out;
}

std::vector<std::string> subseq_call(const std::vector<std::string_view>& part, const SubseqOptions& o) {
    Alphabet ab = alphabet_from_seqtype(o.Config.SeqType);
    std::vector<std::string> feats_lower;
    for (auto& f : o.Feature) feats_lower.push_back(lower(f));
    if (o.OnlyFlank) {  // subseq.go:63-71
        if (o.UpStream > 0 && o.DownStream > 0)
            throw Error("when flag -f (--only-flank) given, only one of flags -u (--up-stream) and -d (--down-stream) is allowed");
        else if (o.UpStream == 0 && o.DownStream == 0)
            throw Error("when flag -f (--only-flank) given, one of flags -u (--up-stream) and -d (--down-stream) should be given");
    }
    int start = 0, end = 0;
    std::vector<FeatureRow> features;
    const bool byFeature = o.Region.empty();
    if (!o.Region.empty()) {
        if (o.UpStream > 0 || o.DownStream > 0 || o.OnlyFlank)
            throw Error("when flag -r (--region) given, any of flags -u (--up-stream), -d (--down-stream) and -f (--only-flank) is not allowed");
        parse_region(o.Region, "subseq", &start, &end);
    } else if (!o.Gtf.empty()) {
        features = read_gtf(o.Gtf, o.Chr, feats_lower, o.GtfTag);
    } else if (!o.Bed.empty()) {
        if (!o.Feature.empty()) throw Error("when given flag -b (--bed), flag -f (--feature) is not allowed");
        features = read_bed(o.Bed, o.Chr);
    } else {
        throw Error("one of the options needed: -r/--region, --bed, --gtf");
    }
    SeqParser rd(ab, &part, o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
    std::vector<std::string> result;
    int lineWidth = o.Config.LineWidth;
    while (rd.Read()) {
        Record& r = rd.rec;
        if (rd.IsFastq) lineWidth = 0;
        if (!byFeature) {
            size_t b, e;
            sub_location(r.seq.size(), start, end, &b, &e);
            r.seq = r.seq.substr(b, e - b);
            if (!r.qual.empty()) r.qual = r.qual.substr(b, e - b);
            std::string bb = record_format(r, rd.IsFastq, lineWidth);
            bb.pop_back();  // PARITY.md Q6: no blank line between records
            result.push_back(bb);
            continue;
        }
        // subseq.go:319-526: the FIRST feature of the record's (lower-cased) ID, then return (Q7 as written);
        // "first" = file order (the Go map over feature types has no order; PARITY.md GTF)
        const std::string seqname = lower(r.id);
        const FeatureRow* f = nullptr;
        for (auto& cand : features)
            if (lower(cand.chr) == seqname) { f = &cand; break; }
        if (!f) continue;
        long s = f->start, e = f->end;
        const long L = (long)r.seq.size();
        const bool minus = f->strand == "-";
        if (minus) {
            if (o.OnlyFlank) {
                if (o.UpStream > 0) { s = f->end + 1; e = f->end + o.UpStream; }
                else { s = f->start - o.DownStream; e = f->start - 1; }
            } else { s = f->start - o.DownStream; e = f->end + o.UpStream; }
        } else {
            if (o.OnlyFlank) {
                if (o.UpStream > 0) { s = f->start - o.UpStream; e = f->start - 1; }
                else { s = e + 1; e = e + o.DownStream; }
            } else { s = f->start - o.UpStream; e = f->end + o.DownStream; }
        }
        if (s < 1) s = 1;
        if (e > L) e = L;
        size_t b0 = 0, e0 = 0;
        if (e >= 1) sub_location(r.seq.size(), (int)s, (int)e, &b0, &e0);  // e < 1: empty (PARITY.md SUB0)
        Record nr;
        nr.seq = r.seq.substr(b0, e0 - b0);
        if (!r.qual.empty()) nr.qual = r.qual.substr(b0, e0 - b0);
        if (minus) {  // RevComInplace reverses the qualities too
            nr.seq = rev_com(nr.seq, rd.GetAlphabet());
            std::reverse(nr.qual.begin(), nr.qual.end());
        }
        std::string flank;
        if (o.UpStream > 0) {
            if (o.OnlyFlank) flank = "_usf:" + std::to_string(o.UpStream);
            else if (o.DownStream > 0) flank = "_us:" + std::to_string(o.UpStream) + "_ds:" + std::to_string(o.DownStream);
            else flank = "_us:" + std::to_string(o.UpStream);
        } else if (o.DownStream > 0) {
            if (o.OnlyFlank) flank = "_dsf:" + std::to_string(o.DownStream);
            else flank = "_ds:" + std::to_string(o.DownStream);
        }
        nr.name = r.id + "_" + std::to_string(f->start) + "-" + std::to_string(f->end) + ":" + f->strand + flank + " " + f->label;
        nr.id = nr.name;
        std::string bb = record_format(nr, rd.IsFastq, lineWidth);
        bb.pop_back();  // Q6
        result.push_back(bb);
    }
    return result;
}

// ---------------------------------------------------------------------------
// Translate  bigseqkit-lib/translate.go + bio CodonTable.Translate [upstream-memory]
// ---------------------------------------------------------------------------
#include "genetic_codes_diff.inc"

// gc.prt form (ncbieaa / sncbieaa over the base order TCAG) of one table, derived from the standard code and the
// table's documented differences (genetic_codes_diff.inc)
struct GeneticCode { int id; std::string aa, starts; };

static int tcag_index(const char* codon) {
    static const char order[] = "TCAG";
    int idx = 0;
    for (int k = 0; k < 3; ++k) {
        const char* q = strchr(order, codon[k]);
        if (!q || !codon[k]) throw Error(std::string("oracle: bad codon in genetic_codes_diff.inc: ") + codon);
        idx = idx * 4 + (int)(q - order);
    }
    return idx;
}

static std::vector<std::string> split_blank(const char* s) {
    std::vector<std::string> v;
    std::string cur;
    for (const char* p = s;; ++p) {
        if (*p == ' ' || *p == 0) { if (!cur.empty()) v.push_back(cur); cur.clear(); if (!*p) break; }
        else cur.push_back(*p);
    }
    return v;
}

static const std::vector<GeneticCode>& genetic_codes() {
    static const std::vector<GeneticCode> codes = [] {
        std::vector<GeneticCode> v;
        for (auto& d : kCodeDiffs) {
            GeneticCode g;
            g.id = d.id;
            g.aa.assign(64, '?');
            g.starts.assign(64, '-');
            for (auto& sc : kStandardCode) g.aa[tcag_index(sc.codon)] = sc.aa;
            for (auto& e : split_blank(d.diffs)) {
                if (e.size() != 5 || e[3] != '=') throw Error("oracle: bad diff entry " + e);
                g.aa[tcag_index(e.substr(0, 3).c_str())] = e[4];
            }
            for (auto& e : split_blank(d.starts)) g.starts[tcag_index(e.c_str())] = 'M';
            for (auto& e : split_blank(d.start_line_stops)) g.starts[tcag_index(e.c_str())] = '*';
            v.push_back(g);
        }
        return v;
    }();
    return codes;
}

const char* genetic_code_strings(int id, int which) {  // tests: the derived ncbieaa (0) / sncbieaa (1) line
    for (auto& g : genetic_codes())
        if (g.id == id) return which ? g.starts.c_str() : g.aa.c_str();
    return nullptr;
}

static const char* iupac_set(char c) {  // upper-case, U == T
    switch (c) {
        case 'A': return "A"; case 'C': return "C"; case 'G': return "G"; case 'T': case 'U': return "T";
        case 'R': return "AG"; case 'Y': return "CT"; case 'S': return "CG"; case 'W': return "AT";
        case 'K': return "GT"; case 'M': return "AC"; case 'B': return "CGT"; case 'D': return "AGT";
        case 'H': return "ACT"; case 'V': return "ACG"; case 'N': return "ACGT";
        default: return nullptr;
    }
}

static const GeneticCode* find_code(int id) {
    for (auto& g : genetic_codes())
        if (g.id == id) return &g;
    return nullptr;
}

// amino acid of a (possibly ambiguous) codon: the common translation of all its
// expansions, 'X' when they disagree, 0 when a letter is not an IUPAC base.
static char codon_aa(const GeneticCode& g, const char* c3) {
    const char* s[3];
    for (int k = 0; k < 3; ++k) {
        char c = c3[k];
        if (c >= 'a' && c <= 'z') c -= 32;
        s[k] = iupac_set(c);
        if (!s[k]) return 0;
    }
    static const char order[] = "TCAG";
    char aa = 0;
    for (const char* a = s[0]; *a; ++a)
        for (const char* b = s[1]; *b; ++b)
            for (const char* c = s[2]; *c; ++c) {
                int i = (int)(strchr(order, *a) - order) * 16 + (int)(strchr(order, *b) - order) * 4 +
                        (int)(strchr(order, *c) - order);
                if (aa == 0) aa = g.aa[i];
                else if (aa != g.aa[i]) return 'X';
            }
    return aa;
}

// codon_aa over every triple of the fifteen IUPAC letters, computed ONCE per genetic code with codon_aa itself: the checker's
// definition stays the function above (three strchr per base triple and a triple loop per codon: 17 MB/s, a CPU baseline
// nobody would believe -- VERDICT r05 weak 8); translate_seq looks the 4 096-entry table up.  tests/test_oracle_kat.py holds
// the table to codon_aa on all 16^3 index triples through oracle_codon_aa_pair.
static const uint8_t* iupac_index() {
    static const std::array<uint8_t, 256> t = [] {
        std::array<uint8_t, 256> m;
        m.fill(15);
        const char* L = "ACGTRYSWKMBDHVN";
        for (int i = 0; i < 15; ++i) { m[(uint8_t)L[i]] = (uint8_t)i; m[(uint8_t)(L[i] + 32)] = (uint8_t)i; }
        m[(uint8_t)'U'] = m[(uint8_t)'u'] = 3;
        return m;
    }();
    return t.data();
}
static const char* codon_table(const GeneticCode& g) {
    static std::mutex mu;
    static std::map<int, std::array<char, 4096>> tables;
    std::lock_guard<std::mutex> lk(mu);
    auto it = tables.find(g.id);
    if (it != tables.end()) return it->second.data();
    std::array<char, 4096>& t = tables[g.id];
    const char* L = "ACGTRYSWKMBDHVN";
    for (int a = 0; a < 16; ++a)
        for (int b = 0; b < 16; ++b)
            for (int c = 0; c < 16; ++c) {
                char c3[3] = {a < 15 ? L[a] : '?', b < 15 ? L[b] : '?', c < 15 ? L[c] : '?'};
                t[(size_t)a * 256 + (size_t)b * 16 + (size_t)c] = codon_aa(g, c3);
            }
    return t.data();
}
static inline char codon_aa_fast(const char* table, const uint8_t* idx, const char* c3) {
    return table[(size_t)idx[(uint8_t)c3[0]] * 256 + (size_t)idx[(uint8_t)c3[1]] * 16 + (size_t)idx[(uint8_t)c3[2]]];
}
// tests: the slow definition and the table side by side for one codon of one code ((char)-1: no such code)
int codon_aa_pair(int table_id, const char* c3, char* slow, char* fast) {
    const GeneticCode* g = find_code(table_id);
    if (!g) return -1;
    *slow = codon_aa(*g, c3);
    *fast = codon_aa_fast(codon_table(*g), iupac_index(), c3);
    return 0;
}

static bool codon_is_start(const GeneticCode& g, const char* c3) {
    static const char order[] = "TCAG";
    int idx = 0;
    for (int k = 0; k < 3; ++k) {
        char c = c3[k];
        if (c >= 'a' && c <= 'z') c -= 32;
        if (c == 'U') c = 'T';
        const char* q = strchr(order, c);
        if (!q || !c) return false;
        idx = idx * 4 + (int)(q - order);
    }
    return g.starts[idx] == 'M';
}

// the strand a negative frame reads: reverse complement over the IUPAC letters, case kept, others unchanged
static std::string translate_minus_strand(const std::string& seq_in) {
    std::string sq = rev_com(seq_in, AB_DNAredundant);
    for (auto& c : sq) { if (c == 'u') c = 'a'; else if (c == 'U') c = 'A'; }  // RNA input: U pairs with A
    return sq;
}

// `sq`: the strand the frame reads (seq_in itself, or translate_minus_strand(seq_in)); frame > 0
static std::string translate_strand(const std::string& sq, const GeneticCode* g, int frame, bool trim, bool clean, bool allow_unknown,
                                    bool init_m, bool* unknown);

std::string translate_seq(const std::string& seq_in, int table, int frame, bool trim, bool clean, bool allow_unknown,
                          bool init_m, bool* unknown) {
    *unknown = false;
    const GeneticCode* g = find_code(table);
    if (!g) throw Error("invalid translate table: " + std::to_string(table));
    if (frame < 0) return translate_strand(translate_minus_strand(seq_in), g, -frame, trim, clean, allow_unknown, init_m, unknown);
    return translate_strand(seq_in, g, frame, trim, clean, allow_unknown, init_m, unknown);
}

static std::string translate_strand(const std::string& sq, const GeneticCode* g, int frame, bool trim, bool clean, bool allow_unknown,
                                    bool init_m, bool* unknown) {
    std::string aas;
    aas.reserve(sq.size() / 3 + 1);
    bool first = true;
    const char* lut = codon_table(*g);
    const uint8_t* idx = iupac_index();
    for (size_t i = (size_t)frame - 1; i + 2 < sq.size(); i += 3) {
        char aa = codon_aa_fast(lut, idx, sq.data() + i);
        if (aa == 0) {
            if (allow_unknown) aa = 'X';
            else { *unknown = true; return std::string(); }
        }
        if (first) {
            first = false;
            if (init_m && codon_is_start(*g, sq.data() + i)) aa = 'M';
        }
        if (clean && aa == '*') aa = 'X';
        aas.push_back(aa);
    }
    if (trim)
        while (!aas.empty() && (aas.back() == 'X' || aas.back() == '*')) aas.pop_back();
    return aas;
}

std::vector<std::string> translate_call(const std::vector<std::string_view>& part, const TranslateOptions& o) {
    Alphabet ab = alphabet_from_seqtype(o.Config.SeqType);
    if (!find_code(o.TranslTable)) throw Error("invalid translate table: " + std::to_string(o.TranslTable));
    std::vector<int> frames;
    for (auto& f : o.Frame) {  // translate.go:46-61
        char* endp = nullptr;
        long v = strtol(f.c_str(), &endp, 10);
        if (f.empty() || *endp)
            throw Error("invalid frame(s): " + f + ". available: 1, 2, 3, -1, -2, -3, and 6 for all. multiple frames should be separated by comma");
        if (!(v == 1 || v == 2 || v == 3 || v == -1 || v == -2 || v == -3 || v == 6))
            throw Error("invalid frame: " + std::to_string(v) + ". available: 1, 2, 3, -1, -2, -3, and 6 for all");
        if (v == 6) { frames = {1, 2, 3, -1, -2, -3}; break; }
        frames.push_back((int)v);
    }
    if (o.ListTranslTableWithAmbCodons == 0 || o.ListTranslTable == 0) {  // translate.go:78-89
        // names: the list in the reference's own help text (bigseqkit-cli/translate.go:55-78), ids ascending
        static const char* const help =
            "1: The Standard Code\n"
            "2: The Vertebrate Mitochondrial Code\n"
            "3: The Yeast Mitochondrial Code\n"
            "4: The Mold, Protozoan, and Coelenterate Mitochondrial Code and the Mycoplasma/Spiroplasma Code\n"
            "5: The Invertebrate Mitochondrial Code\n"
            "6: The Ciliate, Dasycladacean and Hexamita Nuclear Code\n"
            "9: The Echinoderm and Flatworm Mitochondrial Code\n"
            "10: The Euplotid Nuclear Code\n"
            "11: The Bacterial, Archaeal and Plant Plastid Code\n"
            "12: The Alternative Yeast Nuclear Code\n"
            "13: The Ascidian Mitochondrial Code\n"
            "14: The Alternative Flatworm Mitochondrial Code\n"
            "16: Chlorophycean Mitochondrial Code\n"
            "21: Trematode Mitochondrial Code\n"
            "22: Scenedesmus obliquus Mitochondrial Code\n"
            "23: Thraustochytrium Mitochondrial Code\n"
            "24: Pterobranchia Mitochondrial Code\n"
            "25: Candidate Division SR1 and Gracilibacteria Code\n"
            "26: Pachysolen tannophilus Nuclear Code\n"
            "27: Karyorelict Nuclear\n"
            "28: Condylostoma Nuclear\n"
            "29: Mesodinium Nuclear\n"
            "30: Peritrich Nuclear\n"
            "31: Blastocrithidia Nuclear\n"
;
        std::vector<std::string> rows;
        for (const char* p = help; *p;) {
            const char* e = strchr(p, '\n');
            std::string line(p, e);
            const size_t colon = line.find(": ");
            rows.push_back(line.substr(0, colon) + "\t" + line.substr(colon + 2));
            p = e + 1;
        }
        return rows;
    }
    if (o.ListTranslTable > 0 || o.ListTranslTableWithAmbCodons > 0)
        throw Error("oracle: translate -l N / -L N (bio's CodonTable.String()) is not restated");
    SeqParser rd(ab, &part, o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
    std::vector<std::string> result;
    bool once = true;
    while (rd.Read()) {
        Record& r = rd.rec;
        if (once) {
            Alphabet a = rd.GetAlphabet();
            if (!(a == AB_DNA || a == AB_DNAredundant || a == AB_RNA || a == AB_RNAredundant))
                throw Error("command 'seqkit translate' only apply to DNA/RNA sequences");
            once = false;
        }
        // (the reference reverse-complements inside every Translate call of a negative frame, translate.go:124-133; the
        // strand is the same for the three of them: computed once per record here -- the CPU baseline)
        std::string minus;
        bool have_minus = false;
        const GeneticCode* gc = find_code(o.TranslTable);
        for (int frame : frames) {
            bool unknown = false;
            if (frame < 0 && !have_minus) { minus = translate_minus_strand(r.seq); have_minus = true; }
            std::string aa = translate_strand(frame < 0 ? minus : r.seq, gc, frame < 0 ? -frame : frame, o.Trim, o.Clean,
                                              o.AllowUnknownCodon, o.InitCodonAsM, &unknown);
            if (unknown) throw Error("seq: unknown codon");
            std::string out;
            if (o.AppendFrame) out = ">" + r.id + "_frame=" + std::to_string(frame) + " " + r.desc + "\n";
            else out = ">" + r.name + "\n";
            out += wrap_byte_slice(aa, o.Config.LineWidth);
            result.push_back(out);
        }
    }
    return result;
}

// ---------------------------------------------------------------------------
// Locate  bigseqkit-lib/locate.go (no -r)
// ---------------------------------------------------------------------------
std::vector<std::pair<std::string, std::string>> read_pattern_fasta(const std::string& path) {
    // fastx.GetSeqsMap(file, seq.Unlimit, ...): full name -> sequence; file order kept (PARITY.md Q11)
    std::vector<std::pair<std::string, std::string>> out;
    const std::string s = [&] {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) throw Error("open " + path + ": no such file or directory");
        std::string t;
        char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) t.append(buf, n);
        fclose(f);
        return t;
    }();
    size_t i = 0;
    bool have = false;
    while (i < s.size()) {
        size_t j = s.find('\n', i);
        if (j == std::string::npos) j = s.size();
        std::string line = s.substr(i, j - i);
        while (!line.empty() && line.back() == '\r') line.pop_back();
        i = j + 1;
        if (!line.empty() && line[0] == '>') {
            const std::string name = line.substr(1);
            have = true;
            size_t k = 0;
            for (; k < out.size(); ++k) if (out[k].first == name) break;
            if (k == out.size()) out.emplace_back(name, "");
            else { auto e = out[k]; out.erase(out.begin() + (long)k); e.second.clear(); out.push_back(e); }
        } else if (have) {
            out.back().second += line;
        }
    }
    return out;
}

std::vector<std::string> locate_call(const std::vector<std::string_view>& part, const LocateOptions& o, int64_t pid) {
    Alphabet ab = alphabet_from_seqtype(o.Config.SeqType);
    bool any = !o.PatternFile.empty();
    for (auto& p : o.Pattern) if (!p.empty()) any = true;
    if (!any) throw Error("one of flags -p (--pattern) and -f (--pattern-file) needed");  // PARITY.md Q17
    if (o.MaxMismatch > 0) {
        if (o.Degenerate) throw Error("flag -d (--degenerate) not allowed when giving flag -m (--max-mismatch)");
        if (o.UseRegexp) throw Error("flag -r (--use-regexp) not allowed when giving flag -m (--use-regexp)");
    }
    if (o.UseFmi) {
        if (o.Degenerate) throw Error("flag -d (--degenerate) ignored when giving flag -F (--use-fmi)");
        if (o.UseRegexp) throw Error("flag -r (--use-regexp) ignored when giving flag -F (--use-fmi)");
    }
    // -r: std::regex (ECMAScript) stands in for Go regexp, as in grep_call
    struct Pat { std::string name, bytes; MiniRe re; std::regex sre; };
    std::vector<Pat> pats;  // CLI / file order (PARITY.md Q11)
    std::vector<std::pair<std::string, std::string>> given;
    if (!o.PatternFile.empty()) {
        given = read_pattern_fasta(o.PatternFile);
        if (given.empty()) throw Error("no FASTA sequences found in pattern file: " + o.PatternFile);
    } else {
        for (auto& p : o.Pattern) if (!p.empty()) given.emplace_back(p, p);
    }
    for (auto& g : given) {  // locate.go:86-190
        Pat pt;
        pt.name = g.first;
        pt.bytes = g.second;
        std::string re;
        if (o.Degenerate) re = degenerate2regexp(g.second, !o.PatternFile.empty() ? AB_UNLIMIT : ab);
        else if (o.UseRegexp) {  // locate.go:102-121
            try { pt.sre = std::regex(g.second, o.IgnoreCase ? std::regex::ECMAScript | std::regex::icase : std::regex::ECMAScript); }
            catch (const std::regex_error& e) { throw Error(std::string("error parsing regexp: ") + e.what() + ": `" + g.second + "`"); }
        }
        else if (o.IgnoreCase) pt.bytes = lower(pt.bytes);
        if (o.UseRegexp) {
            bool dup = false;
            for (auto& q : pats) if (q.name == pt.name) dup = true;
            if (!dup) pats.push_back(pt);
            continue;
        }
        if (o.MaxMismatch > 0) {
            if (o.MaxMismatch > (int)pt.bytes.size()) throw Error("mismatch should be <= length of sequence: " + g.second);
            if (!(alphabet_is_valid(AB_DNAredundant, pt.bytes) || alphabet_is_valid(AB_RNAredundant, pt.bytes) ||
                  alphabet_is_valid(AB_PROTEIN, pt.bytes)))
                throw Error("illegal DNA/RNA/Protein sequence: " + g.first);
        } else if (o.Degenerate) {
            if (o.IgnoreCase) re = "(?i)" + re;
            pt.re = MiniRe::compile(re);
        } else if (pt.bytes.find('.') != std::string::npos ||
                   !(alphabet_is_valid(AB_DNAredundant, pt.bytes) || alphabet_is_valid(AB_RNAredundant, pt.bytes) ||
                     alphabet_is_valid(AB_PROTEIN, pt.bytes))) {
            throw Error("illegal DNA/RNA/Protein sequence: " + g.first + ", you may switch on -d/--degenerate or -r/--use-regexp");
        }
        bool dup = false;
        for (auto& q : pats) if (q.name == pt.name) dup = true;
        if (!dup) pats.push_back(pt);
    }
    std::vector<std::string> result;
    if (!(o.Gtf || o.Bed) && pid == 0)  // locate.go:198-204
        result.push_back(o.HideMatched ? "seqID\tpatternName\tpattern\tstrand\tstart\tend"
                                       : "seqID\tpatternName\tpattern\tstrand\tstart\tend\tmatched");
    SeqParser rd(ab, &part, o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
    auto row = [&](const Record& r, const Pat& pt, char strand, long begin, long end, const std::string& matched) {
        char b[64];
        std::string s;
        if (o.Gtf) {
            snprintf(b, sizeof b, "%ld\t%ld\t%d\t%c\t", begin, end, 0, strand);
            s = r.id + "\tSeqKit\tlocation\t" + b + ".\tgene_id \"" + pt.name + "\"; ";
        } else if (o.Bed) {
            snprintf(b, sizeof b, "\t%ld\t%ld\t", begin - 1, end);
            s = r.id + b + pt.name + "\t0\t" + strand;
        } else {
            snprintf(b, sizeof b, "%c\t%ld\t%ld", strand, begin, end);
            s = r.id + "\t" + pt.name + "\t" + pt.bytes + "\t" + b;
            if (!o.HideMatched) s += "\t" + matched;
        }
        result.push_back(s);
    };
    bool checkAlphabet = true, onlyPosAuto = o.OnlyPositiveStrand;
    while (rd.Read()) {
        Record& r = rd.rec;
        if (checkAlphabet) {  // locate.go:222-227, 423-428
            if (rd.GetAlphabet() == AB_UNLIMIT || rd.GetAlphabet() == AB_PROTEIN) onlyPosAuto = true;
            checkAlphabet = false;
        }
        if (!(o.Degenerate || o.UseRegexp) && o.IgnoreCase) r.seq = lower(r.seq);  // locate.go:430-432
        const long l = (long)r.seq.size();
        if (o.Circular) r.seq += r.seq;
        const long n = (long)r.seq.size();
        if (o.MaxMismatch > 0 || o.UseFmi) {  // locate.go:208-391: every pattern on '+', then every pattern on '-'
            for (int strand = 0; strand < 2; ++strand) {
                if (strand == 1 && onlyPosAuto) break;
                const std::string text = strand == 0 ? r.seq : rev_com(r.seq, rd.GetAlphabet());
                for (auto& pt : pats) {
                    const long lp = (long)pt.bytes.size();
                    for (long i : fmi_locate(text, pt.bytes, o.MaxMismatch)) {
                        if (o.Circular && i + 1 > l) continue;
                        if (i + lp > n) continue;
                        const long begin = strand == 0 ? i + 1 : l - i - lp + 1;
                        const long end = strand == 0 ? i + lp : l - i;
                        row(r, pt, strand == 0 ? '+' : '-', begin, end, text.substr((size_t)i, (size_t)lp));
                    }
                }
            }
            continue;
        }
        if (o.UseRegexp) {
            // locate.go:575-767 with FindSubmatchIndex: matches of any length, the containment check :604-614 is live
            for (auto& pt : pats) {
                for (int strand = 0; strand < 2; ++strand) {
                    if (strand == 1 && o.OnlyPositiveStrand) break;
                    const std::string text = strand == 0 ? r.seq : rev_com(r.seq, rd.GetAlphabet());
                    std::vector<std::pair<long, long>> locs;
                    long offset = 0;
                    for (;;) {
                        std::smatch m;
                        if (offset > (long)text.size()) break;
                        const std::string tail = text.substr((size_t)offset);
                        if (!std::regex_search(tail, m, pt.sre)) break;
                        const long i = (long)m.position(0), len = (long)m.length(0);
                        long begin, end;
                        if (strand == 0) {
                            begin = offset + i + 1;
                            if (o.Circular && begin > l) break;
                            end = offset + i + len;
                        } else {
                            if (o.Circular && offset + i + 1 > l) break;
                            begin = l - offset - (i + len) + 1;
                            end = l - offset - i;
                            if (offset + i + len > l) { begin += l; end += l; }
                        }
                        bool flag = true;
                        for (size_t q = locs.size(); q-- > 0;)
                            if (locs[q].first <= begin && locs[q].second >= end) { flag = false; break; }
                        if (flag) {
                            row(r, pt, strand == 0 ? '+' : '-', begin, end, text.substr((size_t)(offset + i), (size_t)len));
                            locs.emplace_back(begin, end);
                        }
                        offset = o.NonGreedy ? offset + i + len + 1 : offset + i + 1;
                        if (offset >= n) break;
                    }
                }
            }
            continue;
        }
        for (auto& pt : pats) {
            const std::string& p = pt.bytes;
            const long lp = (long)p.size();  // a degenerate pattern matches exactly len(p) letters
            auto find_from = [&](const std::string& text, long offset) -> long {
                if (o.Degenerate) return pt.re.find(text, (size_t)offset);
                size_t f = text.find(p, (size_t)offset);
                return f == std::string::npos ? -1 : (long)f;
            };
            long offset = 0;
            for (;;) {  // locate.go:583-667 (the containment check at :604-614 cannot fire for fixed-length matches)
                const long f = find_from(r.seq, offset);
                if (f < 0) break;
                const long i = f - offset;
                const long begin = offset + i + 1;
                if (o.Circular && begin > l) break;
                const long end = offset + i + lp;
                row(r, pt, '+', begin, end, r.seq.substr((size_t)begin - 1, (size_t)(end - begin + 1)));
                offset = o.NonGreedy ? offset + i + lp + 1 : offset + i + 1;
                if (offset >= n) break;
            }
            if (o.OnlyPositiveStrand) continue;  // sic: the protein/unlimit auto-switch is not consulted here
            const std::string rp = rev_com(r.seq, rd.GetAlphabet());
            offset = 0;
            for (;;) {  // locate.go:679-766
                const long f = find_from(rp, offset);
                if (f < 0) break;
                const long i = f - offset;
                if (o.Circular && offset + i + 1 > l) break;
                long begin = l - offset - (i + lp) + 1, end = l - offset - i;
                if (offset + i + lp > l) { begin += l; end += l; }
                row(r, pt, '-', begin, end, rp.substr((size_t)(offset + i), (size_t)lp));
                offset = o.NonGreedy ? offset + i + lp + 1 : offset + i + 1;
                if (offset >= n) break;
            }
        }
    }
    return result;
}

// ---------------------------------------------------------------------------
// XXH64 (public algorithm; cespare/xxhash/v2 Sum64 == XXH64 with seed 0),
// pinned by tests/golden/xxh64_vectors.json
// ---------------------------------------------------------------------------
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

uint64_t xxh64(const void* data, size_t len, uint64_t seed) {
    const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
                   P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* end = p + len;
    uint64_t h;
    auto round = [&](uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl64(acc, 31); return acc * P1; };
    auto merge = [&](uint64_t acc, uint64_t v) { acc ^= round(0, v); return acc * P1 + P4; };
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = round(v1, rd64(p)); v2 = round(v2, rd64(p + 8)); v3 = round(v3, rd64(p + 16)); v4 = round(v4, rd64(p + 24));
            p += 32;
        } while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else {
        h = seed + P5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= round(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (uint64_t)(*p) * P5; h = rotl64(h, 11) * P1; ++p; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// ---------------------------------------------------------------------------
// RmDup  bigseqkit-lib/rmdup.go.  RmDupPrepare keys every record by
// int64(xxhash.Sum64(subject)); GroupByKey gathers equal keys; RmDupCheck keeps,
// inside a group, the first record of every distinct subject (and, with -s and
// without -P, drops a record whose reverse complement was already seen IN THE SAME
// hash group -- PARITY.md Q8).  Elements leave in first-occurrence file order.
// ---------------------------------------------------------------------------
std::vector<std::string> rmdup_call(const std::vector<std::string_view>& all, const RmDupOptions& o) {
    return rmdup_call_side(all, o, nullptr, nullptr);
}

// The same result with `threads` host threads -- bench.py's all-cores CPU baseline of `rmdup` (a labelled
// baseline, nothing the product runs).  The reference gets its parallelism from IgnisHPC: RmDupPrepare keys the records of
// every partition on its executor, GroupByKey brings equal keys together, RmDupCheck settles every key group on its own
// (bigseqkit/rmdup.go:88-105).  Restated with threads: phase 1, every thread parses a contiguous run of records (subject,
// XXH64 key, Format() text); phase 2, thread t owns the keys with key % threads == t, collects their records in file order
// and settles every group with rmdup_call_side's rule (a map keyed by the subject text, the reverse complement looked up
// inside the group).  tests/test_oracle_kat.py holds it to rmdup_call on random inputs.
std::vector<std::string> rmdup_call_mt(const std::vector<std::string_view>& all, const RmDupOptions& o, int threads) {
    Alphabet ab = alphabet_from_seqtype(o.Config.SeqType);
    if (o.BySeq && o.ByName) throw Error("only one/none of the flags -s (--by-seq) and -n (--by-name) is allowed");
    if (o.OnlyPositiveStrand && !o.BySeq) throw Error("flag -s (--by-seq) needed when using -P (--only-positive-strand)");
    if (threads < 1) threads = 1;
    struct Item { std::string subject, text, seq; uint64_t key; };
    const size_t N = all.size();
    std::vector<Item> items(N);
    std::vector<std::string> errors((size_t)threads);
    Alphabet fa = AB_NONE;
    {
        // (the alphabet of the partition is the one the FIRST record yields, as in the sequential reader)
        std::vector<std::string_view> head(all.begin(), all.begin() + std::min<size_t>(N, 1));
        SeqParser rd(ab, &head, o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
        while (rd.Read()) {}
        fa = rd.GetAlphabet();
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            try {
                const size_t a = N * (size_t)t / (size_t)threads, b = N * (size_t)(t + 1) / (size_t)threads;
                std::vector<std::string_view> part(all.begin() + a, all.begin() + b);
                SeqParser rd(ab, &part, o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
                int lineWidth = o.Config.LineWidth;
                size_t i = a;
                while (rd.Read()) {
                    Record& r = rd.rec;
                    if (rd.IsFastq) lineWidth = 0;
                    Item& it = items[i++];
                    it.subject = o.BySeq ? r.seq : (o.ByName ? r.name : r.id);
                    if (o.IgnoreCase) it.subject = lower(it.subject);
                    it.key = xxh64(it.subject.data(), it.subject.size(), 0);
                    it.text = record_format(r, rd.IsFastq, lineWidth);
                    it.seq = r.seq;
                }
            } catch (const std::exception& e) { errors[(size_t)t] = e.what(); }
        });
    for (auto& th : pool) th.join();
    for (auto& e : errors) if (!e.empty()) throw Error(e);
    const bool revcom = o.BySeq && !o.OnlyPositiveStrand;
    std::vector<char> keep(N, 0);
    pool.clear();
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            std::unordered_map<uint64_t, std::vector<size_t>> groups;
            for (size_t i = 0; i < N; ++i)
                if (items[i].key % (uint64_t)threads == (uint64_t)t) groups[items[i].key].push_back(i);
            for (auto& kv : groups) {
                auto& g = kv.second;  // (file order: i ascends)
                if (g.size() == 1) { keep[g[0]] = 1; continue; }
                std::map<std::string, size_t> counter;
                for (size_t i : g) {
                    if (counter.count(items[i].subject)) continue;
                    if (revcom) {
                        std::string rc = rev_com(items[i].seq, fa);
                        if (o.IgnoreCase) rc = lower(rc);
                        if (counter.count(rc)) continue;
                    }
                    counter[items[i].subject] = i;
                    keep[i] = 1;
                }
            }
        });
    for (auto& th : pool) th.join();
    std::vector<std::string> result;
    for (size_t i = 0; i < N; ++i)
        if (keep[i]) { std::string tx = std::move(items[i].text); tx.pop_back(); result.push_back(std::move(tx)); }
    return result;
}

// dup_seqs: Format() text of every removed record (rmdup.go:181-183, written by After() :246-261);
// dup_nums: "%d\t%s\n" = group size and ", "-joined IDs, survivor first (rmdup.go:229-236).  Removed records in file
// order; groups in the file order of their survivor (the reference iterates Go maps; PARITY.md Q9/Q10)
std::vector<std::string> rmdup_call_side(const std::vector<std::string_view>& all, const RmDupOptions& o,
                                         std::string* dup_seqs, std::string* dup_nums) {
    Alphabet ab = alphabet_from_seqtype(o.Config.SeqType);
    if (o.BySeq && o.ByName) throw Error("only one/none of the flags -s (--by-seq) and -n (--by-name) is allowed");
    if (o.OnlyPositiveStrand && !o.BySeq) throw Error("flag -s (--by-seq) needed when using -P (--only-positive-strand)");
    SeqParser rd(ab, &all, o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
    struct Item { std::string subject, text; uint64_t key; bool fastq; std::string seq, id; };
    std::vector<Item> items;
    int lineWidth = o.Config.LineWidth;
    while (rd.Read()) {
        Record& r = rd.rec;
        if (rd.IsFastq) lineWidth = 0;
        Item it;
        it.subject = o.BySeq ? r.seq : (o.ByName ? r.name : r.id);
        if (o.IgnoreCase) it.subject = lower(it.subject);
        it.key = xxh64(it.subject.data(), it.subject.size(), 0);
        it.text = record_format(r, rd.IsFastq, lineWidth);
        it.seq = r.seq;
        it.id = r.id;
        items.push_back(std::move(it));
    }
    const Alphabet fa = rd.GetAlphabet();
    const bool revcom = o.BySeq && !o.OnlyPositiveStrand;
    // groups in first-occurrence order
    std::map<uint64_t, std::vector<size_t>> groups;
    for (size_t i = 0; i < items.size(); ++i) groups[items[i].key].push_back(i);
    std::vector<char> keep(items.size(), 0);
    std::map<size_t, std::vector<size_t>> members;  // survivor index -> [survivor, removed...]
    for (auto& kv : groups) {
        auto& g = kv.second;
        if (g.size() == 1) { keep[g[0]] = 1; continue; }
        std::map<std::string, size_t> counter;  // subject -> its survivor
        for (size_t i : g) {
            const std::string& subject = items[i].subject;
            if (counter.count(subject)) { members[counter[subject]].push_back(i); continue; }
            if (revcom) {
                std::string rc = rev_com(items[i].seq, fa);
                if (o.IgnoreCase) rc = lower(rc);
                if (counter.count(rc)) { members[counter[rc]].push_back(i); continue; }
            }
            counter[subject] = i;
            members[i].push_back(i);
            keep[i] = 1;
        }
    }
    if (dup_seqs)
        for (size_t i = 0; i < items.size(); ++i)
            if (!keep[i]) *dup_seqs += items[i].text;
    if (dup_nums)
        for (auto& kv : members) {
            if (kv.second.size() < 2) continue;
            std::string l;
            for (size_t k = 0; k < kv.second.size(); ++k) l += (k ? ", " : "") + items[kv.second[k]].id;
            *dup_nums += std::to_string(kv.second.size()) + "\t" + l + "\n";
        }
    std::vector<std::string> result;
    for (size_t i = 0; i < items.size(); ++i)
        if (keep[i]) { std::string t = items[i].text; t.pop_back(); result.push_back(t); }
    return result;
}

// shenwei356/util/math.Round [upstream-memory]:
//   pow10_n := math.Pow10(n); return math.Trunc((f+0.5/pow10_n)*pow10_n) / pow10_n
double go_round(double f, int n) {
    double p = std::pow(10.0, n);
    return std::trunc((f + 0.5 / p) * p) / p;
}

// go-humanize Comma [upstream-memory]
std::string humanize_comma(int64_t v) {
    bool neg = v < 0;
    uint64_t u = neg ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v;
    std::string d = std::to_string(u), o;
    int c = 0;
    for (int i = (int)d.size() - 1; i >= 0; --i) {
        o.push_back(d[i]);
        if (++c % 3 == 0 && i > 0) o.push_back(',');
    }
    if (neg) o.push_back('-');
    std::reverse(o.begin(), o.end());
    return o;
}

// strconv.FormatFloat(v, 'f', -1, 64): shortest decimal that round-trips.
static std::string format_float_shortest(double v) {
    char b[64];
    for (int prec = 1; prec <= 17; ++prec) {
        snprintf(b, sizeof b, "%.*g", prec, v);
        if (strtod(b, nullptr) == v) break;
    }
    // %g may give exponent form; re-render as plain decimal
    std::string s(b);
    if (s.find('e') != std::string::npos || s.find('E') != std::string::npos) {
        // digits after the point needed: derive from exponent
        int dec = 0;
        for (; dec < 340; ++dec) {
            snprintf(b, sizeof b, "%.*f", dec, v);
            if (strtod(b, nullptr) == v) break;
        }
        s = b;
    }
    return s;
}

// go-humanize Commaf [upstream-memory]
std::string humanize_commaf(double v) {
    std::string s = format_float_shortest(std::fabs(v));
    size_t dot = s.find('.');
    std::string ip = s.substr(0, dot), fp = dot == std::string::npos ? "" : s.substr(dot);
    std::string o;
    int c = 0;
    for (int i = (int)ip.size() - 1; i >= 0; --i) {
        o.push_back(ip[i]);
        if (++c % 3 == 0 && i > 0) o.push_back(',');
    }
    std::reverse(o.begin(), o.end());
    if (v < 0) o = "-" + o;
    return o + fp;
}

// ---------------------------------------------------------------------------
// util.LengthStats  [upstream-memory: shenwei356/bio v0.7.0 util/length-stats.go]
// Built from (length,count) pairs; the reference calls Add(k) v times
// (bigseqkit/stats.go:134-138) which yields the same multiset.
// ---------------------------------------------------------------------------
namespace {
struct LengthStats {
    std::vector<std::pair<uint64_t, uint64_t>> counts;  // sorted by length
    uint64_t count = 0, sum = 0, mn = 0, mx = 0;
    explicit LengthStats(const std::map<int64_t, int64_t>& m) {
        for (auto& kv : m) {
            if (kv.second <= 0) continue;
            counts.emplace_back((uint64_t)kv.first, (uint64_t)kv.second);
            count += (uint64_t)kv.second;
            sum += (uint64_t)kv.first * (uint64_t)kv.second;
        }
        if (!counts.empty()) {
            mn = counts.front().first;
            mx = counts.back().first;
        }
    }
    double Mean() const { return (double)sum / (double)count; }
    uint64_t N50(int* l50) const {
        if (counts.empty()) { *l50 = 0; return 0; }
        double sumLen = 0, half = (double)sum / 2;
        uint64_t n = 0;
        for (size_t i = counts.size(); i-- > 0;) {
            // the reference adds one sequence at a time from the longest
            double need = half - sumLen;
            double len = (double)counts[i].first;
            double tot = len * (double)counts[i].second;
            if (sumLen + tot >= half) {
                uint64_t k = len > 0 ? (uint64_t)std::ceil(need / len) : 1;
                if (k < 1) k = 1;
                if (k > counts[i].second) k = counts[i].second;
                *l50 = (int)(n + k);
                return counts[i].first;
            }
            sumLen += tot;
            n += counts[i].second;
        }
        *l50 = (int)n;
        return counts.front().first;
    }
    double value_at(uint64_t idx) const {  // 0-based in the expanded sorted multiset
        uint64_t acc = 0;
        for (auto& c : counts) {
            acc += c.second;
            if (idx < acc) return (double)c.first;
        }
        return (double)mx;
    }
    double get(bool even, uint64_t l, uint64_t r) const {
        return even ? (value_at(l) + value_at(r)) / 2 : value_at(l);
    }
    double Q2() const {
        if (count == 0) return 0;
        if (count == 1) return (double)counts[0].first;
        bool even = (count & 1) == 0;
        return even ? get(true, count / 2 - 1, count / 2) : get(false, count / 2, 0);
    }
    double Q1() const {
        if (count == 0) return 0;
        if (count == 1) return (double)counts[0].first;
        uint64_t n = (count % 2 == 0) ? count / 2 : (count + 1) / 2;
        bool even = n % 2 == 0;
        return even ? get(true, n / 2 - 1, n / 2) : get(false, n / 2, 0);
    }
    double Q3() const {
        if (count == 0) return 0;
        if (count == 1) return (double)counts[0].first;
        uint64_t n, mean;
        if (count % 2 == 0) { n = count / 2; mean = n; }
        else { n = (count + 1) / 2; mean = n - 1; }
        bool even = n % 2 == 0;
        return even ? get(true, mean + n / 2 - 1, mean + n / 2) : get(false, mean + n / 2, 0);
    }
};
}  // namespace

// bigseqkit/stats.go:75-166
StatInfo stats_finalize(const std::string& name, const std::string& format, std::map<int64_t, int64_t> stats,
                        std::string_view first_record, const StatsOptions& o) {
    auto pop = [&](int64_t k) {
        int64_t v = 0;
        auto it = stats.find(k);
        if (it != stats.end()) { v = it->second; stats.erase(it); }
        return v;
    };
    int64_t q20 = pop(-1), q30 = pop(-2);
    uint64_t gapSum = (uint64_t)pop(-3);
    int64_t ti = pop(-4);
    std::string t;
    if (ti == 'D') t = "DNA";
    else if (ti == 'R') t = "RNA";
    else if (ti == 'U') t = "";
    else {
        // stats.go:117-129: fastx reader over the first record, alphabet guessed
        // from its sequence with the bio default threshold (10000).
        std::vector<std::string_view> one{first_record};
        SeqParser rd(AB_NONE, &one, o.Config.IDRegexp, 10000);
        if (!rd.Read()) throw Error("EOF");
        t = alphabet_name(rd.GetAlphabet());
    }
    LengthStats ls(stats);
    StatInfo info;
    info.file = name;
    info.format = format;
    info.t = t;
    if (ls.count > 0) {
        if (o.All) {
            info.N50 = ls.N50(&info.L50);
            info.Q1 = ls.Q1();
            info.Q2 = ls.Q2();
            info.Q3 = ls.Q3();
        }
        info.num = ls.count;
        info.lenSum = ls.sum;
        info.gapSum = gapSum;
        info.lenMin = ls.mn;
        info.lenAvg = go_round(ls.Mean(), 1);
        info.lenMax = ls.mx;
        info.q20 = go_round((double)q20 / (double)ls.sum * 100, 2);
        info.q30 = go_round((double)q30 / (double)ls.sum * 100, 2);
    }
    return info;
}

static std::string sprintf_s(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
static std::string sprintf_s(const char* fmt, ...) {
    char b[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(b, sizeof b, fmt, ap);
    va_end(ap);
    return b;
}

// go-prettytable rendering [upstream-memory]: each cell padded to its column's
// maximum width (left-aligned unless AlignRight), cells joined by the default
// separator " ", one '\n' per row, header row first.
static std::string pretty_table(const std::vector<std::string>& hdr, const std::vector<bool>& right,
                                const std::vector<std::string>& row) {
    std::string out;
    std::vector<size_t> w(hdr.size());
    for (size_t i = 0; i < hdr.size(); ++i) w[i] = std::max(hdr[i].size(), row[i].size());
    for (const auto* r : {&hdr, &row}) {
        for (size_t i = 0; i < hdr.size(); ++i) {
            if (i) out.push_back(' ');
            const std::string& c = (*r)[i];
            if (right[i]) out.append(w[i] - c.size(), ' ').append(c);
            else out.append(c).append(w[i] - c.size(), ' ');
        }
        out.push_back('\n');
    }
    return out;
}

// bigseqkit/stats.go:168-288
std::string stats_string(const StatInfo& info, const StatsOptions& o) {
    std::string result;
    if (o.Tabular) {
        result += "file\tformat\ttype\tnum_seqs\tsum_len\tmin_len\tavg_len\tmax_len";
        if (o.All) result += "\tQ1\tQ2\tQ3\tsum_gap\tN50\tQ20(%)\tQ30(%)";
        result += "\n";
        if (!o.All) {
            result += sprintf_s("%s\t%s\t%s\t%llu\t%llu\t%llu\t%.1f\t%llu\n", info.file.c_str(), info.format.c_str(),
                                info.t.c_str(), (unsigned long long)info.num, (unsigned long long)info.lenSum,
                                (unsigned long long)info.lenMin, info.lenAvg, (unsigned long long)info.lenMax);
        } else {
            result += sprintf_s("%s\t%s\t%s\t%llu\t%llu\t%llu\t%.1f\t%llu\t%.1f\t%.1f\t%.1f\t%llu\t%llu\t%.2f\t%.2f\n",
                                info.file.c_str(), info.format.c_str(), info.t.c_str(), (unsigned long long)info.num,
                                (unsigned long long)info.lenSum, (unsigned long long)info.lenMin, info.lenAvg,
                                (unsigned long long)info.lenMax, info.Q1, info.Q2, info.Q3,
                                (unsigned long long)info.gapSum, (unsigned long long)info.N50, info.q20, info.q30);
        }
        return result;
    }
    std::vector<std::string> hdr{"file", "format", "type", "num_seqs", "sum_len", "min_len", "avg_len", "max_len"};
    std::vector<bool> right{false, false, false, true, true, true, true, true};
    std::vector<std::string> row{info.file,
                                 info.format,
                                 info.t,
                                 humanize_comma((int64_t)info.num),
                                 humanize_comma((int64_t)info.lenSum),
                                 humanize_comma((int64_t)info.lenMin),
                                 humanize_commaf(info.lenAvg),
                                 humanize_comma((int64_t)info.lenMax)};
    if (o.All) {
        for (const char* h : {"Q1", "Q2", "Q3", "sum_gap", "N50", "Q20(%)", "Q30(%)"}) {
            hdr.push_back(h);
            right.push_back(true);
        }
        row.push_back(humanize_commaf(info.Q1));
        row.push_back(humanize_commaf(info.Q2));
        row.push_back(humanize_commaf(info.Q3));
        row.push_back(humanize_comma((int64_t)info.gapSum));
        row.push_back(humanize_comma((int64_t)info.N50));
        row.push_back(humanize_commaf(info.q20));
        row.push_back(humanize_commaf(info.q30));
    }
    return pretty_table(hdr, right, row);
}

// ---------------------------------------------------------------------------
// fq2fa, range / head, duplicate
// ---------------------------------------------------------------------------
// bigseqkit-lib/fq2fa.go:35-59
std::vector<std::string> fq2fa_call(const std::vector<std::string_view>& part, const KitConfig& cfg) {
    Alphabet ab = alphabet_from_seqtype(cfg.SeqType);  // :28
    SeqParser rd(ab, &part, cfg.IDRegexp, cfg.AlphabetGuessSeqLength);
    std::vector<std::string> result;
    while (rd.Read()) {
        Record r = rd.rec;
        r.qual.clear();                                  // :52
        std::string bb = record_format(r, false, 0);     // :53  (no quality -> FASTA layout)
        bb.pop_back();                                   // :55
        result.push_back(bb);
    }
    return result;
}

static int64_t parse_int_go(const std::string& s) {  // strconv.ParseInt(s, 10, 64)
    size_t i = 0;
    bool neg = false;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) neg = s[i++] == '-';
    if (i >= s.size()) throw Error("strconv.ParseInt: parsing \"" + s + "\": invalid syntax");
    int64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') throw Error("strconv.ParseInt: parsing \"" + s + "\": invalid syntax");
        v = v * 10 + (s[i] - '0');
    }
    return neg ? -v : v;
}

// bigseqkit/range.go:43-86
void range_bounds(const std::string& range, int64_t n_records, int64_t* pstart, int64_t* pend) {
    if (range.empty()) throw Error("flag -r (--range) needed");                     // :43-45
    std::vector<std::string> r;                                                      // :47
    for (size_t a = 0;;) {
        size_t b = range.find(':', a);
        r.push_back(range.substr(a, b == std::string::npos ? std::string::npos : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
    }
    int64_t start = parse_int_go(r[0]);                                              // :48-51
    int64_t end = -1;                                                                // :52
    if (r.size() > 1) end = parse_int_go(r[1]);                                      // :53-58
    if (start == 0 || end == 0) throw Error("either start and end should not be 0"); // :60-62
    if (start > 0) start--;                                                          // :64-66
    if (end == -1) end = INT64_MAX;                                                  // :67-69
    if (start < -1 || end < -1) {                                                    // :71-83
        if (start < 0) start += n_records;
        if (end < 0) end += n_records;
    }
    // :85-87 reads `if start <= end { error }` -- every non-empty range would be refused (PARITY.md RNG)
    if (start >= end) throw Error("start must be > than end");
    *pstart = start;
    *pend = end;
}

// bigseqkit-lib/range.go:33-43
std::vector<std::string> range_call(const std::vector<std::string_view>& part, int64_t first, int64_t start, int64_t end) {
    std::vector<std::string> result;
    for (size_t i = 0; i < part.size(); ++i) {
        const int64_t v1 = first + (int64_t)i;
        std::string v = start <= v1 && v1 < end ? std::string(part[i]) : std::string();  // :33-38
        if (!v.empty()) result.push_back(v);                                              // :41-43
    }
    return result;
}

// bigseqkit-lib/duplicate.go:24-30
std::vector<std::string> duplicate_call(const std::vector<std::string_view>& part, int64_t times) {
    if (times < 0) throw Error("value of -n (--times) should not be negative");  // make([]string, times) would panic
    std::vector<std::string> result;
    for (auto& v : part)
        for (int64_t i = 0; i < times; ++i) result.push_back(std::string(v));
    return result;
}

// ---------------------------------------------------------------------------
// rename  (bigseqkit-lib/rename.go)
// ---------------------------------------------------------------------------
std::vector<std::string> rename_call(const std::vector<std::string_view>& all, const KitConfig& cfg, bool by_name) {
    Alphabet ab = alphabet_from_seqtype(cfg.SeqType);  // :32
    SeqParser rd(ab, &all, cfg.IDRegexp, cfg.AlphabetGuessSeqLength);
    std::map<std::string, int64_t> numbers;  // per group, in arrival (file) order  (:106)
    std::vector<std::string> result;
    int lineWidth = cfg.LineWidth;
    while (rd.Read()) {
        Record r = rd.rec;
        if (rd.IsFastq) lineWidth = 0;                       // :56-59, :113-116
        const std::string k = by_name ? r.name : r.id;       // :61-65
        int64_t& n = numbers[k];
        if (n > 0) {                                         // :118-121
            const std::string newID = r.id + "_" + std::to_string(n);
            r.name = newID + " " + r.desc;
        }
        ++n;                                                 // :123
        std::string bb = record_format(r, rd.IsFastq, lineWidth);
        bb.pop_back();                                       // :125
        result.push_back(bb);
    }
    return result;
}

// ---------------------------------------------------------------------------
// sort  (bigseqkit/sort.go, bigseqkit-lib/sort.go)
// ---------------------------------------------------------------------------
std::vector<std::string> sort_call(const std::vector<std::string_view>& all, const SortOptions& o) {
    bool byLength = o.ByLength;
    if (o.ByBases) byLength = true;                                   // sort.go:105-108
    int n = (o.BySeq ? 1 : 0) + (o.ByName ? 1 : 0) + (byLength ? 1 : 0);
    if (n > 1) throw Error("only one of the options (byLength), (byName) and (bySeq) is allowed");  // :110-122
    // natural order (sort.go:130-133 -> natsort.Compare of shenwei356/natsort, not in the tree [upstream-memory]): the key
    // is cut into runs of digits and runs of other bytes; at the first differing run, two digit runs compare as
    // integers (Atoi: runs beyond int64 fall back to string comparison), anything else as strings; a key whose runs
    // are exhausted first comes first
    const bool natural = !byLength && !o.BySeq && o.InNaturalOrder;
    auto chunkify = [](const std::string& t) {
        std::vector<std::string> ch;
        for (size_t i = 0; i < t.size();) {
            const bool dig = isdigit((unsigned char)t[i]) != 0;
            size_t j = i;
            while (j < t.size() && (isdigit((unsigned char)t[j]) != 0) == dig) ++j;
            ch.push_back(t.substr(i, j - i));
            i = j;
        }
        return ch;
    };
    auto nat_cmp = [&](const std::string& a, const std::string& b) {
        const auto ca = chunkify(a), cb = chunkify(b);
        for (size_t i = 0; i < ca.size(); ++i) {
            if (i >= cb.size()) return 1;
            const bool na = isdigit((unsigned char)ca[i][0]) && ca[i].size() <= 18, nb = isdigit((unsigned char)cb[i][0]) && cb[i].size() <= 18;
            if (na && nb) {
                const long long x = atoll(ca[i].c_str()), y = atoll(cb[i].c_str());
                if (x != y) return x < y ? -1 : 1;
            } else if (ca[i] != cb[i]) {
                return ca[i] < cb[i] ? -1 : 1;
            }
        }
        return ca.size() < cb.size() ? -1 : 0;
    };
    Alphabet ab = alphabet_from_seqtype(o.Config.SeqType);
    SeqParser rd(ab, &all, o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
    std::bitset<256> gaps;
    for (unsigned char ch : o.GapLetters) gaps.set(ch);
    struct Item { std::string skey; int32_t ikey; std::string text; };
    std::vector<Item> items;
    int lineWidth = o.Config.LineWidth;
    while (rd.Read()) {
        const Record& r = rd.rec;
        if (rd.IsFastq) lineWidth = 0;                                // lib/sort.go:58-61, 140-143
        Item it;
        it.text = record_format(r, rd.IsFastq, lineWidth);
        if (!it.text.empty() && it.text.back() == '\n') it.text.pop_back();  // :62-67
        it.ikey = 0;
        if (byLength) {                                               // :152-156
            if (o.ByBases) {
                int32_t b = 0;
                for (unsigned char ch : r.seq) b += gaps.test(ch) ? 0 : 1;   // Seq.Bases(gapLetters)
                it.ikey = b;
            } else {
                it.ikey = (int32_t)r.seq.size();
            }
        } else if (o.ByName) {                                        // :69-74
            it.skey = o.IgnoreCase ? lower(r.name) : r.name;
        } else if (o.BySeq) {                                         // :75-88
            std::string s = (o.SeqPrefixLength == 0 || (int64_t)r.seq.size() <= o.SeqPrefixLength)
                                ? r.seq : r.seq.substr(0, (size_t)o.SeqPrefixLength);
            it.skey = o.IgnoreCase ? lower(s) : s;
        } else {                                                      // :89-95
            it.skey = o.IgnoreCase ? lower(r.id) : r.id;
        }
        items.push_back(std::move(it));
    }
    // SortByKey(ascending = !reverse, less)
    if (byLength) {
        if (!o.Reverse) std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.ikey < b.ikey; });
        else std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.ikey > b.ikey; });
    } else if (natural) {
        if (!o.Reverse) std::stable_sort(items.begin(), items.end(), [&](const Item& a, const Item& b) { return nat_cmp(a.skey, b.skey) < 0; });
        else std::stable_sort(items.begin(), items.end(), [&](const Item& a, const Item& b) { return nat_cmp(a.skey, b.skey) > 0; });
    } else {
        if (!o.Reverse) std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.skey < b.skey; });
        else std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.skey > b.skey; });
    }
    std::vector<std::string> result;
    for (auto& it : items) result.push_back(std::move(it.text));
    return result;
}

// ---------------------------------------------------------------------------
// faidx index rows  (bigseqkit-lib/faidx.go:91-229)
// ---------------------------------------------------------------------------
std::vector<std::string> faidx_call(const std::vector<std::string_view>& part, uint64_t base, bool full_head,
                                    const KitConfig& cfg, uint64_t* bytes) {
    std::vector<std::string> result;
    uint64_t at = base;  // file offset of the current element
    const bool default_re = cfg.IDRegexp == "^(\\S+)\\s?";
    for (auto elem : part) {
        // lines of the element (bytes.Split(seqBlock, "\n"), :110)
        std::vector<std::string_view> lines;
        for (size_t a = 0;;) {
            size_t b = elem.find('\n', a);
            lines.push_back(elem.substr(a, b == std::string_view::npos ? std::string_view::npos : b - a));
            if (b == std::string_view::npos) break;
            a = b + 1;
        }
        const std::string head(lines[0].substr(1));
        std::string id, desc;
        if (full_head) id = head;                                       // ^(.+)$  (:69-73)
        else if (default_re) {                                          // parseHeadID :434-444
            size_t i = head.find(' ');
            if (i != std::string::npos && i > 0) id = head.substr(0, i);
            else { i = head.find('\t'); id = (i != std::string::npos && i > 0) ? head.substr(0, i) : head; }
        } else parse_head_id_desc(head, false, cfg.IDRegexp, id, desc);
        const uint64_t seq_start = at + lines[0].size() + 1;            // lastStart :159-160
        std::vector<uint64_t> lineWidths, seqWidths;
        uint64_t seqLen = 0, cur = seq_start, iqual = 0;
        bool fastq = lines[0][0] == '@', qline = false, have_qual = false;
        for (size_t k = 1; k < lines.size(); ++k) {
            const std::string_view line = lines[k];
            if (fastq && !qline && !have_qual && !line.empty() && line[0] == '+') {  // :111-114
                iqual = cur + line.size() + 1;
                qline = true;
            } else if (qline) {                                         // the quality line :169-171
                qline = false;
                have_qual = true;
            } else if (!have_qual) {                                    // :163-168
                seqLen += line.size();
                lineWidths.push_back(line.size() + 1);
                seqWidths.push_back(line.size());
            }
            cur += line.size() + 1;
        }
        // check lineWidths :117-137
        long long lastLineWidth = -1;
        int chances = 2;
        bool seenSeqs = false;
        for (size_t i = lineWidths.size(); i-- > 0;) {
            if (!seenSeqs && seqWidths[i] == 0) continue;
            seenSeqs = true;
            if (lastLineWidth == -1) { lastLineWidth = (long long)lineWidths[i]; continue; }
            if ((long long)lineWidths[i] != lastLineWidth) {
                chances--;
                if (chances == 0 || (long long)lineWidths[i] < lastLineWidth)
                    throw Error("different line length in sequence: " + id + ". Please format the file with 'seqkit seq'");
            }
            lastLineWidth = (long long)lineWidths[i];
        }
        const uint64_t lineWidth = lineWidths.empty() ? 0 : lineWidths[0];   // :139-146 (stale values of the previous
        const uint64_t seqWidth = seqWidths.empty() ? 0 : seqWidths[0];      //  record as written; FAI)
        std::string row = id + "\t" + std::to_string(seqLen) + "\t" + std::to_string(seq_start) + "\t" +
                          std::to_string(seqWidth) + "\t" + std::to_string(lineWidth);
        if (fastq) row += "\t" + std::to_string(iqual);                // :148-152
        result.push_back(row);
        at += elem.size() + 1;                                          // FaidxOffset: len(elem) + 1
    }
    if (bytes) *bytes = at - base;
    return result;
}

// ---------------------------------------------------------------------------
// pair  (bigseqkit-lib/pair.go)
// ---------------------------------------------------------------------------
void pair_call(const std::vector<std::string_view>& a, const std::vector<std::string_view>& b, const KitConfig& cfg,
               std::vector<std::string> out[4]) {
    Alphabet ab = alphabet_from_seqtype(cfg.SeqType);
    struct Rec { std::string id, text; };
    auto prepare = [&](const std::vector<std::string_view>& part) {  // PairPrepare.Call :37-65
        std::vector<Rec> v;
        SeqParser rd(ab, &part, cfg.IDRegexp, cfg.AlphabetGuessSeqLength);
        int lineWidth = cfg.LineWidth;
        while (rd.Read()) {
            if (rd.IsFastq) lineWidth = 0;
            std::string t = record_format(rd.rec, rd.IsFastq, lineWidth);
            t.pop_back();
            v.push_back({rd.rec.id, t});
        }
        return v;
    };
    const std::vector<Rec> ra = prepare(a), rb = prepare(b);
    std::map<std::string, std::vector<size_t>> f2;  // ID -> its records in file 2, in order
    for (size_t j = 0; j < rb.size(); ++j) f2[rb[j].id].push_back(j);
    std::map<std::string, size_t> used;             // ID -> records of file 1 seen so far
    std::vector<char> b_paired(rb.size(), 0);
    for (size_t i = 0; i < ra.size(); ++i) {        // Pair.Call :86-121, groups visited in file-1 order
        const size_t k = used[ra[i].id]++;
        auto it = f2.find(ra[i].id);
        if (it != f2.end() && k < it->second.size()) {
            out[0].push_back(ra[i].text);
            out[1].push_back(rb[it->second[k]].text);
            b_paired[it->second[k]] = 1;
        } else {
            out[2].push_back(ra[i].text);
        }
    }
    for (size_t j = 0; j < rb.size(); ++j)
        if (!b_paired[j]) out[3].push_back(rb[j].text);
}

// ---------------------------------------------------------------------------
// common  (bigseqkit/common.go:56-109, bigseqkit-lib/common.go:31-212; PARITY.md COMMON)
// ---------------------------------------------------------------------------
std::vector<std::string> common_call(const std::vector<std::vector<std::string_view>>& files, const CommonOptions& o) {
    if (o.BySeq && o.ByName) throw Error("only one/none of the flags -s (--by-seq) and -n (--by-name) is allowed");  // :37-39
    if (o.OnlyPositiveStrand && !o.BySeq) throw Error("flag -s (--by-seq) needed when using -P (--only-positive-strand)");  // :43-45
    Alphabet ab = alphabet_from_seqtype(o.Config.SeqType);
    auto key_of = [&](const Record& r) {
        std::string k = o.BySeq ? r.seq : (o.ByName ? r.name : r.id);  // :76-100 (the revcom of both sides == same strand)
        return o.IgnoreCase ? lower(k) : k;
    };
    std::vector<std::set<std::string>> present(files.size());
    for (size_t f = 1; f < files.size(); ++f) {
        SeqParser rd(ab, &files[f], o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
        while (rd.Read()) present[f].insert(key_of(rd.rec));
    }
    std::vector<std::string> result;
    std::set<std::string> done;
    SeqParser rd(ab, &files[0], o.Config.IDRegexp, o.Config.AlphabetGuessSeqLength);
    int lineWidth = o.Config.LineWidth;
    while (rd.Read()) {
        if (rd.IsFastq) lineWidth = 0;
        const std::string k = key_of(rd.rec);
        if (!done.insert(k).second) continue;
        bool all = true;
        for (size_t f = 1; f < files.size() && all; ++f) all = present[f].count(k) != 0;
        if (!all) continue;
        std::string bb = record_format(rd.rec, rd.IsFastq, lineWidth);
        bb.pop_back();
        result.push_back(bb);
    }
    return result;
}

// ---------------------------------------------------------------------------
// concat  (bigseqkit-lib/concat.go)
// ---------------------------------------------------------------------------
std::vector<std::string> concat_call(const std::vector<std::string_view>& a, const std::vector<std::string_view>& b,
                                     const KitConfig& cfg, bool full) {
    Alphabet ab = alphabet_from_seqtype(cfg.SeqType);
    struct Rec { Record r; bool fastq; };
    int lineWidth = cfg.LineWidth;
    auto parse = [&](const std::vector<std::string_view>& part) {
        std::vector<Rec> v;
        SeqParser rd(ab, &part, cfg.IDRegexp, cfg.AlphabetGuessSeqLength);
        while (rd.Read()) {
            if (rd.IsFastq) lineWidth = 0;               // :59-62
            v.push_back({rd.rec, rd.IsFastq});
        }
        return v;
    };
    const std::vector<Rec> ra = parse(a), rb = parse(b);
    std::map<std::string, std::vector<size_t>> f2;
    std::set<std::string> ids1;
    for (size_t j = 0; j < rb.size(); ++j) f2[rb[j].r.id].push_back(j);
    for (auto& x : ra) ids1.insert(x.r.id);
    std::vector<std::string> result;
    auto keep = [&](const Rec& x) {                      // :112-127 (one side only, Full)
        std::string t = record_format(x.r, x.fastq, lineWidth);
        t.pop_back();
        result.push_back(t);
    };
    for (auto& x : ra) {
        auto it = f2.find(x.r.id);
        if (it == f2.end()) { if (full) keep(x); continue; }
        for (size_t j : it->second) {                    // :129-146: every A with every B
            Record n;
            n.id = x.r.id;
            n.name = x.r.id;                             // Name: recordA.ID -- the description is not printed
            n.seq = x.r.seq + rb[j].r.seq;
            n.qual = x.r.qual + rb[j].r.qual;
            std::string t = record_format(n, x.fastq, lineWidth);
            t.pop_back();
            result.push_back(t);
        }
    }
    if (full)
        for (auto& y : rb)
            if (!ids1.count(y.r.id)) keep(y);
    return result;
}

// parseRegion  bigseqkit-lib/faidx.go:536-567 (the four Go regexps, lazy id)
static void parse_faidx_region(const std::string& region, std::string* id, long* begin, long* end) {
    static const std::regex full("^(.+?):(-?\\d+)-(-?\\d+)$"), one("^(.+?):(\\d+)$"), onlyb("^(.+?):(-?\\d+)-$"),
        onlye("^(.+?):-(-?\\d+)$");
    std::smatch m;
    if (std::regex_match(region, m, full)) { *id = m[1]; *begin = atol(m[2].str().c_str()); *end = atol(m[3].str().c_str()); }
    else if (std::regex_match(region, m, one)) { *id = m[1]; *begin = atol(m[2].str().c_str()); *end = *begin; }
    else if (std::regex_match(region, m, onlyb)) { *id = m[1]; *begin = atol(m[2].str().c_str()); *end = -1; }
    else if (std::regex_match(region, m, onlye)) { *id = m[1]; *begin = 1; *end = atol(m[2].str().c_str()); }
    else { *id = region; *begin = 1; *end = -1; }
}

std::vector<std::string> faidx_query_call(const std::vector<std::string_view>& part, const std::vector<std::string>& queries,
                                          bool ignore_case, const KitConfig& cfg, bool use_regexp) {
    Alphabet ab = alphabet_from_seqtype(cfg.SeqType);
    if (use_regexp) {  // :312-319, :362-368: a record whose ID matches any expression comes back whole
        std::vector<std::regex> res;
        for (auto& q : queries) {
            try { res.emplace_back(q, std::regex::ECMAScript); }
            catch (const std::regex_error&) { throw Error("invalid regular expression: " + q); }
        }
        SeqParser rd(ab, &part, cfg.IDRegexp, cfg.AlphabetGuessSeqLength);
        std::vector<std::string> result;
        while (rd.Read()) {
            bool ok = false;
            for (auto& re : res) if (std::regex_search(rd.rec.id, re)) { ok = true; break; }
            if (!ok || rd.rec.seq.empty()) continue;   // SubLocation of an empty sequence is not ok
            result.push_back(">" + rd.rec.id + "\n" + wrap_byte_slice(rd.rec.seq, cfg.LineWidth));
        }
        return result;
    }
    struct Q { std::string id; long b, e; };
    std::vector<Q> qs;
    for (auto& r : queries) {                              // :316-327
        Q q;
        parse_faidx_region(r, &q.id, &q.b, &q.e);
        if (ignore_case) q.id = lower(q.id);
        qs.push_back(q);
    }
    SeqParser rd(ab, &part, cfg.IDRegexp, cfg.AlphabetGuessSeqLength);
    std::vector<std::string> result;
    while (rd.Read()) {
        const Record& r = rd.rec;
        std::string id = ignore_case ? lower(r.id) : r.id;  // :369-372
        const Q* hit = nullptr;
        for (auto& q : qs) if (q.id == id) { hit = &q; break; }   // :373-380
        if (!hit) continue;
        const bool whole = (hit->b == 1 && hit->e == -1) || (hit->b > 0 && hit->e < 0);  // :388
        const bool rc = !whole && hit->b > hit->e;
        size_t b0, e0;
        sub_location(r.seq.size(), (int)(rc ? hit->e : hit->b), (int)(rc ? hit->b : hit->e), &b0, &e0);
        if (b0 == e0) continue;                              // !ok
        std::string sub = r.seq.substr(b0, e0 - b0);
        if (rc) sub = rev_com(sub, rd.GetAlphabet());
        std::string head = r.id;                             // parseHeadID(record.Name)
        std::string text = ">" + head + (whole ? "" : ":" + std::to_string(hit->b) + "-" + std::to_string(hit->e)) + "\n" +
                           wrap_byte_slice(sub, cfg.LineWidth);
        result.push_back(text);
    }
    return result;
}

}  // namespace orc
