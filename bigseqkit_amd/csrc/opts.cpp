#include "opts.hpp"

#include <cmath>
#include <cstring>

#include "json.hpp"

namespace bsk {

namespace {
Field fb(const char* n, bool v) { Field f; f.name = n; f.type = FieldType::Bool; f.b = v; return f; }
Field fi(const char* n, int64_t v) { Field f; f.name = n; f.type = FieldType::Int; f.i = v; return f; }
Field ff(const char* n, double v) { Field f; f.name = n; f.type = FieldType::Float; f.f = v; return f; }
Field fs(const char* n, const char* v) { Field f; f.name = n; f.type = FieldType::String; f.s = v; return f; }
Field fl(const char* n, std::vector<std::string> v) { Field f; f.name = n; f.type = FieldType::StringList; f.sl = std::move(v); return f; }
Field fnull(const char* n) { Field f; f.name = n; f.type = FieldType::Int; f.is_null = true; return f; }

std::vector<Field> kitconfig_defaults() {  // bigseqkit/helper.go:86-103
    return {fs("SeqType", "auto"), fnull("ChunkSize"), fnull("BufferSize"), fi("LineWidth", 60),
            fs("IDRegexp", "^(\\S+)\\s?"), fb("IDNCBI", false), fb("Quiet", false),
            fi("AlphabetGuessSeqLength", 10000), fi("ValidateSeqLength", 10000)};
}
}  // namespace

Options Options::defaults(Op op) {
    Options o;
    o.op = op;
    o.config = kitconfig_defaults();
    switch (op) {
        case Op::Stats:  // bigseqkit/stats.go:28-38
            o.fields = {fb("Tabular", false), fs("GapLetters", "- ."), fb("All", false), fb("SkipErr", false),
                        fs("FqEncoding", "sanger"), fb("Basename", false)};
            break;
        case Op::Seq:  // bigseqkit/seq.go:32-55
            o.fields = {fb("Reverse", false), fb("Complement", false), fb("Name", false), fb("Seq", false),
                        fb("Qual", false), fb("OnlyId", false), fb("RemoveGaps", false), fs("GapLetters", "- \t."),
                        fb("LowerCase", false), fb("UpperCase", false), fb("Dna2rna", false), fb("Rna2dna", false),
                        fb("ValidateSeq", false), fi("ValidateSeqLength", 10000), fi("MaxLen", -1), fi("MinLen", -1),
                        fi("QualAsciiBase", 33), ff("MinQual", -1), ff("MaxQual", -1)};
            break;
        case Op::Grep:  // bigseqkit/grep.go:31-49
            o.fields = {fl("Pattern", {""}), fs("PatternFile", ""), fb("UseRegexp", false), fb("DeleteMatched", false),
                        fb("InvertMatch", false), fb("ByName", false), fb("BySeq", false),
                        fb("OnlyPositiveStrand", false), fi("MaxMismatch", 0), fb("IgnoreCase", false),
                        fb("Degenerate", false), fs("Region", ""), fb("Circular", false), fb("Count", false)};
            break;
        case Op::Locate:  // bigseqkit/locate.go:27-45
            o.fields = {fl("Pattern", {""}), fs("PatternFile", ""), fb("Degenerate", false), fb("UseRegexp", false),
                        fb("UseFmi", false), fb("IgnoreCase", false), fb("OnlyPositiveStrand", false),
                        fi("ValidateSeqLength", 10000), fb("NonGreedy", false), fb("Gtf", false), fb("Bed", false),
                        fi("MaxMismatch", 0), fb("HideMatched", false), fb("Circular", false)};
            break;
        case Op::Subseq:  // bigseqkit/subseq.go:22-35
            o.fields = {fl("Chr", {}), fs("Region", ""), fs("Gtf", ""), fl("Feature", {}), fi("UpStream", 0),
                        fi("DownStream", 0), fb("OnlyFlank", false), fs("Bed", ""), fs("GtfTag", "")};
            break;
        case Op::Translate:  // bigseqkit/translate.go:22-35
            o.fields = {fi("TranslTable", 1), fl("Frame", {"1"}), fb("Trim", false), fb("Clean", false),
                        fb("AllowUnknownCodon", false), fb("InitCodonAsM", false), fi("ListTranslTable", -1),
                        fi("ListTranslTableWithAmbCodons", -1), fb("AppendFrame", false)};
            break;
        case Op::RmDup:  // bigseqkit/rmdup.go:23-33
            o.fields = {fb("ByName", false), fb("BySeq", false), fb("IgnoreCase", false), fs("DupSeqsFile", ""),
                        fs("DupNumFile", ""), fb("OnlyPositiveStrand", false)};
            break;
        case Op::Fq2Fa: break;                                      // bigseqkit/fq2fa.go:15-18
        case Op::Range: o.fields = {fs("Range", "")}; break;        // bigseqkit/range.go:19-24
        case Op::Head: o.fields = {fi("N", 10)}; break;             // bigseqkit/head.go:17-22
        case Op::Duplicate: o.fields = {fi("Times", 1)}; break;     // bigseqkit/duplicate.go:14-19
        case Op::Rename: o.fields = {fb("ByName", false)}; break;   // bigseqkit/rename.go:17-22
        case Op::Pair: o.fields = {fb("SaveUnpaired", false)}; break;   // bigseqkit/pair.go:17-22
        case Op::Concat: o.fields = {fb("Full", false), fs("Separator", "|")}; break;   // bigseqkit/concat.go:18-24
        case Op::Common:  // bigseqkit/common.go:21-29
            o.fields = {fb("ByName", false), fb("BySeq", false), fb("IgnoreCase", false), fb("OnlyPositiveStrand", false)};
            break;
        case Op::Faidx:  // bigseqkit/faidx.go:20-29
            o.fields = {fb("UseRegexp", false), fb("IgnoreCase", false), fb("FullHead", false), fs("RegionFile", ""),
                        fl("Regions", {})};
            break;
        case Op::Sort:  // bigseqkit/sort.go:26-39
            o.fields = {fb("InNaturalOrder", false), fb("BySeq", false), fb("ByName", false), fb("ByLength", false),
                        fb("ByBases", false), fs("GapLetters", "- \t."), fb("Reverse", false), fb("IgnoreCase", false),
                        fi("SeqPrefixLength", 10000)};
            break;
    }
    return o;
}

static void assign(Field& f, const json::Value& v) {
    using K = json::Value;
    if (v.kind == K::Null) return;  // nil pointer -> setDefaults keeps the default
    switch (f.type) {
        case FieldType::Bool:
            if (v.kind != K::Bool) throw OptError("invalid options JSON: field " + f.name + " must be a bool");
            f.b = v.b;
            break;
        case FieldType::Int:
            if (v.kind != K::Number || !v.is_int)
                throw OptError("invalid options JSON: field " + f.name + " must be an integer");
            f.i = v.inum;
            f.is_null = false;
            break;
        case FieldType::Float:
            if (v.kind != K::Number) throw OptError("invalid options JSON: field " + f.name + " must be a number");
            f.f = v.num;
            break;
        case FieldType::String:
            if (v.kind != K::String) throw OptError("invalid options JSON: field " + f.name + " must be a string");
            f.s = v.str;
            break;
        case FieldType::StringList:
            if (v.kind != K::Array) throw OptError("invalid options JSON: field " + f.name + " must be an array");
            f.sl.clear();
            for (auto& e : v.arr) {
                if (e->kind != K::String)
                    throw OptError("invalid options JSON: field " + f.name + " must be an array of strings");
                f.sl.push_back(e->str);
            }
            break;
    }
}

Options Options::from_json(Op op, const std::string& text) {
    Options o = defaults(op);
    json::ValuePtr root;
    try {
        std::string t = text;
        root = json::Parser(t).parse();
    } catch (const std::exception& e) { throw OptError(e.what()); }
    if (root->kind != json::Value::Object) throw OptError("invalid options JSON: not an object");
    for (auto& kv : root->obj) {
        if (kv.first == "Config") {
            if (kv.second->kind == json::Value::Null) continue;
            if (kv.second->kind != json::Value::Object) throw OptError("invalid options JSON: Config must be an object");
            for (auto& ckv : kv.second->obj) {
                bool found = false;
                for (auto& f : o.config)
                    if (f.name == ckv.first) { assign(f, *ckv.second); found = true; break; }
                (void)found;  // unknown keys are ignored like encoding/json does
            }
            continue;
        }
        for (auto& f : o.fields)
            if (f.name == kv.first) { assign(f, *kv.second); break; }
    }
    // KitConfig.setDefaults: IDNCBI overrides IDRegexp (bigseqkit/helper.go:97-100)
    if (o.cb("IDNCBI")) o.cmut("IDRegexp").s = "\\|([^\\|]+)\\| ";
    return o;
}

static std::string go_float(double v) {
    // encoding/json: shortest representation, 'f' unless exponent < -6 || >= 21
    char b[64];
    for (int prec = 1; prec <= 17; ++prec) {
        snprintf(b, sizeof b, "%.*g", prec, v);
        if (strtod(b, nullptr) == v) break;
    }
    std::string s(b);
    if (s.find('e') != std::string::npos) {
        double a = std::fabs(v);
        if (a >= 1e-6 && a < 1e21) {
            for (int dec = 0; dec < 30; ++dec) {
                snprintf(b, sizeof b, "%.*f", dec, v);
                if (strtod(b, nullptr) == v) break;
            }
            s = b;
        }
    }
    return s;
}

static void emit(std::string& o, const Field& f) {
    o += json::quote(f.name);
    o += ":";
    if (f.is_null) { o += "null"; return; }
    switch (f.type) {
        case FieldType::Bool: o += f.b ? "true" : "false"; break;
        case FieldType::Int: o += std::to_string(f.i); break;
        case FieldType::Float: o += go_float(f.f); break;
        case FieldType::String: o += json::quote(f.s); break;
        case FieldType::StringList:
            o += "[";
            for (size_t i = 0; i < f.sl.size(); ++i) {
                if (i) o += ",";
                o += json::quote(f.sl[i]);
            }
            o += "]";
            break;
    }
}

std::string Options::to_json() const {
    std::string o = "{\"Config\":{";
    for (size_t i = 0; i < config.size(); ++i) {
        if (i) o += ",";
        emit(o, config[i]);
    }
    o += "}";
    for (auto& f : fields) {
        o += ",";
        emit(o, f);
    }
    o += "}\n";
    return o;
}

const Field& Options::find(const char* n) const {
    for (auto& f : fields)
        if (f.name == n) return f;
    throw std::logic_error(std::string("no option field ") + n);
}
const Field& Options::cfind(const char* n) const {
    for (auto& f : config)
        if (f.name == n) return f;
    throw std::logic_error(std::string("no config field ") + n);
}

bool op_from_name(const std::string& name, Op* out) {
    struct { const char* n; Op op; } tbl[] = {
        {"Stats", Op::Stats}, {"SeqTransform", Op::Seq}, {"Seq", Op::Seq}, {"Grep", Op::Grep},
        {"Locate", Op::Locate}, {"SubseqTransform", Op::Subseq}, {"Subseq", Op::Subseq},
        {"Translate", Op::Translate}, {"RmDup", Op::RmDup}, {"RmDupPrepare", Op::RmDup}, {"RmDupCheck", Op::RmDup},
        {"Fq2Fa", Op::Fq2Fa}, {"Range", Op::Range}, {"RangePrepare", Op::Range}, {"Head", Op::Head},
        {"Duplicate", Op::Duplicate}, {"Rename", Op::Rename}, {"RenamePrepare", Op::Rename},
        {"Sort", Op::Sort}, {"Faidx", Op::Faidx}, {"Pair", Op::Pair},
        {"PairPrepare", Op::Pair}, {"Common", Op::Common}, {"CommonPrepare", Op::Common},
        {"Concat", Op::Concat}, {"ConcatPrepare", Op::Concat}};
    for (auto& t : tbl)
        if (name == t.n) { *out = t.op; return true; }
    return false;
}

const char* op_name(Op op) {
    switch (op) {
        case Op::Stats: return "Stats";
        case Op::Seq: return "SeqTransform";
        case Op::Grep: return "Grep";
        case Op::Locate: return "Locate";
        case Op::Subseq: return "SubseqTransform";
        case Op::Translate: return "Translate";
        case Op::RmDup: return "RmDup";
        case Op::Fq2Fa: return "Fq2Fa";
        case Op::Range: return "Range";
        case Op::Head: return "Head";
        case Op::Duplicate: return "Duplicate";
        case Op::Rename: return "Rename";
        case Op::Sort: return "Sort";
        case Op::Faidx: return "Faidx";
        case Op::Pair: return "Pair";
        case Op::Common: return "Common";
        case Op::Concat: return "Concat";
    }
    return "";
}

// ---------------------------------------------------------------------------
static std::string lower(std::string s) {
    for (auto& c : s)
        if (c >= 'A' && c <= 'Z') c += 32;
    return s;
}

Alphabet alphabet_from_seqtype(const std::string& t) {  // bigseqkit/helper.go:68-84
    std::string v = lower(t);
    if (v == "dna") return AB_DNAredundant;
    if (v == "rna") return AB_RNAredundant;
    if (v == "protein") return AB_PROTEIN;
    if (v == "unlimit") return AB_UNLIMIT;
    if (v == "auto") return AB_NONE;
    throw OptError("invalid sequence type: " + t + ", available value: dna|rna|protein|unlimit|auto");
}

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

namespace {
// letters + gap + ambiguous letters of each bio alphabet, as a 256-bit set
struct Set256 {
    uint64_t w[4] = {0, 0, 0, 0};
    constexpr Set256(const char* s) {
        for (; *s; ++s) {
            unsigned c = (unsigned char)*s;
            w[c >> 6] |= 1ull << (c & 63);
        }
    }
    bool has(uint8_t c) const { return (w[c >> 6] >> (c & 63)) & 1; }
};
const Set256 S_DNA("acgtACGT -.nN"), S_RNA("acguACGU -.nN"), S_DNAR("acgtryswkmbdhvACGTRYSWKMBDHV -.nN"),
    S_RNAR("acguryswkmbdhvACGURYSWKMBDHV -.nN"),
    S_PROT("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ -xX*_.");
const Set256* set_of(Alphabet a) {
    switch (a) {
        case AB_DNA: return &S_DNA;
        case AB_RNA: return &S_RNA;
        case AB_DNAredundant: return &S_DNAR;
        case AB_RNAredundant: return &S_RNAR;
        case AB_PROTEIN: return &S_PROT;
        default: return nullptr;
    }
}
}  // namespace

bool alphabet_valid_letters(Alphabet a, const uint8_t* s, size_t n) {
    const Set256* set = set_of(a);
    if (!set) return true;
    for (size_t i = 0; i < n; ++i)
        if (!set->has(s[i])) return false;
    return true;
}

Alphabet guess_alphabet_less_conservatively(const uint8_t* s, size_t n, int64_t thr) {
    if (n == 0) return AB_UNLIMIT;
    if (thr != 0 && (int64_t)n > thr) n = (size_t)thr;
    // one pass: which letters occur
    Set256 seen("");
    for (size_t i = 0; i < n; ++i) seen.w[s[i] >> 6] |= 1ull << (s[i] & 63);
    auto subset = [&](const Set256& big) {
        for (int k = 0; k < 4; ++k)
            if (seen.w[k] & ~big.w[k]) return false;
        return true;
    };
    if (subset(S_DNA)) return AB_DNAredundant;  // "less conservatively": DNA -> DNAredundant
    if (subset(S_RNA)) return AB_RNAredundant;
    if (subset(S_DNAR)) return AB_DNAredundant;
    if (subset(S_RNAR)) return AB_RNAredundant;
    if (subset(S_PROT)) return AB_PROTEIN;
    return AB_UNLIMIT;
}

int quality_offset(const std::string& enc) {
    std::string s = lower(enc);
    if (s == "sanger" || s == "illumina-1.8+") return 33;
    if (s == "solexa" || s == "illumina-1.3+" || s == "illumina-1.5+") return 64;
    if (s.empty()) return 0;
    throw OptError("unsupported quality encoding: " + enc +
                   ". available values: 'sanger', 'solexa', 'illumina-1.3+', 'illumina-1.5+', 'illumina-1.8+'");
}

}  // namespace bsk
