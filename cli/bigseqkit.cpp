// bigseqkit -- seqkit-compatible command line for the seven hot-path commands, on libbsk.so.
//
// Mirrors the cobra CLI of the reference (flag names, shorthands and defaults):
//   persistent flags   /root/reference/bigseqkit-cli/helper.go:161-173
//   seq / stats / grep / locate / subseq / translate / rmdup flag tables
//                      bigseqkit-cli/{seq.go:54-73, stats.go:61-65, grep.go:81-96, locate.go:61-74,
//                                     subseq.go:56-67, translate.go:86-94, rmdup.go:45-51}
//   input sniffing     bigseqkit-cli/helper.go:63-78   (extension, then first byte)
//   output naming      bigseqkit-cli/helper.go:105-127 (<file>-out, --merge: one file, else a directory of parts)
// Where the reference builds an IgnisHPC dataflow (ignisDriver, helper.go:87-132) this program
// builds the option JSON of the command (same schema as bigseqkit.OptionsToString) and drives the
// C ABI of include/bsk.h: one bsk_create, one bsk_<op>_run per input file (partition), bsk_destroy.
// Extras of this CLI: --device N, --dry-run (print the operator name and option JSON, no GPU needed),
// and "-o -" for standard output.
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../include/bsk.h"

namespace {

enum Kind { BOOL, INT, FLOAT, STR, SLICE };
struct Flag {
    const char* name;  // long name
    char shorthand;    // 0 = none
    Kind kind;
    const char* field;  // JSON field ("" = CLI-only), prefixed with "Config." for KitConfig
    const char* def;    // textual default
};

const Flag kPersistent[] = {
    {"seq-type", 't', STR, "Config.SeqType", "auto"},
    {"line-width", 'w', INT, "Config.LineWidth", "60"},
    {"id-regexp", 0, STR, "Config.IDRegexp", "^(\\S+)\\s?"},
    {"id-ncbi", 0, BOOL, "Config.IDNCBI", "false"},
    {"out-file", 'o', STR, "", ""},
    {"quiet", 0, BOOL, "Config.Quiet", "false"},
    {"alphabet-guess-seq-length", 0, INT, "Config.AlphabetGuessSeqLength", "10000"},
    {"infile-list", 0, STR, "", ""},
    {"merge", 0, BOOL, "", "false"},
    {"partitions", 0, INT, "", "0"},
    {"order", 0, BOOL, "", "false"},
    {"device", 0, INT, "", "0"},
    {"dry-run", 0, BOOL, "", "false"},
};

struct Command {
    const char* use;
    const char* op;  // operator name for bsk_create
    std::vector<Flag> flags;
};

const Command kCommands[] = {
    {"seq", "SeqTransform",
     {{"reverse", 'r', BOOL, "Reverse", "false"}, {"complement", 'p', BOOL, "Complement", "false"},
      {"name", 'n', BOOL, "Name", "false"}, {"seq", 's', BOOL, "Seq", "false"}, {"qual", 'q', BOOL, "Qual", "false"},
      {"only-id", 'i', BOOL, "OnlyId", "false"}, {"remove-gaps", 'g', BOOL, "RemoveGaps", "false"},
      {"gap-letters", 'G', STR, "GapLetters", "- \t."}, {"lower-case", 'l', BOOL, "LowerCase", "false"},
      {"upper-case", 'u', BOOL, "UpperCase", "false"}, {"dna2rna", 0, BOOL, "Dna2rna", "false"},
      {"rna2dna", 0, BOOL, "Rna2dna", "false"}, {"color", 'k', BOOL, "", "false"},
      {"validate-seq", 'v', BOOL, "ValidateSeq", "false"}, {"validate-seq-length", 'V', INT, "ValidateSeqLength", "10000"},
      {"min-len", 'm', INT, "MinLen", "-1"}, {"max-len", 'M', INT, "MaxLen", "-1"},
      {"qual-ascii-base", 'b', INT, "QualAsciiBase", "33"}, {"min-qual", 'Q', FLOAT, "MinQual", "-1"},
      {"max-qual", 'R', FLOAT, "MaxQual", "-1"}}},
    {"stats", "Stats",
     {{"tabular", 'T', BOOL, "Tabular", "false"}, {"gap-letters", 'G', STR, "GapLetters", "- ."},
      {"all", 'a', BOOL, "All", "false"}, {"skip-err", 'e', BOOL, "SkipErr", "false"},
      {"fq-encoding", 'E', STR, "FqEncoding", "sanger"}}},
    {"grep", "Grep",
     {{"pattern", 'p', SLICE, "Pattern", ""}, {"pattern-file", 'f', STR, "PatternFile", ""},
      {"use-regexp", 'r', BOOL, "UseRegexp", "false"}, {"delete-matched", 0, BOOL, "DeleteMatched", "false"},
      {"invert-match", 'v', BOOL, "InvertMatch", "false"}, {"by-name", 'n', BOOL, "ByName", "false"},
      {"by-seq", 's', BOOL, "BySeq", "false"}, {"only-positive-strand", 'P', BOOL, "OnlyPositiveStrand", "false"},
      {"max-mismatch", 'm', INT, "MaxMismatch", "0"}, {"ignore-case", 'i', BOOL, "IgnoreCase", "false"},
      {"degenerate", 'd', BOOL, "Degenerate", "false"}, {"region", 'R', STR, "Region", ""},
      {"circular", 'c', BOOL, "Circular", "false"}, {"immediate-output", 'I', BOOL, "", "false"},
      {"count", 'C', BOOL, "Count", "false"}}},
    {"locate", "Locate",
     {{"pattern", 'p', SLICE, "Pattern", ""}, {"pattern-file", 'f', STR, "PatternFile", ""},
      {"degenerate", 'd', BOOL, "Degenerate", "false"}, {"use-regexp", 'r', BOOL, "UseRegexp", "false"},
      {"use-fmi", 'F', BOOL, "UseFmi", "false"}, {"ignore-case", 'i', BOOL, "IgnoreCase", "false"},
      {"only-positive-strand", 'P', BOOL, "OnlyPositiveStrand", "false"},
      {"validate-seq-length", 'V', INT, "ValidateSeqLength", "10000"}, {"non-greedy", 'G', BOOL, "NonGreedy", "false"},
      {"gtf", 0, BOOL, "Gtf", "false"}, {"bed", 0, BOOL, "Bed", "false"}, {"max-mismatch", 'm', INT, "MaxMismatch", "0"},
      {"hide-matched", 'M', BOOL, "HideMatched", "false"}, {"circular", 'c', BOOL, "Circular", "false"}}},
    {"subseq", "SubseqTransform",
     {{"chr", 0, SLICE, "Chr", ""}, {"region", 'r', STR, "Region", ""}, {"gtf", 0, STR, "Gtf", ""},
      {"feature", 0, SLICE, "Feature", ""}, {"up-stream", 'u', INT, "UpStream", "0"},
      {"down-stream", 'd', INT, "DownStream", "0"}, {"only-flank", 'f', BOOL, "OnlyFlank", "false"},
      {"bed", 0, STR, "Bed", ""}, {"gtf-tag", 0, STR, "GtfTag", "gene_id"}}},
    {"translate", "Translate",
     {{"transl-table", 'T', INT, "TranslTable", "1"}, {"frame", 'f', SLICE, "Frame", "1"},
      {"trim", 0, BOOL, "Trim", "false"}, {"clean", 0, BOOL, "Clean", "false"},
      {"allow-unknown-codon", 'x', BOOL, "AllowUnknownCodon", "false"}, {"init-codon-as-M", 'M', BOOL, "InitCodonAsM", "false"},
      {"list-transl-table", 'l', INT, "ListTranslTable", "-1"},
      {"list-transl-table-with-amb-codons", 'L', INT, "ListTranslTableWithAmbCodons", "-1"},
      {"append-frame", 'F', BOOL, "AppendFrame", "false"}}},
    {"rmdup", "RmDup",
     {{"by-name", 'n', BOOL, "ByName", "false"}, {"by-seq", 's', BOOL, "BySeq", "false"},
      {"ignore-case", 'i', BOOL, "IgnoreCase", "false"}, {"dup-seqs-file", 'd', STR, "DupSeqsFile", ""},
      {"dup-num-file", 'D', STR, "DupNumFile", ""}, {"only-positive-strand", 'P', BOOL, "OnlyPositiveStrand", "false"}}},
};

[[noreturn]] void die(const std::string& m) {
    std::cerr << "Error: " << m << "\n";
    exit(1);
}

std::string jquote(const std::string& s) {
    std::string o = "\"";
    char b[8];
    for (unsigned char c : s) {
        if (c == '"') o += "\\\"";
        else if (c == '\\') o += "\\\\";
        else if (c == '\n') o += "\\n";
        else if (c == '\t') o += "\\t";
        else if (c < 0x20 || c == '<' || c == '>' || c == '&') { snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o.push_back((char)c);
    }
    return o + "\"";
}

// pflag StringSlice: comma separated, double quotes protect commas
std::vector<std::string> split_csv(const std::string& v) {
    std::vector<std::string> out;
    std::string cur;
    bool q = false;
    for (size_t i = 0; i < v.size(); ++i) {
        char c = v[i];
        if (c == '"') {
            if (q && i + 1 < v.size() && v[i + 1] == '"') { cur.push_back('"'); ++i; }
            else q = !q;
        } else if (c == ',' && !q) { out.push_back(cur); cur.clear(); }
        else cur.push_back(c);
    }
    out.push_back(cur);
    return out;
}

struct Values {
    std::map<std::string, std::string> scalar;            // flag name -> text
    std::map<std::string, std::vector<std::string>> slice;  // flag name -> values (only when given)
};

const Flag* find_flag(const Command& c, const std::string& name, char sh) {
    for (auto& f : c.flags)
        if ((!name.empty() && name == f.name) || (sh && sh == f.shorthand)) return &f;
    for (auto& f : kPersistent)
        if ((!name.empty() && name == f.name) || (sh && sh == f.shorthand)) return &f;
    return nullptr;
}

bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
bool is_dir(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }

bool has_suffix(std::string s, const std::vector<const char*>& es) {
    for (auto& c : s) c = (char)tolower((unsigned char)c);
    for (auto e : es) {
        size_t n = strlen(e);
        if (s.size() >= n && s.compare(s.size() - n, n, e) == 0) return true;
    }
    return false;
}

std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) die("open " + path + ": no such file or directory");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// bigseqkit-cli/helper.go:63-78
int sniff_format(const std::string& file, const std::string& data) {
    if (has_suffix(file, {".fa", ".fna", ".ffn", ".faa", ".frn"})) return BSK_FORMAT_FASTA;
    if (has_suffix(file, {".fq", ".fastq"})) return BSK_FORMAT_FASTQ;
    if (!data.empty() && data[0] == '>') return BSK_FORMAT_FASTA;
    if (!data.empty() && data[0] == '@') return BSK_FORMAT_FASTQ;
    die(" <file> must be fasta or fastq");
}

void usage() {
    std::cout << "bigseqkit -- MI355X-native seqkit-compatible commands (libbsk)\n\nUsage:\n  bigseqkit <command> [flags] files...\n\nCommands:\n";
    for (auto& c : kCommands) std::cout << "  " << c.use << "\n";
    std::cout << "\nGlobal flags: -t/--seq-type -w/--line-width --id-regexp --id-ncbi -o/--out-file --quiet\n"
                 "  --alphabet-guess-seq-length --infile-list --merge --partitions --order --device --dry-run\n";
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) { usage(); return argc < 2; }
    const Command* cmd = nullptr;
    for (auto& c : kCommands)
        if (!strcmp(argv[1], c.use)) cmd = &c;
    if (!cmd) die(std::string("unknown command \"") + argv[1] + "\" for \"bigseqkit\"");

    Values val;
    std::vector<std::string> files;
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--") { for (int k = i + 1; k < argc; ++k) files.push_back(argv[k]); break; }
        if (a.size() < 2 || a[0] != '-' || a == "-") { files.push_back(a); continue; }
        std::vector<std::pair<const Flag*, std::string>> parsed;  // flag, inline value ("\x01" = none)
        if (a[1] == '-') {
            std::string name = a.substr(2), v = "\x01";
            size_t eq = name.find('=');
            if (eq != std::string::npos) { v = name.substr(eq + 1); name = name.substr(0, eq); }
            const Flag* f = find_flag(*cmd, name, 0);
            if (!f) die("unknown flag: --" + name);
            parsed.push_back({f, v});
        } else {
            for (size_t k = 1; k < a.size(); ++k) {
                const Flag* f = find_flag(*cmd, "", a[k]);
                if (!f) die(std::string("unknown shorthand flag: '") + a[k] + "' in " + a);
                if (f->kind != BOOL && k + 1 < a.size()) {  // -p=ACGT or -pACGT
                    std::string v = a.substr(k + 1);
                    if (!v.empty() && v[0] == '=') v.erase(0, 1);
                    parsed.push_back({f, v});
                    break;
                }
                parsed.push_back({f, "\x01"});
            }
        }
        for (auto& pr : parsed) {
            const Flag* f = pr.first;
            std::string v = pr.second;
            if (f->kind == BOOL) {
                if (v == "\x01") v = "true";
                if (v != "true" && v != "false") die(std::string("invalid argument \"") + v + "\" for \"--" + f->name + "\" flag");
                val.scalar[f->name] = v;
                continue;
            }
            if (v == "\x01") {
                if (i + 1 >= argc) die(std::string("flag needs an argument: --") + f->name);
                v = argv[++i];
            }
            if (f->kind == SLICE) {
                auto parts = split_csv(v);
                auto& dst = val.slice[f->name];
                dst.insert(dst.end(), parts.begin(), parts.end());
            } else {
                if (f->kind == INT) {
                    char* e = nullptr;
                    strtol(v.c_str(), &e, 10);
                    if (v.empty() || *e) die(std::string("invalid argument \"") + v + "\" for \"--" + f->name + "\" flag");
                }
                if (f->kind == FLOAT) {
                    char* e = nullptr;
                    strtod(v.c_str(), &e);
                    if (v.empty() || *e) die(std::string("invalid argument \"") + v + "\" for \"--" + f->name + "\" flag");
                }
                val.scalar[f->name] = v;
            }
        }
    }
    auto get = [&](const char* name, const Flag* table, size_t n) -> std::string {
        auto it = val.scalar.find(name);
        if (it != val.scalar.end()) return it->second;
        for (size_t k = 0; k < n; ++k)
            if (!strcmp(table[k].name, name)) return table[k].def;
        return "";
    };
    auto pget = [&](const char* name) { return get(name, kPersistent, sizeof(kPersistent) / sizeof(Flag)); };

    // getFlagPositiveInt / getFlagNonNegativeInt (bigseqkit-cli/helper.go:248-265), message as written
    for (const char* nm : {"line-width", "max-mismatch", "up-stream", "down-stream", "validate-seq-length", "alphabet-guess-seq-length",
                           "qual-ascii-base", "transl-table"}) {
        const Flag* f = find_flag(*cmd, nm, 0);
        if (!f) continue;
        auto it = val.scalar.find(nm);
        long v = strtol((it != val.scalar.end() ? it->second : std::string(f->def)).c_str(), nullptr, 10);
        const bool positive = !strcmp(nm, "qual-ascii-base") || !strcmp(nm, "transl-table");
        if (positive ? v <= 0 : v < 0) die(std::string("value of flag --") + nm + " should be greater than 0");
        if (!strcmp(nm, "validate-seq-length") && v > 0 && v < 1000)
            die("value of flag --validate-seq-length too small, should >= 1000");
    }
    // getIDRegexp (helper.go:297-309)
    if (val.scalar.count("id-ncbi") && val.scalar["id-ncbi"] == "true") val.scalar["id-regexp"] = "\\|([^\\|]+)\\| ";
    // bigseqkit-cli/helper.go:332-338
    {
        long g = strtol(pget("alphabet-guess-seq-length").c_str(), nullptr, 10);
        if (g > 0 && g < 1000) die("value of flag --alphabet-guess-seq-length too small, should >= 1000");
    }

    // ---- option JSON (schema of bigseqkit.OptionsToString).  Like the reference's parseSeqKit*Options,
    // every field carries the flag's value, i.e. the CLI default when the flag was not given.
    auto emit_field = [&](std::string& js, const Flag& f, const char* field) {
        js += jquote(field);
        js += ":";
        if (f.kind == SLICE) {
            auto it = val.slice.find(f.name);
            std::vector<std::string> v = it != val.slice.end() ? it->second : (*f.def ? split_csv(f.def) : std::vector<std::string>());
            js += "[";
            for (size_t k = 0; k < v.size(); ++k) js += (k ? "," : "") + jquote(v[k]);
            js += "]";
            return;
        }
        auto it = val.scalar.find(f.name);
        const std::string v = it != val.scalar.end() ? it->second : std::string(f.def);
        if (f.kind == STR) js += jquote(v);
        else js += v;
    };
    std::string js = "{\"Config\":{";
    bool first = true;
    for (auto& f : kPersistent) {
        if (strncmp(f.field, "Config.", 7)) continue;
        if (!first) js += ",";
        first = false;
        emit_field(js, f, f.field + 7);
    }
    js += "}";
    for (auto& f : cmd->flags) {
        if (!*f.field) continue;
        js += ",";
        emit_field(js, f, f.field);
    }
    js += "}";

    // input files (+ --infile-list, one per line)
    {
        std::string lst = pget("infile-list");
        if (!lst.empty()) {
            std::istringstream ss(read_file(lst));
            std::string line;
            while (std::getline(ss, line))
                if (!line.empty()) files.push_back(line);
        }
    }
    const bool dry = pget("dry-run") == "true";
    if (dry) {
        std::cout << cmd->op << "\n" << js << "\n";
        for (auto& f : files) std::cout << f << "\n";
        return 0;
    }
    if (files.empty()) die("no input files (stdin is not supported by the IgnisHPC CLI either)");

    const int device = (int)strtol(pget("device").c_str(), nullptr, 10);
    bsk_ctx* ctx = nullptr;
    if (bsk_create(cmd->op, js.c_str(), device, &ctx) != BSK_OK) die(bsk_global_error());

    const std::string use = cmd->use;
    std::string out_text;  // records of all inputs, in input order (union, cli/helper.go:134-141)
    std::string stats_head, stats_body;
    uint64_t grep_total = 0;
    const bool grep_count = use == "grep" && val.scalar.count("count") && val.scalar["count"] == "true";
    for (size_t fi = 0; fi < files.size(); ++fi) {
        const std::string data = read_file(files[fi]);
        const int fmt = sniff_format(files[fi], data);
        if (use == "stats") {
            bsk_ctx* sc = nullptr;  // one Stats per input (cli/stats.go:16-21)
            if (bsk_create("Stats", js.c_str(), device, &sc) != BSK_OK) die(bsk_global_error());
            if (bsk_stats_run(sc, data.data(), data.size(), 0, fmt, 0, nullptr, nullptr) != BSK_OK) die(bsk_last_error(sc));
            std::vector<int64_t> keys(1 << 20), vals(1 << 20);
            size_t n = 0;
            if (bsk_stats_collect(sc, nullptr, keys.data(), vals.data(), keys.size(), &n) != BSK_OK) die(bsk_last_error(sc));
            bsk_statinfo info;
            bsk_stats_finalize(sc, keys.data(), vals.data(), n, &info);
            std::vector<char> buf(1 << 16);
            std::string name = "input" + std::to_string(fi);
            if (bsk_stats_string(sc, name.c_str(), "N/A", &info, buf.data(), buf.size()) != BSK_OK) die(bsk_last_error(sc));
            std::string table = buf.data();
            size_t nl = table.find('\n');
            stats_head = table.substr(0, nl + 1);
            stats_body += table.substr(nl + 1) + "\n";  // Join(lines[1:]) + "\n": a blank line per input, as written
            bsk_destroy(sc);
            continue;
        }
        bsk_out out;
        int rc;
        const int64_t pid = (int64_t)fi;
        if (use == "seq") rc = bsk_seq_run(ctx, data.data(), data.size(), 0, fmt, pid, nullptr, &out);
        else if (use == "grep") rc = bsk_grep_run(ctx, data.data(), data.size(), 0, fmt, pid, nullptr, &out);
        else if (use == "locate") rc = bsk_locate_run(ctx, data.data(), data.size(), 0, fmt, pid, nullptr, &out);
        else if (use == "subseq") rc = bsk_subseq_run(ctx, data.data(), data.size(), 0, fmt, pid, nullptr, &out);
        else if (use == "translate") rc = bsk_translate_run(ctx, data.data(), data.size(), 0, fmt, pid, nullptr, &out);
        else rc = bsk_rmdup_run(ctx, data.data(), data.size(), 0, fmt, pid, nullptr, &out);
        if (rc != BSK_OK) die(bsk_last_error(ctx));
        if (grep_count) {
            uint64_t c = 0;
            bsk_grep_last_count(ctx, &c);
            grep_total += c;
            continue;
        }
        const size_t at = out_text.size();
        out_text.resize(at + out.len);
        if (bsk_out_to_host(ctx, &out, out_text.data() + at, out.len) != BSK_OK) die(bsk_last_error(ctx));
    }
    if (use == "rmdup" && bsk_rmdup_finish(ctx) != BSK_OK) die(bsk_last_error(ctx));
    bsk_destroy(ctx);
    if (use == "stats") { std::cout << stats_head << stats_body; return 0; }
    if (grep_count) { std::cout << grep_total; return 0; }  // fmt.Print: no newline (cli/grep.go:14)

    // ---- store (cli/helper.go:105-127)
    std::string outp = pget("out-file");
    if (outp.empty()) outp = files.size() == 1 ? files[0] + "-out" : (getenv("IGNIS_JOB_NAME") ? std::string(getenv("IGNIS_JOB_NAME")) + "-out" : std::string());
    if (outp.empty()) die("out file -o required");
    if (outp == "-") { fwrite(out_text.data(), 1, out_text.size(), stdout); return 0; }
    if (pget("merge") == "true") {  // StoreFASTX: one file
        std::ofstream f(outp, std::ios::binary);
        if (!f) die("cannot create " + outp);
        f.write(out_text.data(), (std::streamsize)out_text.size());
        return 0;
    }
    // StoreFASTXN == SaveAsTextFile: a directory of part files (records never split)
    long parts = strtol(pget("partitions").c_str(), nullptr, 10);
    if (parts < 1) parts = 1;
    if (!is_dir(outp) && mkdir(outp.c_str(), 0755) != 0) die("cannot create directory " + outp);
    size_t pos = 0;
    for (long p = 0; p < parts; ++p) {
        size_t end = p + 1 == parts ? out_text.size() : out_text.size() * (size_t)(p + 1) / (size_t)parts;
        while (end < out_text.size() && end > 0 && out_text[end - 1] != '\n') ++end;  // line granularity
        char nm[64];
        snprintf(nm, sizeof nm, "/part%05ld", p);
        std::ofstream f(outp + nm, std::ios::binary);
        f.write(out_text.data() + pos, (std::streamsize)(end - pos));
        pos = end;
    }
    return 0;
}
