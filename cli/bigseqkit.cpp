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
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/sendfile.h>
#include <sys/stat.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <atomic>
#include <future>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <chrono>

#include "../include/bsk.h"
#include "../bigseqkit_amd/csrc/json.hpp"

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
    {"plan", 0, BOOL, "", "false"},      // one JSON object: operator, options, files, output place (python -m bigseqkit_amd.run)
    {"devices", 0, STR, "", ""},         // "0,1,2,3" / "0-7": one worker THREAD per GPU in this process, collectives over librccl
                                         // (run_devices below); a device named twice shares its GPU (tests)
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
    {"fq2fa", "Fq2Fa", {}},                                                          // cli/fq2fa.go:27
    {"range", "Range", {{"range", 'r', STR, "Range", ""}}},                            // cli/range.go:48
    {"head", "Head", {{"number", 'n', INT, "N", "10"}}},                               // cli/head.go:40
    {"duplicate", "Duplicate", {{"times", 'n', INT, "Times", "1"}}},                   // cli/duplicate.go:28-40 (alias dup)
    {"concat", "Concat", {{"full", 'f', BOOL, "Full", "false"}, {"separator", 's', STR, "Separator", "|"}}},   // cli/concat.go:51-52
    {"common", "Common",                                                               // cli/common.go:52-56
     {{"by-name", 'n', BOOL, "ByName", "false"}, {"by-seq", 's', BOOL, "BySeq", "false"},
      {"ignore-case", 'i', BOOL, "IgnoreCase", "false"}, {"only-positive-strand", 'P', BOOL, "OnlyPositiveStrand", "false"}}},
    {"pair", "Pair",                                                                   // cli/pair.go (flags of its init())
     {{"read1", '1', STR, "", ""}, {"read2", '2', STR, "", ""}, {"out-dir", 'O', STR, "", ""}, {"force", 'f', BOOL, "", "false"},
      {"save-unpaired", 'u', BOOL, "SaveUnpaired", "false"}}},
    {"rename", "Rename", {{"by-name", 'n', BOOL, "ByName", "false"}}},                  // cli/rename.go
    {"faidx", "Faidx",                                                                 // cli/faidx.go:68-72
     {{"use-regexp", 'r', BOOL, "UseRegexp", "false"}, {"ignore-case", 'i', BOOL, "IgnoreCase", "false"},
      {"full-head", 'f', BOOL, "FullHead", "false"}, {"region-file", 'l', STR, "RegionFile", ""},
      {"index-file", 'd', STR, "", ""}}},
    {"sort", "Sort",                                                                   // cli/sort.go:50-61
     {{"natural-order", 'N', BOOL, "InNaturalOrder", "false"}, {"by-name", 'n', BOOL, "ByName", "false"},
      {"by-seq", 's', BOOL, "BySeq", "false"}, {"by-length", 'l', BOOL, "ByLength", "false"},
      {"by-bases", 'b', BOOL, "ByBases", "false"}, {"gap-letters", 'G', STR, "GapLetters", "- \t."},
      {"reverse", 'r', BOOL, "Reverse", "false"}, {"ignore-case", 'i', BOOL, "IgnoreCase", "false"},
      {"two-pass", '2', BOOL, "", "false"}, {"keep-temp", 'k', BOOL, "", "false"},
      {"seq-prefix-length", 'L', INT, "SeqPrefixLength", "10000"}}},
    {"rmdup", "RmDup",
     {{"by-name", 'n', BOOL, "ByName", "false"}, {"by-seq", 's', BOOL, "BySeq", "false"},
      {"ignore-case", 'i', BOOL, "IgnoreCase", "false"}, {"dup-seqs-file", 'd', STR, "DupSeqsFile", ""},
      {"dup-num-file", 'D', STR, "DupNumFile", ""}, {"only-positive-strand", 'P', BOOL, "OnlyPositiveStrand", "false"}}},
};

// The process ends here, without atexit handlers / static destructors: everything the command wrote is flushed and closed
// by then, and the teardown of the HIP runtime at exit() is not ours to wait for -- it took one CLI run in ~200 down with
// SIGSEGV after the output was complete (tests/test_cli.py on the MI355X box, round 3).
[[noreturn]] void leave(int rc) {
    std::cout.flush();
    std::cerr.flush();
    fflush(nullptr);
    _exit(rc);
}

[[noreturn]] void die(const std::string& m) {
    std::cerr << "Error: " << m << "\n";
    leave(1);
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
    // (one read() loop into the string: the stream-buffer copy this replaced moved a file of 8 GB twice)
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) die("open " + path + ": no such file or directory");
    struct stat sb;
    std::string text;
    if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) {
        text.resize((size_t)sb.st_size);
        size_t done = 0;
        while (done < text.size()) {
            const ssize_t got = read(fd, &text[done], std::min<size_t>(text.size() - done, (size_t)256 << 20));
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) break;
            done += (size_t)got;
        }
        text.resize(done);
    } else {  // a pipe, a character device: until it ends
        char buf[1 << 16];
        for (;;) {
            const ssize_t got = read(fd, buf, sizeof buf);
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) break;
            text.append(buf, (size_t)got);
        }
    }
    close(fd);
    return text;
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

// one parsed command line: `args[0]` is the command, the rest flags and input files
struct Invocation {
    const Command* cmd = nullptr;
    Values val;
    std::vector<std::string> files;
    std::string js;  // option JSON of the operator
    std::string pget(const char* name) const {
        auto it = val.scalar.find(name);
        if (it != val.scalar.end()) return it->second;
        for (auto& f : kPersistent)
            if (!strcmp(f.name, name)) return f.def;
        if (cmd)
            for (auto& f : cmd->flags)
                if (!strcmp(f.name, name)) return f.def;
        return "";
    }
};

Invocation parse_invocation(const std::vector<std::string>& args) {
    Invocation inv;
    const int argc = (int)args.size();
    auto argv = [&](int i) -> const std::string& { return args[(size_t)i]; };
    for (auto& c : kCommands)
        if (args[0] == c.use || (args[0] == "dup" && !strcmp(c.use, "duplicate"))) inv.cmd = &c;
    if (!inv.cmd) die(std::string("unknown command \"") + args[0] + "\" for \"bigseqkit\"");
    const Command* cmd = inv.cmd;
    Values& val = inv.val;
    std::vector<std::string>& files = inv.files;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv(i);
        if (a == "--") { for (int k = i + 1; k < argc; ++k) files.push_back(argv(k)); break; }
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
                v = argv(++i);
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
    std::string& js = inv.js;
    js = "{\"Config\":{";
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
    return inv;
}

// ---------------------------------------------------------------------------
// execution: inputs are PARTS (the reference's dataframe partitions): the text of a file on the host, or the
// device-resident output of an upstream command of a `pipe` job (it never leaves HBM)
// ---------------------------------------------------------------------------
struct Part {
    int fmt = BSK_FORMAT_FASTA;
    std::string host;        // file text (when dptr == nullptr)
    void* dptr = nullptr;    // device text of `n` bytes
    size_t n = 0;
    bsk_ctx* owner = nullptr;  // context whose output buffer dptr is (destroyed with the part)
    bool owned_alloc = false;  // dptr came from bsk_device_alloc
    // a file that does not fit the GPU (round 6; the reference takes any file size through its partitions,
    // bigseqkit/helper.go:148-178): its read-only mapping -- `stats` streams it through the library's double-buffered
    // host-shard path (bsk_stats_run with on_device = 0: record-aligned 256 MiB chunks, H2D of chunk i + 1 under the kernels of i)
    const void* map = nullptr;
    size_t map_n = 0;
    const void* ptr() const { return dptr ? dptr : (map ? map : (const void*)host.data()); }
    size_t size() const { return dptr ? n : (map ? map_n : host.size()); }
    int on_device() const { return dptr ? 1 : 0; }
};

void release(std::vector<Part>& parts) {
    for (auto& p : parts) {
        if (p.owner) bsk_destroy(p.owner);
        else if (p.owned_alloc && p.dptr) bsk_device_free(p.dptr);
        if (p.map) munmap(const_cast<void*>(p.map), p.map_n);
        p.owner = nullptr; p.dptr = nullptr; p.map = nullptr;
    }
    parts.clear();
}

struct Output {
    std::vector<Part> parts;  // FASTA / FASTQ records (device-resident when keep_on_device)
    std::string text;         // everything else: stats table, grep -C count, locate rows, or the host copy of the records
    bool is_text = false;     // `text` is the result (not record text)
    int fmt = -1;             // format of the record text in `text` (-1: line oriented)
};

bool g_faidx_query = false;  // `faidx` with regions: records out, not index rows

int run_op(const std::string& use, bsk_ctx* ctx, const Part& in, int64_t pid, uint64_t first_record, bsk_out* out) {
    const void* p = in.ptr();
    const size_t n = in.size();
    const int dev = in.on_device();
    if (use == "seq") return bsk_seq_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
    if (use == "grep") return bsk_grep_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
    if (use == "locate") return bsk_locate_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
    if (use == "subseq") return bsk_subseq_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
    if (use == "translate") return bsk_translate_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
    if (use == "fq2fa") return bsk_fq2fa_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
    if (use == "rename") return bsk_rename_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
    if (use == "sort") return bsk_sort_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
    if (use == "faidx" && g_faidx_query) return bsk_faidx_query_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
    if (use == "faidx") return bsk_faidx_run(ctx, p, n, dev, in.fmt, pid, first_record /* = byte offset here */, nullptr, out);
    if (use == "duplicate") return bsk_duplicate_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
    if (use == "range" || use == "head") return bsk_range_run(ctx, p, n, dev, in.fmt, pid, first_record, nullptr, out);
    return bsk_rmdup_run(ctx, p, n, dev, in.fmt, pid, nullptr, out);
}

// keep_on_device: the caller is an inner node of a pipe job; record outputs stay in HBM (one context per part)
Output execute(const Invocation& inv, std::vector<Part>& inputs, bool keep_on_device) {
    const std::string use = inv.cmd->use;
    const std::string& js = inv.js;
    const int device = (int)strtol(inv.pget("device").c_str(), nullptr, 10);
    Output res;
    // rmdup is global over the union of its inputs (bigseqkit/rmdup.go:97 groups the whole dataframe): one shard
    if ((use == "rmdup" || use == "rename" || use == "sort") && inputs.size() > 1) {
        size_t total = 0;
        for (auto& p : inputs) {
            if (p.fmt != inputs[0].fmt) die(use + ": inputs of different formats");
            total += p.size();
        }
        Part all;
        all.fmt = inputs[0].fmt;
        all.dptr = bsk_device_alloc(total + inputs.size() + 1);  // room for a newline after every part
        if (!all.dptr) die(bsk_global_error());
        all.owned_alloc = true;
        size_t at = 0;
        for (auto& p : inputs) {
            if (!p.size()) continue;
            if (bsk_device_copy((char*)all.dptr + at, p.ptr(), p.size(), p.on_device() ? BSK_COPY_D2D : BSK_COPY_H2D) != BSK_OK)
                die(bsk_global_error());
            at += p.size();
            // a file that does not end in a newline must not run into the next file's first header
            char last = 0;
            if (bsk_device_copy(&last, (char*)all.dptr + at - 1, 1, BSK_COPY_D2H) != BSK_OK) die(bsk_global_error());
            if (last != '\n') {
                const char nl = '\n';
                if (bsk_device_copy((char*)all.dptr + at, &nl, 1, BSK_COPY_H2D) != BSK_OK) die(bsk_global_error());
                ++at;
            }
        }
        all.n = at;
        release(inputs);
        inputs.push_back(all);
    }
    if (use == "stats") {
        std::string head, body;
        for (size_t fi = 0; fi < inputs.size(); ++fi) {
            const Part& in = inputs[fi];
            bsk_ctx* sc = nullptr;  // one Stats per input (cli/stats.go:16-21)
            if (bsk_create("Stats", js.c_str(), device, &sc) != BSK_OK) die(bsk_global_error());
            if (bsk_stats_run(sc, in.ptr(), in.size(), in.on_device(), in.fmt, 0, nullptr, nullptr) != BSK_OK) die(bsk_last_error(sc));
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
            head = table.substr(0, nl + 1);
            body += table.substr(nl + 1) + "\n";  // Join(lines[1:]) + "\n": a blank line per input, as written
            bsk_destroy(sc);
        }
        res.is_text = true;
        res.text = head + body;
        return res;
    }
    const bool grep_count = use == "grep" && inv.pget("count") == "true";
    const bool rows_out = use == "locate" || (use == "faidx" && !g_faidx_query);  // line-oriented text, not records
    const bool records_out = !(rows_out || grep_count);
    uint64_t grep_total = 0;
    bsk_ctx* ctx = nullptr;
    auto fresh = [&]() {
        bsk_ctx* c = nullptr;
        if (bsk_create(inv.cmd->op, js.c_str(), device, &c) != BSK_OK) die(bsk_global_error());
        return c;
    };
    if (!(keep_on_device && records_out)) {
        ctx = fresh();
        // the result goes to the host right after the run, while the input is still there: the operators that can leave their
        // text as ordered slices (rmdup, seq -n, subseq -r: include/bsk.h bsk_out.d_seg_*) do, and bsk_out_to_host gathers them
        bsk_ctx_set(ctx, "out", "slices");
    }
    // range / head: the record index runs over the union of the inputs (cli/helper.go unions the files into one
    // dataframe); negative positions need the total (bigseqkit/range.go:69-80)
    std::vector<uint64_t> first(inputs.size(), 0);
    uint64_t n_records = 0;
    bool range_needs = false;
    if (use == "range" || use == "head") {
        bsk_ctx* probe = fresh();
        int needs = 0;
        bsk_range_needs_count(probe, &needs);
        range_needs = needs != 0;
        if (range_needs || inputs.size() > 1)
            for (size_t fi = 0; fi < inputs.size(); ++fi) {
                uint64_t k = 0;
                if (bsk_index_build(probe, inputs[fi].ptr(), inputs[fi].size(), inputs[fi].on_device(), inputs[fi].fmt, nullptr, &k) != BSK_OK)
                    die(bsk_last_error(probe));
                first[fi] = n_records;
                n_records += k;
            }
        bsk_destroy(probe);
        if (ctx && range_needs && bsk_range_set_count(ctx, n_records) != BSK_OK) die(bsk_last_error(ctx));
    }
    for (size_t fi = 0; fi < inputs.size(); ++fi) {
        const Part& in = inputs[fi];
        bsk_out out;
        bsk_ctx* c = ctx ? ctx : fresh();
        if (!ctx && range_needs && bsk_range_set_count(c, n_records) != BSK_OK) die(bsk_last_error(c));
        if (run_op(use, c, in, (int64_t)fi, first[fi], &out) != BSK_OK) die(bsk_last_error(c));
        if (use == "rmdup" && bsk_rmdup_finish(c) != BSK_OK) die(bsk_last_error(c));
        if (grep_count) {
            uint64_t cnt = 0;
            bsk_grep_last_count(c, &cnt);
            grep_total += cnt;
            continue;
        }
        const int ofmt = use == "translate" || use == "fq2fa" || use == "faidx" ? BSK_FORMAT_FASTA : in.fmt;
        if (!ctx) {  // device-resident part owned by its context
            Part o;
            o.fmt = ofmt;
            o.dptr = out.d_data;
            o.n = out.len;
            o.owner = c;
            res.parts.push_back(o);
            continue;
        }
        res.fmt = rows_out ? -1 : ofmt;
        const size_t at = res.text.size();
        res.text.resize(at + out.len);
        if (bsk_out_to_host(c, &out, res.text.data() + at, out.len) != BSK_OK) die(bsk_last_error(c));
    }
    if (ctx) bsk_destroy(ctx);
    if (grep_count) { res.is_text = true; res.text = std::to_string(grep_total); }  // fmt.Print: no newline (cli/grep.go:14)
    if (rows_out) res.is_text = keep_on_device;  // rows cannot feed another command
    return res;
}

// device >= 0: the files go straight to that GPU (bsk_shard_load: several readers, pread || H2D) -- what every command but
// the ones that join their inputs on the host (concat, common, pair) takes; device < 0: host strings
// files of at least this many bytes do not go to the GPU as one shard: `stats` streams them from the host, the other
// commands say what to do.  BSK_HOST_PIPELINE_FROM moves the line (tests: 0)
static size_t whole_file_limit() {
    size_t lim = (size_t)200 << 30;  // (288 GB of HBM: the shard, the stats vector and nothing else)
    if (const char* e = getenv("BSK_HOST_PIPELINE_FROM")) lim = (size_t)strtoull(e, nullptr, 10);
    return lim;
}

std::vector<Part> read_parts(const std::vector<std::string>& files, int device = -1, const std::string& use = "") {
    std::vector<Part> parts;
    for (auto& f : files) {
        Part p;
        if (device < 0) {
            p.host = read_file(f);
            p.fmt = sniff_format(f, p.host);
        } else {
            const int fd = open(f.c_str(), O_RDONLY);
            if (fd < 0) die("open " + f + ": no such file or directory");
            struct stat sb;
            if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {  // (a pipe: read to its end first)
                close(fd);
                p.host = read_file(f);
                p.fmt = sniff_format(f, p.host);
                parts.push_back(std::move(p));
                continue;
            }
            char first = 0;
            const ssize_t got = pread(fd, &first, 1, 0);
            p.fmt = sniff_format(f, got == 1 ? std::string(1, first) : std::string());
            p.n = (size_t)sb.st_size;
            const bool too_big = use == "stats" && p.n > 0 && p.n >= whole_file_limit();
            if (too_big || bsk_shard_load(fd, 0, p.n, device, 0, &p.dptr) != BSK_OK) {
                p.dptr = nullptr;
                if (use != "stats") {
                    close(fd);
                    die(std::string(bsk_global_error()) + " -- '" + use + "': " + f + " (" + std::to_string(p.n) + " bytes) must fit one GPU next to its "
                        "result; cut it over several GPUs (--devices 0-7: fq2fa, grep, locate, rmdup, seq, stats, subseq, translate)");
                }
                // stats: the file stays on the host, its mapping is streamed in record-aligned chunks
                p.map = mmap(nullptr, p.n, PROT_READ, MAP_SHARED, fd, 0);
                if (p.map == MAP_FAILED) die("mmap " + f + " failed");
                p.map_n = p.n;
                p.n = 0;
                close(fd);
                parts.push_back(std::move(p));
                continue;
            }
            p.owned_alloc = true;
            close(fd);
        }
        parts.push_back(std::move(p));
    }
    return parts;
}

// `pipe` job (bigseqkit-cli/pipe.go:12-40): {"pipe": [job, ...], "cmd": ["grep", "-s", ...]} -- the outputs of the
// jobs under "pipe" become inputs of "cmd", ahead of the files named in it.  ("sh" hooks are not run.)
Output run_job(const bsk::json::Value& j, bool root) {
    std::vector<Part> inputs;
    if (const auto* deps = j.get("pipe")) {
        if (deps->kind == bsk::json::Value::Array)
            for (auto& d : deps->arr) {
                Output o = run_job(*d, false);
                if (o.is_text) die("bad execution dependency");  // pipe.go:26-28: the job produced no dataframe
                for (auto& p : o.parts) inputs.push_back(p);
            }
    }
    const auto* cmd = j.get("cmd");
    if (!cmd || cmd->kind != bsk::json::Value::Array || cmd->arr.empty()) die("incorrect job format");
    std::vector<std::string> args;
    for (auto& a : cmd->arr) args.push_back(a->str);
    Invocation inv = parse_invocation(args);
    for (auto& p : read_parts(inv.files, (int)strtol(inv.pget("device").c_str(), nullptr, 10), inv.cmd->use)) inputs.push_back(std::move(p));
    if (inputs.empty()) die("no input for job command " + args[0]);
    Output o = execute(inv, inputs, !root);
    release(inputs);
    return o;
}

void store(const Invocation& inv, const Output& out, const std::vector<std::string>& files) {
    if (out.is_text) { std::cout << out.text; return; }
    const std::string& out_text = out.text;
    // ---- cli/helper.go:105-127
    std::string outp = inv.pget("out-file");
    if (outp.empty()) outp = files.size() == 1 ? files[0] + "-out" : (getenv("IGNIS_JOB_NAME") ? std::string(getenv("IGNIS_JOB_NAME")) + "-out" : std::string());
    if (outp.empty()) die("out file -o required");
    if (outp == "-") { fwrite(out_text.data(), 1, out_text.size(), stdout); return; }
    if (inv.pget("merge") == "true") {  // StoreFASTX: one file
        std::ofstream f(outp, std::ios::binary);
        if (!f) die("cannot create " + outp);
        f.write(out_text.data(), (std::streamsize)out_text.size());
        f.close();  // (die() ends the process without destructors: nothing may still sit in a stream buffer by then)
        if (!f) die("write " + outp + " failed");
        return;
    }
    // StoreFASTXN == SaveAsTextFile: a directory of part files; a record is never split between two parts
    long parts = strtol(inv.pget("partitions").c_str(), nullptr, 10);
    if (parts < 1) parts = 1;
    if (!is_dir(outp) && mkdir(outp.c_str(), 0755) != 0) die("cannot create directory " + outp);
    size_t pos = 0;
    for (long p = 0; p < parts; ++p) {
        size_t end = p + 1 == parts ? out_text.size() : out_text.size() * (size_t)(p + 1) / (size_t)parts;
        if (end < pos) end = pos;
        if (p + 1 < parts && end < out_text.size()) {
            if (out.fmt >= 0) {  // next record start (the anchor rule of the library)
                size_t cut = out_text.size();
                if (bsk_find_record_start((const uint8_t*)out_text.data(), out_text.size(), end, out.fmt, &cut) == BSK_OK) end = cut;
                else end = out_text.size();
            } else {
                while (end < out_text.size() && end > 0 && out_text[end - 1] != '\n') ++end;  // line granularity
            }
        }
        char nm[64];
        snprintf(nm, sizeof nm, "/part%05ld", p);
        std::ofstream f(outp + nm, std::ios::binary);
        f.write(out_text.data() + pos, (std::streamsize)(end - pos));
        f.close();
        if (!f) die("write " + outp + nm + " failed");
        pos = end;
    }
}


// ---------------------------------------------------------------------------
// several GPUs, in THIS process (round 5): one worker thread + one context per device, the collectives of the command
// behind the C ABI (bsk_comm_*: librccl, ncclCommInitAll) -- what ignisDriver does with executors
// (bigseqkit-cli/helper.go:87-141; ReadFASTA/Q[N] + StoreFASTX[N], bigseqkit/helper.go:148-195).  No Python in the process
// tree (until round 4 this flag exec'd `python -m bigseqkit_amd.run`, which stays as the test harness of the Python side).
//   * the FILE is cut, not copied: it is mapped, every cut is the first record start at or behind size * k / world, found in
//     a 1 MiB window of the mapping (bsk_find_record_start: the ReadFixer rule; the window grows while it ends on a record
//     that may be cut off), and every worker brings only its own byte range to its GPU (bsk_shard_load: several readers,
//     pread || H2D in 16 MiB pieces);
//   * seq / grep / locate / subseq / translate / fq2fa: one call over the device-resident shard + bsk_store_put (D2H ||
//     write), or for shards over 48 GB bsk_run_to_store (pinned shard; H2D || kernels || D2H + write in chunks), into
//     <out>/part%05d, one per worker (StoreFASTXN) or, with --merge, into ONE file at offsets from an all-gather of the
//     sizes (the reference passes an MPI token, bigseqkit-lib/helper.go:399-429);
//   * stats: bsk_stats_collect_reduced (StatsReduce: one ncclAllReduce of the stats vector), worker 0 prints the table;
//   * grep -C: bsk_count_allreduce;   rmdup: bsk_rmdup_dist_run (the 24-byte tuple exchange of GroupByKey).
// ---------------------------------------------------------------------------
// BSK_CLI_TIMING=1: where the wall clock of a --devices call goes (stderr)
void mark(const char* what, int rank = -1) {
    static const bool on = getenv("BSK_CLI_TIMING") != nullptr;
    static const auto t0 = std::chrono::steady_clock::now();
    if (!on) return;
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rank < 0) fprintf(stderr, "[timing] %8.3f s  %s\n", s, what);
    else fprintf(stderr, "[timing] %8.3f s  worker %d: %s\n", s, rank, what);
}

std::vector<int> parse_devices(const std::string& text) {
    std::vector<int> out;
    size_t i = 0;
    while (i <= text.size()) {
        const size_t e = std::min(text.find(',', i), text.size());
        const std::string piece = text.substr(i, e - i);
        if (!piece.empty()) {
            const size_t dash = piece.find('-', 1);
            char* end = nullptr;
            if (dash == std::string::npos) {
                const long v = strtol(piece.c_str(), &end, 10);
                if (*end || v < 0) die("--devices: bad device '" + piece + "'");
                out.push_back((int)v);
            } else {
                const long a = strtol(piece.substr(0, dash).c_str(), &end, 10), b = strtol(piece.substr(dash + 1).c_str(), nullptr, 10);
                if (a < 0 || b < a || b - a > 63) die("--devices: bad range '" + piece + "'");
                for (long v = a; v <= b; ++v) out.push_back((int)v);
            }
        }
        i = e + 1;
    }
    if (out.empty()) die("--devices names no device");
    if (out.size() > 64) die("--devices: at most 64 workers");
    return out;
}

std::vector<size_t> cut_points(const uint8_t* map, size_t size, int world, int fmt) {
    std::vector<size_t> cuts{0};
    for (int k = 1; k < world; ++k) {
        const size_t nominal = (size_t)((unsigned __int128)size * (unsigned)k / (unsigned)world);
        const size_t lo = std::max(nominal, cuts.back());
        if (lo >= size) { cuts.push_back(size); continue; }
        // the window: up to `win` bytes before `lo` (a FASTQ start is also judged by the record that ends there: anchor.hpp)
        // and `win` bytes behind it.  A start is taken when the next larger window names the same one -- a candidate that
        // was passed over because the window cut its record off shows up as a different answer -- and does not
        // sit within 64 KiB of the window's end.
        size_t win = 1u << 20;
        auto look = [&](size_t w, size_t* b_out) {
            const size_t a = lo > w ? lo - w : 0;
            const size_t b = std::min(size, lo + w);
            size_t found = 0;
            if (bsk_find_record_start(map + a, b - a, lo - a, fmt, &found) != BSK_OK) die(bsk_global_error());
            *b_out = b;
            return found + a;
        };
        for (;;) {
            size_t b = 0, b4 = 0;
            const size_t found = look(win, &b);
            if (b < size && (found + (64u << 10) > b || look(win * 4, &b4) != found)) { win *= 4; continue; }
            cuts.push_back(std::min(found, size));
            break;
        }
    }
    cuts.push_back(size);
    return cuts;
}

struct Worker {
    std::string error;            // why this worker gave up ("" = fine)
    uint64_t out_bytes = 0, out_records = 0;
    std::string spool;            // --merge / -o -: this worker's part before it is put in place
    std::string text;             // worker 0: what goes to stdout (stats table, grep -C count)
};

// len bytes of the file at `off` into buf with several readers (a copy out of the page cache is one core's work: ~9 GB/s with
// one thread on the box of round 6, where PCIe takes 55)
static bool parallel_pread(int fd, uint8_t* buf, size_t len, size_t off, int threads = 8) {
    if (len < ((size_t)64 << 20)) threads = 1;
    std::vector<std::future<bool>> jobs;
    const size_t per = (len + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; ++t) {
        const size_t a = std::min(len, per * (size_t)t), b = std::min(len, a + per);
        if (a >= b) break;
        jobs.push_back(std::async(std::launch::async, [=]() {
            size_t done = a;
            while (done < b) {
                const ssize_t got = pread(fd, buf + done, std::min<size_t>(b - done, 64u << 20), (off_t)(off + done));
                if (got <= 0) return false;
                done += (size_t)got;
            }
            return true;
        }));
    }
    bool ok = true;
    for (auto& j : jobs) ok = j.get() && ok;
    return ok;
}

int run_devices(const Invocation& inv) {
    const std::string use = inv.cmd->use;
    static const char* const kStreamed[] = {"seq", "grep", "locate", "subseq", "translate", "fq2fa"};
    bool streamed = false;
    for (const char* u : kStreamed) streamed = streamed || use == u;
    if (!streamed && use != "stats" && use != "rmdup")
        die("'" + use + "' runs on one device (bigseqkit " + use + " ... --device N); several GPUs: fq2fa, grep, locate, rmdup, seq, stats, subseq, translate");
    if (inv.files.size() != 1) die("--devices: exactly one input file (it is cut into one shard per GPU)");
    const std::vector<int> devices = parse_devices(inv.pget("devices"));
    const int world = (int)devices.size();
    mark("start");
    if (bsk_device_count() <= 0) die("no HIP device visible (the hot path has no CPU fallback)");
    const std::string& path = inv.files[0];
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) die("open " + path + ": no such file or directory");
    struct stat sb;
    if (fstat(fd, &sb) != 0) die("stat " + path + " failed");
    const size_t size = (size_t)sb.st_size;
    const uint8_t* map = nullptr;
    if (size) {
        map = (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) die("mmap " + path + " failed");
    }
    const int fmt = sniff_format(path, size ? std::string((const char*)map, 1) : std::string());
    const std::vector<size_t> cuts = size ? cut_points(map, size, world, fmt) : std::vector<size_t>((size_t)world + 1, 0);
    if (map) munmap((void*)map, size);
    mark("device count + cut points");
    // RCCL greets on file descriptor 1 (its version banner, when a communicator is first used) -- and stdout is where
    // `stats`, `grep -C` and `-o -` put their RESULT.  For the rest of this call descriptor 1 IS stderr; results go to the
    // real stdout through a duplicate of it.
    fflush(stdout);
    const int real_out = dup(1);
    if (real_out < 0 || dup2(2, 1) < 0) die("cannot duplicate the standard output");
    std::vector<bsk_comm*> comms((size_t)world, nullptr);
    if (bsk_comm_init_all(world, devices.data(), comms.data()) != BSK_OK) die(bsk_comm_error(nullptr));
    mark("communicators");

    std::string out_file = inv.pget("out-file");
    if (out_file.empty()) out_file = path + "-out";
    const bool merge = inv.pget("merge") == "true", to_stdout = out_file == "-";
    const bool grep_count = use == "grep" && inv.pget("count") == "true";
    const bool records = !(use == "stats" || grep_count);
    bsk_store* dir_store = nullptr;
    char token[64];
    snprintf(token, sizeof token, "%d-%08x", (int)getpid(), (unsigned)((uintptr_t)&token ^ (uintptr_t)time(nullptr) * 2654435761u));
    if (records && !merge && !to_stdout) {
        if (!is_dir(out_file) && mkdir(out_file.c_str(), 0755) != 0) die("cannot create directory " + out_file);
        // (part files of an earlier run with MORE workers would be read as part of this result: ADVICE r04)
        for (int k = world; k < world + 4096; ++k) {
            char nm[64];
            snprintf(nm, sizeof nm, "/part%05d", k);
            if (unlink((out_file + nm).c_str()) != 0) break;
        }
        if (bsk_store_open(out_file.c_str(), 0, &dir_store) != BSK_OK) die("cannot open the directory " + out_file);
    }
    std::vector<Worker> W((size_t)world);
    std::atomic<int> failed{0};
    // shard + output + tables of a record operator next to each other in 288 GB: shards up to 48 GB (translate -f 6 writes
    // twice its input); BSK_HOST_PIPELINE_FROM moves the line (tests: 0 = always the chunked host pipeline)
    size_t host_pipeline_from = (size_t)48 << 30;
    if (const char* e = getenv("BSK_HOST_PIPELINE_FROM")) host_pipeline_from = (size_t)strtoull(e, nullptr, 10);

    auto work = [&](int rank) {
        Worker& me = W[(size_t)rank];
        bsk_comm* comm = comms[(size_t)rank];
        const int device = devices[(size_t)rank];
        const size_t lo = cuts[(size_t)rank], n = cuts[(size_t)rank + 1] - lo;
        bsk_ctx* ctx = nullptr;
        void* d_shard = nullptr;
        void* h = nullptr;
        // stats: a shard that does not fit (or whose allocation fails) is streamed from the file in pinned pieces (below)
        const bool stats_stream = use == "stats" && n > 0 && n >= std::min(host_pipeline_from * 4, whole_file_limit());
        const bool via_host = streamed && n > host_pipeline_from;
        bsk_store* own = nullptr;
        bool stream_stats = stats_stream;
        auto give_up = [&](const std::string& m) { if (me.error.empty()) me.error = m.empty() ? "failed" : m; failed.fetch_add(1); };
        do {
            if (bsk_create(inv.cmd->op, inv.js.c_str(), device, &ctx) != BSK_OK) { give_up(bsk_global_error()); break; }
            bsk_ctx_set(ctx, "out", "slices");  // (the shard stays on the device until bsk_store_put has drained the result)
            mark("context", rank);
            // this worker's bytes, and only they: from the file to the device in pieces, several readers (bsk_shard_load).
            // A record operator whose shard AND output may not fit the GPU side by side keeps the chunked host pipeline
            // (bsk_run_to_store: pinned shard, 256 MiB chunks through two device buffers).
            if (stats_stream) break;
            if (!via_host) {
                if (bsk_shard_load(fd, (uint64_t)lo, n, device, 0, &d_shard) != BSK_OK) {
                    if (use == "stats") { d_shard = nullptr; stream_stats = true; break; }  // (no room: streamed below)
                    if (use == "rmdup")
                        give_up(std::string(bsk_global_error()) + " -- rmdup needs every worker's shard (" + std::to_string(n) +
                                " bytes here) in HBM next to its tables: more devices make smaller shards");
                    else give_up(bsk_global_error());
                    break;
                }
                mark("shard on the device", rank);
                break;
            }
            // (round 6: the shard is NOT read into one pinned buffer first -- 100 GB took 12 s to pin and 10 s to read with one
            // thread before the first kernel ran: grep on the whole C2 file 29.5 s, bench.py end_to_end_config_size.  It is
            // streamed below in pieces through two pinned buffers, the pread of piece k + 1 under the pipeline of piece k.)
        } while (false);
        // every worker enters every collective, also the one that has given up (its contribution is empty): the others must
        // not wait for it for ever
        const bool ok0 = me.error.empty();
        if (use == "stats") {
            std::vector<int64_t> keys(1 << 20), vals(1 << 20);
            size_t cnt = 0;
            int rc = ok0 ? bsk_stats_reset(ctx, nullptr) : BSK_ERR_INVALID_ARG;
            if (rc == BSK_OK && n && !stream_stats) rc = bsk_stats_run(ctx, d_shard, n, 1, fmt, rank, nullptr, nullptr);
            if (rc == BSK_OK && n && stream_stats) {
                // Round 6: a shard larger than the GPU's memory (the reference takes any file size through its partitions,
                // bigseqkit/helper.go:148-178).  Pieces of ~1 GiB that end on record starts are read into two pinned buffers
                // -- the pread of piece i + 1 runs while piece i crosses PCIe under the kernels (bsk_stats_run with a host
                // pointer: record-aligned chunks through two device buffers) -- and accumulate into the one stats vector.
                const size_t piece = std::max<size_t>(1 << 16, getenv("BSK_STATS_PIECE_BYTES") ? (size_t)strtoull(getenv("BSK_STATS_PIECE_BYTES"), nullptr, 10) : ((size_t)1 << 30));
                const uint8_t* fmap = (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);  // (only the pages around the cuts are touched)
                uint8_t* pin[2] = {(uint8_t*)bsk_host_alloc(piece + (64 << 20)), (uint8_t*)bsk_host_alloc(piece + (64 << 20))};
                if (fmap == MAP_FAILED || !pin[0] || !pin[1]) { rc = BSK_ERR_HIP; give_up("stats: no pinned memory / mapping for the streamed shard"); }
                else {
                    std::vector<size_t> pc{lo};
                    while (pc.back() < lo + n) {
                        size_t nxt = lo + n;
                        if (lo + n - pc.back() > piece) {
                            size_t at = 0;
                            if (bsk_find_record_start(fmap, lo + n, pc.back() + piece, fmt, &at) == BSK_OK && at > pc.back() && at < lo + n && at - pc.back() <= piece + (64u << 20)) nxt = at;
                        }
                        pc.push_back(nxt);
                    }
                    auto read_piece = [&](size_t k) -> bool { return parallel_pread(fd, pin[k & 1], pc[k + 1] - pc[k], pc[k]); };
                    bool okr = read_piece(0);
                    for (size_t k = 0; k + 1 < pc.size() && okr && rc == BSK_OK; ++k) {
                        std::future<bool> next;
                        if (k + 2 < pc.size()) next = std::async(std::launch::async, read_piece, k + 1);
                        rc = bsk_stats_run(ctx, pin[k & 1], pc[k + 1] - pc[k], 0, fmt, (int64_t)rank * 1000000 + (int64_t)k, nullptr, nullptr);
                        if (next.valid()) okr = next.get();
                    }
                    if (!okr && rc == BSK_OK) { rc = BSK_ERR_INVALID_ARG; give_up("short read of " + path); }
                    mark("shard streamed from the file", rank);
                }
                if (fmap != MAP_FAILED) munmap((void*)fmap, size);
                for (uint8_t* q : pin) if (q) bsk_host_free(q);
            }
            if (ok0 && rc != BSK_OK) give_up(bsk_last_error(ctx));
            uint64_t bad = (uint64_t)failed.load();
            if (bsk_count_allreduce(comm, &bad, nullptr) != BSK_OK) { give_up(bsk_comm_error(comm)); return; }
            if (bad) return;  // (somebody failed before the reduction: nobody enters it)
            rc = bsk_stats_collect_reduced(ctx, comm, nullptr, nullptr, keys.data(), vals.data(), keys.size(), &cnt);
            if (rc != BSK_OK) { give_up(bsk_last_error(ctx)); }
            else if (rank == 0) {
                bsk_statinfo info;
                bsk_stats_finalize(ctx, keys.data(), vals.data(), cnt, &info);
                std::vector<char> buf(1 << 16);
                if (bsk_stats_string(ctx, "input0", "N/A", &info, buf.data(), buf.size()) != BSK_OK) give_up(bsk_last_error(ctx));
                else {
                    const std::string table = buf.data();
                    const size_t nl = table.find('\n');
                    me.text = table.substr(0, nl + 1) + table.substr(nl + 1) + "\n";  // (head + Join(lines[1:]) + "\n", as the single-device CLI)
                }
            }
        } else if (grep_count) {
            uint64_t cnt = 0;
            if (ok0) {
                bsk_out o;
                if (bsk_grep_run(ctx, d_shard, n, 1, fmt, rank, nullptr, &o) != BSK_OK || bsk_grep_last_count(ctx, &cnt) != BSK_OK) give_up(bsk_last_error(ctx));
            }
            uint64_t bad = (uint64_t)failed.load();
            if (bsk_count_allreduce(comm, &bad, nullptr) != BSK_OK) { give_up(bsk_comm_error(comm)); return; }
            if (!bad) {
                if (bsk_count_allreduce(comm, &cnt, nullptr) != BSK_OK) give_up(bsk_comm_error(comm));
                else if (rank == 0) me.text = std::to_string(cnt);  // fmt.Print: no newline (bigseqkit-cli/grep.go:14)
            }
        } else {
            bsk_store* st = dir_store;
            uint64_t part = (uint64_t)rank;
            if (ok0 && !dir_store) {  // --merge / -o -: this worker's part goes to a spool file of its own first
                char nm[96];
                snprintf(nm, sizeof nm, ".bsk-%s-part%05d.tmp", token, rank);
                me.spool = (to_stdout ? std::string(getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp") + "/bsk-stdout" : out_file) + nm;
                const int tfd = open(me.spool.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0600);  // (never through a planted link)
                if (tfd < 0) give_up("cannot create " + me.spool);
                else {
                    close(tfd);
                    if (bsk_store_open(me.spool.c_str(), 1, &own) != BSK_OK) give_up("cannot create " + me.spool);
                    st = own;
                    part = 0;
                }
            }
            if (use == "rmdup") {
                bsk_out o{nullptr, 0, 0};
                uint64_t bad = (uint64_t)failed.load();
                if (bsk_count_allreduce(comm, &bad, nullptr) != BSK_OK) { give_up(bsk_comm_error(comm)); bad = 1; }
                if (!bad) {
                    if (bsk_rmdup_dist_run(ctx, comm, d_shard, n, fmt, nullptr, &o) != BSK_OK) give_up(bsk_last_error(ctx));
                    else if (bsk_store_put(st, ctx, part, &o) != BSK_OK) give_up(bsk_store_error(st));
                    me.out_bytes = o.len; me.out_records = o.records;
                }
            } else if (me.error.empty() && via_host) {
                // A shard that does not fit the GPU next to its result: pieces of ~2 GiB that end on record starts (found on a
                // mapping of the file: only the pages around the cuts are touched) go through bsk_run_to_store one after the
                // other -- inside, 256 MiB chunks cross PCIe under the kernels and the drain of the chunk before -- while a
                // reader thread fills the other pinned buffer.  The pieces are the parts 0, 1, 2 ... of ONE file: this worker's
                // spool (--merge / -o -) or its part%05d of the directory, written through a store of its own.
                bsk_store* mine = own;
                uint64_t p0 = 0;
                if (!mine) {
                    char nm[64];
                    snprintf(nm, sizeof nm, "/part%05d", rank);
                    if (bsk_store_open((out_file + nm).c_str(), 1, &mine) != BSK_OK) give_up("cannot create " + out_file + nm);
                }
                const size_t piece = std::max<size_t>(1 << 16, getenv("BSK_STREAM_PIECE_BYTES") ? (size_t)strtoull(getenv("BSK_STREAM_PIECE_BYTES"), nullptr, 10) : ((size_t)2 << 30));
                const uint8_t* fmap = (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
                uint8_t* pin[2] = {(uint8_t*)bsk_host_alloc(piece + (64 << 20)), (uint8_t*)bsk_host_alloc(piece + (64 << 20))};
                if (me.error.empty() && (fmap == MAP_FAILED || !pin[0] || !pin[1])) give_up("no pinned memory / mapping for the streamed shard");
                if (me.error.empty()) {
                    bsk_ctx_set(ctx, "pin_alphabet", "1");  // (one partition through several calls: one alphabet guess)
                    std::vector<size_t> pc{lo};
                    while (pc.back() < lo + n) {
                        size_t nxt = lo + n;
                        if (lo + n - pc.back() > piece) {
                            size_t at = 0;
                            if (bsk_find_record_start(fmap, lo + n, pc.back() + piece, fmt, &at) == BSK_OK && at > pc.back() && at < lo + n && at - pc.back() <= piece + (64u << 20)) nxt = at;
                        }
                        pc.push_back(nxt);
                    }
                    auto read_piece = [&](size_t k) -> bool { return parallel_pread(fd, pin[k & 1], pc[k + 1] - pc[k], pc[k]); };
                    bool okr = read_piece(0);
                    for (size_t k = 0; k + 1 < pc.size() && okr && me.error.empty(); ++k) {
                        std::future<bool> next;
                        if (k + 2 < pc.size()) next = std::async(std::launch::async, read_piece, k + 1);
                        uint64_t ob = 0, orec = 0;
                        // (pid: the header row of `locate` belongs to the first chunk of partition 0 -- the first piece of worker 0 only)
                        const int64_t pid = (rank == 0 && k == 0) ? 0 : (int64_t)rank * 1000000 + (int64_t)k + 1;
                        if (bsk_run_to_store(ctx, pin[k & 1], pc[k + 1] - pc[k], fmt, pid, mine, p0 + k, &ob, &orec) != BSK_OK) give_up(bsk_last_error(ctx));
                        me.out_bytes += ob; me.out_records += orec;
                        if (next.valid()) okr = next.get();
                    }
                    if (!okr) give_up("short read of " + path);
                    mark("shard streamed through the host pipeline", rank);
                }
                if (fmap != MAP_FAILED) munmap((void*)fmap, size);
                for (uint8_t* q : pin) if (q) bsk_host_free(q);
                if (mine && mine != own) { uint64_t tot = 0; if (bsk_store_close(mine, &tot) != BSK_OK) give_up("closing this worker's part failed"); }
            } else if (me.error.empty()) {
                // the shard is on the device: one call over all of it, its output drained in pieces (bsk_store_put: the
                // copy of a piece runs while the piece before it is written)
                Part in;
                in.fmt = fmt; in.dptr = d_shard; in.n = n;
                bsk_out o{nullptr, 0, 0};
                if (n && run_op(use, ctx, in, rank, 0, &o) != BSK_OK) give_up(bsk_last_error(ctx));
                else if (bsk_store_put(st, ctx, part, &o) != BSK_OK) give_up(bsk_store_error(st));
                me.out_bytes = o.len; me.out_records = o.records;
            }
            if (own) {
                uint64_t tot = 0;
                if (bsk_store_close(own, &tot) != BSK_OK) give_up("closing " + me.spool + " failed");
            }
        }
        mark("command done", rank);
        if (d_shard) bsk_device_free(d_shard);
        if (h) bsk_host_free(h);
        if (ctx) bsk_destroy(ctx);
        mark("released", rank);
    };
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r) threads.emplace_back(work, r);
    for (auto& t : threads) t.join();
    for (auto* c : comms) bsk_comm_destroy(c);
    mark("communicators destroyed");
    close(fd);
    uint64_t tot = 0;
    if (dir_store && bsk_store_close(dir_store, &tot) != BSK_OK) { W[0].error = "closing the output failed"; failed.fetch_add(1); }
    auto drop_spools = [&] { for (auto& w : W) if (!w.spool.empty()) unlink(w.spool.c_str()); };
    if (failed.load()) {
        drop_spools();
        for (int r = 0; r < world; ++r)
            if (!W[(size_t)r].error.empty()) die("worker " + std::to_string(r) + " (device " + std::to_string(devices[(size_t)r]) + "): " + W[(size_t)r].error);
        die("a worker failed");
    }
    if (!records) {
        size_t w = 0;
        while (w < W[0].text.size()) {
            const ssize_t k = write(real_out, W[0].text.data() + w, W[0].text.size() - w);
            if (k <= 0) die("write to the standard output failed");
            w += (size_t)k;
        }
        return 0;
    }
    if (dir_store) {
        for (int r = 0; r < world; ++r)  // (an empty part file still marks the partition, as SaveAsTextFile does)
            if (W[(size_t)r].out_bytes == 0) {
                char nm[64];
                snprintf(nm, sizeof nm, "/part%05d", r);
                // (O_TRUNC: a non-empty part of an earlier run into the same directory must not pass for this run's -- ADVICE r05)
                const int e = open((out_file + nm).c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
                if (e >= 0) close(e);
            }
        return 0;
    }
    // ---- one file (or stdout): the parts in rank order == file order (FileStore's order)
    const int dst = to_stdout ? real_out : open(out_file.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (dst < 0) { drop_spools(); die("cannot create " + out_file); }
    bool okw = true;
    std::vector<char> buf(8u << 20);
    for (int r = 0; r < world && okw; ++r) {
        const int src = open(W[(size_t)r].spool.c_str(), O_RDONLY);
        if (src < 0) { okw = false; break; }
        for (;;) {
            const ssize_t k = read(src, buf.data(), buf.size());
            if (k < 0) { okw = false; break; }
            if (k == 0) break;
            ssize_t w = 0;
            while (w < k) { const ssize_t x = write(dst, buf.data() + w, (size_t)(k - w)); if (x <= 0) { okw = false; break; } w += x; }
            if (!okw) break;
        }
        close(src);
    }
    if (!to_stdout && close(dst) != 0) okw = false;
    drop_spools();
    if (!okw) die("write " + out_file + " failed");
    return 0;
}

}  // namespace

static int run_main(int argc, char** argv) {
    if (argc < 2 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) { usage(); return argc < 2; }
    std::vector<std::string> args(argv + 1, argv + argc);
    if (args[0] == "pipe") {
        // bigseqkit pipe --job job.json [files...] [-o out] [--merge] ...   (bigseqkit-cli/pipe.go:42-67)
        std::string job;
        std::vector<std::string> rest{"seq"};  // the persistent flags are parsed with any command table
        for (size_t i = 1; i < args.size(); ++i) {
            if (args[i] == "--job" && i + 1 < args.size()) job = args[++i];
            else if (args[i].rfind("--job=", 0) == 0) job = args[i].substr(6);
            else rest.push_back(args[i]);
        }
        if (job.empty()) die("job not defined");
        Invocation inv = parse_invocation(rest);
        bsk::json::ValuePtr j;
        try {
            const std::string text = read_file(job);
            j = bsk::json::Parser(text).parse();
        } catch (const std::exception&) { die("incorrect job format"); }
        Output o = run_job(*j, true);
        // the files given to `pipe` itself are appended unchanged (pipe.go:61-66)
        for (auto& p : read_parts(inv.files)) o.text += p.host;
        store(inv, o, inv.files);
        return 0;
    }
    Invocation inv = parse_invocation(args);
    bool faidx_query = false;
    if (std::string(inv.cmd->use) == "faidx") {
        // cli/faidx.go: the first argument is the file, the others are regions ("id", "id:b-e", ...)
        std::string regions;
        for (size_t k = 1; k < inv.files.size(); ++k) regions += (k > 1 ? "," : "") + jquote(inv.files[k]);
        if (inv.files.size() > 1) inv.files.resize(1);
        inv.js.insert(inv.js.rfind('}'), ",\"Regions\":[" + regions + "]");
        faidx_query = !regions.empty() || !inv.pget("region-file").empty();
    }
    if (inv.pget("dry-run") == "true") {
        std::cout << inv.cmd->op << "\n" << inv.js << "\n";
        for (auto& f : inv.files) std::cout << f << "\n";
        return 0;
    }
    if (inv.pget("plan") == "true") {
        // what the multi-GPU launcher needs to know, decided by THIS program's flag tables (no second parser in Python)
        std::string files;
        for (size_t k = 0; k < inv.files.size(); ++k) files += (k ? "," : "") + jquote(inv.files[k]);
        std::cout << "{\"use\":" << jquote(inv.cmd->use) << ",\"op\":" << jquote(inv.cmd->op) << ",\"opts\":" << inv.js
                  << ",\"files\":[" << files << "],\"out_file\":" << jquote(inv.pget("out-file"))
                  << ",\"merge\":" << (inv.pget("merge") == "true" ? "true" : "false")
                  << ",\"partitions\":" << strtol(inv.pget("partitions").c_str(), nullptr, 10) << "}\n";
        return 0;
    }
    if (!inv.pget("devices").empty()) return run_devices(inv);  // several GPUs: worker threads + librccl, in this process
    if (inv.files.empty()) die("no input files (stdin is not supported by the IgnisHPC CLI either)");
    g_faidx_query = faidx_query;
    const std::string use_cmd = inv.cmd->use;
    const bool joins_on_host = use_cmd == "concat" || use_cmd == "common" || use_cmd == "pair";
    if (!joins_on_host && bsk_device_count() <= 0) die("no HIP device visible (the hot path has no CPU fallback)");
    std::vector<Part> inputs = read_parts(inv.files, joins_on_host ? -1 : (int)strtol(inv.pget("device").c_str(), nullptr, 10), use_cmd);
    if (std::string(inv.cmd->use) == "concat") {
        if (inputs.size() != 2) die("2 files needed");
        if (inputs[0].fmt != inputs[1].fmt) die("concat: inputs of different formats");
        std::string both = inputs[0].host;
        if (!both.empty() && both.back() != '\n') both += '\n';
        const size_t n_first = both.size();
        both += inputs[1].host;
        bsk_ctx* c = nullptr;
        const int device = (int)strtol(inv.pget("device").c_str(), nullptr, 10);
        if (bsk_create("Concat", inv.js.c_str(), device, &c) != BSK_OK) die(bsk_global_error());
        bsk_out out;
        if (bsk_concat_run(c, both.data(), both.size(), n_first, 0, inputs[0].fmt, nullptr, &out) != BSK_OK) die(bsk_last_error(c));
        Output o;
        o.fmt = inputs[0].fmt;
        o.text.resize(out.len);
        if (out.len && bsk_out_to_host(c, &out, &o.text[0], out.len) != BSK_OK) die(bsk_last_error(c));
        bsk_destroy(c);
        store(inv, o, inv.files);
        return 0;
    }
    if (std::string(inv.cmd->use) == "common") {
        // cli/common.go: at least two files; the records of the first one that are common to all
        if (inputs.size() < 2) die("at least 2 files needed");
        std::string all;
        std::vector<uint64_t> ends;
        for (auto& p : inputs) {
            if (p.fmt != inputs[0].fmt) die("common: inputs of different formats");
            all += p.host;
            if (!p.host.empty() && p.host.back() != '\n') all += '\n';
            ends.push_back(all.size());
        }
        bsk_ctx* c = nullptr;
        const int device = (int)strtol(inv.pget("device").c_str(), nullptr, 10);
        if (bsk_create("Common", inv.js.c_str(), device, &c) != BSK_OK) die(bsk_global_error());
        bsk_out out;
        if (bsk_common_run(c, all.data(), all.size(), ends.data(), (uint32_t)ends.size(), 0, inputs[0].fmt, nullptr, &out) != BSK_OK)
            die(bsk_last_error(c));
        Output o;
        o.fmt = inputs[0].fmt;
        o.text.resize(out.len);
        if (out.len && bsk_out_to_host(c, &out, &o.text[0], out.len) != BSK_OK) die(bsk_last_error(c));
        bsk_destroy(c);
        store(inv, o, inv.files);
        return 0;
    }
    if (std::string(inv.cmd->use) == "pair") {
        // cli/pair.go:13-66: two files; <out-dir>/paired.1, paired.2 and, with -u, unpaired.1, unpaired.2
        if (inputs.size() < 2) die("2 files needed");
        const std::string outdir = inv.pget("out-dir");
        if (outdir.empty()) die("out-dir required");
        if (inputs[0].fmt != inputs[1].fmt) die("pair: inputs of different formats");
        if (!is_dir(outdir) && mkdir(outdir.c_str(), 0755) != 0) die("cannot create directory " + outdir);
        std::string both = inputs[0].host;
        if (!both.empty() && both.back() != '\n') both += '\n';
        const size_t n_first = both.size();
        both += inputs[1].host;
        bsk_ctx* c = nullptr;
        const int device = (int)strtol(inv.pget("device").c_str(), nullptr, 10);
        if (bsk_create("Pair", inv.js.c_str(), device, &c) != BSK_OK) die(bsk_global_error());
        bsk_out outs[4];
        if (bsk_pair_run(c, both.data(), both.size(), n_first, 0, inputs[0].fmt, nullptr, outs) != BSK_OK) die(bsk_last_error(c));
        const char* names[4] = {"paired.1", "paired.2", "unpaired.1", "unpaired.2"};
        const bool unp = inv.pget("save-unpaired") == "true";
        for (int k = 0; k < (unp ? 4 : 2); ++k) {
            std::string text(outs[k].len, '\0');
            if (outs[k].len && bsk_out_to_host(c, &outs[k], &text[0], text.size()) != BSK_OK) die(bsk_last_error(c));
            std::ofstream f(outdir + "/" + names[k], std::ios::binary);
            if (!f) die("cannot create " + outdir + "/" + names[k]);
            f.write(text.data(), (std::streamsize)text.size());
            f.close();
            if (!f) die("write " + outdir + "/" + names[k] + " failed");
        }
        bsk_destroy(c);
        return 0;
    }
    Output o = execute(inv, inputs, false);
    store(inv, o, inv.files);
    return 0;
}

// a crash names its frames on stderr (the tests show stderr when a run fails) instead of leaving a bare signal number
static void on_crash(int sig) {
    static const char msg[] = "bigseqkit: fatal signal, frames:\n";
    if (write(2, msg, sizeof msg - 1) < 0) {}
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char** argv) {
    signal(SIGSEGV, on_crash);
    signal(SIGBUS, on_crash);
    signal(SIGABRT, on_crash);
    leave(run_main(argc, argv));
}
