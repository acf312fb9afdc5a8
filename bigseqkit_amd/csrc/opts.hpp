// Option structs of the hot-path commands, kept schema-compatible with the
// reference's Go structs (JSON field name == Go field name, nested "Config").
//   KitConfig         /root/reference/bigseqkit/helper.go:29-39, defaults :86-103
//   SeqOptions        bigseqkit/seq.go:9-30,       defaults :32-55
//   StatsOptions      bigseqkit/stats.go:18-26,    defaults :28-38
//   GrepOptions       bigseqkit/grep.go:13-29,     defaults :31-49
//   LocateOptions     bigseqkit/locate.go:9-25,    defaults :27-45
//   SubseqOptions     bigseqkit/subseq.go:9-20,    defaults :22-35
//   TranslateOptions  bigseqkit/translate.go:9-20, defaults :22-35
//   RmDupOptions      bigseqkit/rmdup.go:13-21,    defaults :23-33
//   Fq2FaOptions      bigseqkit/fq2fa.go:11-18;  RangeOptions bigseqkit/range.go:14-24;  HeadOptions bigseqkit/head.go:12-22;
//   DuplicateOptions  bigseqkit/duplicate.go:9-19
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace bsk {

struct OptError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

enum class FieldType { Bool, Int, Float, String, StringList };

struct Field {
    std::string name;
    FieldType type;
    bool is_null = false;  // ChunkSize / BufferSize stay null (helper.go:88-89)
    bool b = false;
    int64_t i = 0;
    double f = 0;
    std::string s;
    std::vector<std::string> sl;
};

enum class Op { Stats, Seq, Grep, Locate, Subseq, Translate, RmDup, Fq2Fa, Range, Head, Duplicate, Rename, Sort, Faidx, Pair, Common, Concat };

class Options {
   public:
    Op op;
    std::vector<Field> config;  // KitConfig, in Go declaration order
    std::vector<Field> fields;  // command fields, in Go declaration order

    // defaults (== setDefaults()) for `op`
    static Options defaults(Op op);
    // StringToOptions + setDefaults: missing / null fields take the default
    static Options from_json(Op op, const std::string& text);
    // OptionsToString: Go encoding/json text incl. the trailing '\n'
    std::string to_json() const;

    bool b(const char* n) const { return find(n).b; }
    int64_t i(const char* n) const { return find(n).i; }
    double f(const char* n) const { return find(n).f; }
    const std::string& s(const char* n) const { return find(n).s; }
    const std::vector<std::string>& sl(const char* n) const { return find(n).sl; }
    Field& mut(const char* n) { return const_cast<Field&>(find(n)); }

    bool cb(const char* n) const { return cfind(n).b; }
    int64_t ci(const char* n) const { return cfind(n).i; }
    const std::string& cs(const char* n) const { return cfind(n).s; }
    Field& cmut(const char* n) { return const_cast<Field&>(cfind(n)); }

   private:
    const Field& find(const char* n) const;
    const Field& cfind(const char* n) const;
};

// op name as used in the reference's libSource("...") strings
bool op_from_name(const std::string& name, Op* out);
const char* op_name(Op op);

// sequence alphabets (what KitConfig.GetAlphabet yields; helper.go:68-84)
enum Alphabet { AB_NONE = 0, AB_DNA, AB_DNAredundant, AB_RNA, AB_RNAredundant, AB_PROTEIN, AB_UNLIMIT };
Alphabet alphabet_from_seqtype(const std::string& t);  // throws OptError
const char* alphabet_name(Alphabet a);
// seq.GuessAlphabetLessConservatively [shenwei356/bio v0.7.0, not in tree]
Alphabet guess_alphabet_less_conservatively(const uint8_t* s, size_t n, int64_t thr);
bool alphabet_valid_letters(Alphabet a, const uint8_t* s, size_t n);
// parseQualityEncoding + Offset (bigseqkit-lib/helper.go:119-136)
int quality_offset(const std::string& enc);  // throws OptError

}  // namespace bsk
