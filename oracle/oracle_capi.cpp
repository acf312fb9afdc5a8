// TEST INFRASTRUCTURE -- C entry points of the CPU oracle (ctypes-friendly).
// See oracle_core.hpp for the parity status ("parity unpinned") and PARITY.md.
#include <cstdio>
#include <cstring>

#include "oracle_core.hpp"

using namespace orc;

extern "C" {

typedef struct {
    const char* SeqType;
    int LineWidth;
    const char* IDRegexp;
    int IDNCBI;
    int Quiet;
    int AlphabetGuessSeqLength;
    int ValidateSeqLength;
} orc_kitconfig;

typedef struct {
    orc_kitconfig Config;
    int Tabular;
    const char* GapLetters;
    int All;
    int SkipErr;
    const char* FqEncoding;
    int Basename;
} orc_stats_opts;

typedef struct {
    orc_kitconfig Config;
    int Reverse, Complement, Name, Seq, Qual, OnlyId, RemoveGaps;
    const char* GapLetters;
    int LowerCase, UpperCase, Dna2rna, Rna2dna, ValidateSeq, ValidateSeqLength, MaxLen, MinLen, QualAsciiBase;
    double MinQual, MaxQual;
} orc_seq_opts;

typedef struct {
    orc_kitconfig Config;
    const char* const* Pattern;  // npattern entries
    int npattern;
    int InvertMatch, ByName, BySeq, OnlyPositiveStrand, IgnoreCase;
    const char* Region;
    int Circular, Count, UseRegexp, Degenerate, MaxMismatch, DeleteMatched;
    const char* PatternFile;
} orc_grep_opts;

typedef struct {
    orc_kitconfig Config;
    const char* Region;
    int UpStream, DownStream, OnlyFlank;
    const char* Gtf;
    const char* Bed;
    const char* Chr;      /* newline-joined lists */
    const char* Feature;
    const char* GtfTag;
} orc_subseq_opts;

typedef struct {
    orc_kitconfig Config;
    const char* const* Pattern;
    int npattern;
    int IgnoreCase, OnlyPositiveStrand, NonGreedy, Gtf, Bed, HideMatched, Circular;
    int Degenerate, UseRegexp, UseFmi, MaxMismatch;
    const char* PatternFile;
} orc_locate_opts;

typedef struct {
    orc_kitconfig Config;
    int TranslTable;
    const char* const* Frame;
    int nframe;
    int Trim, Clean, AllowUnknownCodon, InitCodonAsM, ListTranslTable, ListTranslTableWithAmbCodons, AppendFrame;
} orc_translate_opts;

typedef struct {
    orc_kitconfig Config;
    int ByName, BySeq, IgnoreCase, OnlyPositiveStrand;
} orc_rmdup_opts;

}  // extern "C"

static KitConfig conv(const orc_kitconfig& c) {
    KitConfig k;
    if (c.SeqType) k.SeqType = c.SeqType;
    k.LineWidth = c.LineWidth;
    if (c.IDRegexp) k.IDRegexp = c.IDRegexp;
    k.IDNCBI = c.IDNCBI != 0;
    if (k.IDNCBI) k.IDRegexp = "\\|([^\\|]+)\\| ";  // bigseqkit/helper.go:97-100
    k.Quiet = c.Quiet != 0;
    k.AlphabetGuessSeqLength = c.AlphabetGuessSeqLength;
    k.ValidateSeqLength = c.ValidateSeqLength;
    return k;
}

static StatsOptions conv(const orc_stats_opts& c) {
    StatsOptions o;
    o.Config = conv(c.Config);
    o.Tabular = c.Tabular != 0;
    if (c.GapLetters) o.GapLetters = c.GapLetters;
    o.All = c.All != 0;
    o.SkipErr = c.SkipErr != 0;
    if (c.FqEncoding) o.FqEncoding = c.FqEncoding;
    o.Basename = c.Basename != 0;
    return o;
}

static SeqOptions conv(const orc_seq_opts& c) {
    SeqOptions o;
    o.Config = conv(c.Config);
    o.Reverse = c.Reverse; o.Complement = c.Complement; o.Name = c.Name; o.Seq = c.Seq; o.Qual = c.Qual;
    o.OnlyId = c.OnlyId; o.RemoveGaps = c.RemoveGaps;
    if (c.GapLetters) o.GapLetters = c.GapLetters;
    o.LowerCase = c.LowerCase; o.UpperCase = c.UpperCase; o.Dna2rna = c.Dna2rna; o.Rna2dna = c.Rna2dna;
    o.ValidateSeq = c.ValidateSeq; o.ValidateSeqLength = c.ValidateSeqLength;
    o.MaxLen = c.MaxLen; o.MinLen = c.MinLen; o.QualAsciiBase = c.QualAsciiBase;
    o.MinQual = c.MinQual; o.MaxQual = c.MaxQual;
    return o;
}

static GrepOptions conv(const orc_grep_opts& c) {
    GrepOptions o;
    o.Config = conv(c.Config);
    o.Pattern.clear();
    for (int i = 0; i < c.npattern; ++i) o.Pattern.push_back(c.Pattern[i]);
    o.InvertMatch = c.InvertMatch; o.ByName = c.ByName; o.BySeq = c.BySeq;
    o.OnlyPositiveStrand = c.OnlyPositiveStrand; o.IgnoreCase = c.IgnoreCase;
    if (c.Region) o.Region = c.Region;
    o.Circular = c.Circular; o.Count = c.Count; o.UseRegexp = c.UseRegexp; o.Degenerate = c.Degenerate;
    o.MaxMismatch = c.MaxMismatch; o.DeleteMatched = c.DeleteMatched;
    if (c.PatternFile) o.PatternFile = c.PatternFile;
    return o;
}

static SubseqOptions conv(const orc_subseq_opts& c) {
    SubseqOptions o;
    o.Config = conv(c.Config);
    if (c.Region) o.Region = c.Region;
    o.UpStream = c.UpStream; o.DownStream = c.DownStream; o.OnlyFlank = c.OnlyFlank;
    if (c.Gtf) o.Gtf = c.Gtf;
    if (c.Bed) o.Bed = c.Bed;
    auto split = [](const char* s) {
        std::vector<std::string> v;
        if (!s || !*s) return v;
        std::string t = s;
        size_t i = 0;
        for (;;) {
            size_t j = t.find('\n', i);
            if (j == std::string::npos) { v.push_back(t.substr(i)); break; }
            v.push_back(t.substr(i, j - i));
            i = j + 1;
        }
        return v;
    };
    o.Chr = split(c.Chr);
    o.Feature = split(c.Feature);
    if (c.GtfTag) o.GtfTag = c.GtfTag;
    return o;
}

static TranslateOptions conv(const orc_translate_opts& c) {
    TranslateOptions o;
    o.Config = conv(c.Config);
    o.TranslTable = c.TranslTable;
    o.Frame.clear();
    for (int i = 0; i < c.nframe; ++i) o.Frame.push_back(c.Frame[i]);
    o.Trim = c.Trim; o.Clean = c.Clean; o.AllowUnknownCodon = c.AllowUnknownCodon; o.InitCodonAsM = c.InitCodonAsM;
    o.ListTranslTable = c.ListTranslTable; o.ListTranslTableWithAmbCodons = c.ListTranslTableWithAmbCodons;
    o.AppendFrame = c.AppendFrame;
    return o;
}

static LocateOptions conv(const orc_locate_opts& c) {
    LocateOptions o;
    o.Config = conv(c.Config);
    o.Pattern.clear();
    for (int i = 0; i < c.npattern; ++i) o.Pattern.push_back(c.Pattern[i]);
    o.IgnoreCase = c.IgnoreCase; o.OnlyPositiveStrand = c.OnlyPositiveStrand; o.NonGreedy = c.NonGreedy;
    o.Gtf = c.Gtf; o.Bed = c.Bed; o.HideMatched = c.HideMatched; o.Circular = c.Circular;
    o.Degenerate = c.Degenerate; o.UseRegexp = c.UseRegexp; o.UseFmi = c.UseFmi; o.MaxMismatch = c.MaxMismatch;
    if (c.PatternFile) o.PatternFile = c.PatternFile;
    return o;
}

static RmDupOptions conv(const orc_rmdup_opts& c) {
    RmDupOptions o;
    o.Config = conv(c.Config);
    o.ByName = c.ByName; o.BySeq = c.BySeq; o.IgnoreCase = c.IgnoreCase; o.OnlyPositiveStrand = c.OnlyPositiveStrand;
    return o;
}

// elements joined the way FileStore writes them (bigseqkit-lib/helper.go:447): e + "\n"
static int emit(const std::vector<std::string>& els, uint8_t* out, size_t cap, size_t* nout, uint64_t* nrec) {
    size_t tot = 0;
    for (auto& e : els) tot += e.size() + 1;
    *nout = tot;
    if (nrec) *nrec = els.size();
    if (tot > cap) return 2;
    size_t p = 0;
    for (auto& e : els) {
        memcpy(out + p, e.data(), e.size());
        p += e.size();
        out[p++] = '\n';
    }
    return 0;
}

static int fail(char* err, size_t cap, const std::exception& e) {
    if (err && cap) snprintf(err, cap, "%s", e.what());
    return 1;
}

// Split [buf, buf+n) into `nparts` contiguous groups of whole records
// (IgnisHPC partitions), run Stats.Call on each and fold with StatsReduce.
static std::map<int64_t, int64_t> stats_over_parts(std::string_view buf, bool fastq, const StatsOptions& o, int nparts,
                                                   std::string_view* first) {
    auto recs = split_records(buf, fastq);
    if (first) *first = recs.empty() ? std::string_view() : recs[0];
    if (nparts < 1) nparts = 1;
    std::map<int64_t, int64_t> acc;
    bool have = false;
    for (int p = 0; p < nparts; ++p) {
        size_t a = recs.size() * (size_t)p / (size_t)nparts, b = recs.size() * (size_t)(p + 1) / (size_t)nparts;
        std::vector<std::string_view> part(recs.begin() + a, recs.begin() + b);
        auto m = stats_call(part, o);
        acc = have ? stats_reduce(acc, m) : m;
        have = true;
    }
    return acc;
}

template <class Opt, class Fn>
static int run_parts(const uint8_t* buf, size_t n, int fastq, const Opt& so, int nparts, Fn fn, bool sum_counts,
                     uint8_t* out, size_t cap, size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    try {
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        if (nparts < 1) nparts = 1;
        std::vector<std::string> all;
        int64_t total = 0;
        for (int p = 0; p < nparts; ++p) {
            size_t a = recs.size() * (size_t)p / (size_t)nparts, b = recs.size() * (size_t)(p + 1) / (size_t)nparts;
            std::vector<std::string_view> part(recs.begin() + a, recs.begin() + b);
            auto r = fn(part, so);
            if (sum_counts) total += strtoll(r.at(0).c_str(), nullptr, 10);  // GrepReduceCount (grep.go:598-611)
            else all.insert(all.end(), r.begin(), r.end());
        }
        if (sum_counts) all.push_back(std::to_string(total));
        return emit(all, out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

extern "C" {

// number of records and strictness probes
int orc_count_records(const uint8_t* buf, size_t n, int fastq, uint64_t* out, char* err, size_t errcap) {
    try {
        *out = split_records(std::string_view((const char*)buf, n), fastq != 0).size();
        return 0;
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

int orc_is_strict_4line_fastq(const uint8_t* buf, size_t n) {
    return is_strict_4line_fastq(std::string_view((const char*)buf, n)) ? 1 : 0;
}

// record boundaries (start offsets of each element and its length) -- used to
// check the HIP index kernels.
int orc_record_spans(const uint8_t* buf, size_t n, int fastq, uint64_t* starts, uint64_t* lens, size_t cap,
                     size_t* nout, char* err, size_t errcap) {
    try {
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        *nout = recs.size();
        for (size_t i = 0; i < recs.size() && i < cap; ++i) {
            starts[i] = (uint64_t)(recs[i].data() - (const char*)buf);
            lens[i] = recs[i].size();
        }
        return 0;
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// Stats.Call (+StatsReduce over nparts partitions) -> sorted (key,value) pairs
int orc_stats_map(const uint8_t* buf, size_t n, int fastq, const orc_stats_opts* o, int nparts, int64_t* keys,
                  int64_t* vals, size_t cap, size_t* nout, char* err, size_t errcap) {
    try {
        auto m = stats_over_parts(std::string_view((const char*)buf, n), fastq != 0, conv(*o), nparts, nullptr);
        *nout = m.size();
        size_t i = 0;
        for (auto& kv : m) {
            if (i >= cap) break;
            keys[i] = kv.first;
            vals[i] = kv.second;
            ++i;
        }
        return 0;
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// driver Stats() + StatsString()
int orc_stats_string(const uint8_t* buf, size_t n, int fastq, const orc_stats_opts* o, int nparts, const char* name,
                     const char* format, char* out, size_t cap, char* err, size_t errcap) {
    try {
        StatsOptions so = conv(*o);
        std::string_view first;
        auto m = stats_over_parts(std::string_view((const char*)buf, n), fastq != 0, so, nparts, &first);
        StatInfo info = stats_finalize(name, format, m, first, so);
        std::string s = stats_string(info, so);
        snprintf(out, cap, "%s", s.c_str());
        return 0;
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// Finalise a map that came from elsewhere (e.g. the HIP path) -- lets a test
// compare the driver-side arithmetic in isolation.
int orc_stats_string_from_map(const int64_t* keys, const int64_t* vals, size_t nkv, const uint8_t* first_rec,
                              size_t first_len, const orc_stats_opts* o, const char* name, const char* format,
                              char* out, size_t cap, char* err, size_t errcap) {
    try {
        StatsOptions so = conv(*o);
        std::map<int64_t, int64_t> m;
        for (size_t i = 0; i < nkv; ++i) m[keys[i]] = vals[i];
        StatInfo info = stats_finalize(name, format, m, std::string_view((const char*)first_rec, first_len), so);
        std::string s = stats_string(info, so);
        snprintf(out, cap, "%s", s.c_str());
        return 0;
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// SeqTransform over nparts partitions (each partition re-decides FASTQ-ness / alphabet from its first record)
int orc_seq(const uint8_t* buf, size_t n, int fastq, const orc_seq_opts* o, int nparts, uint8_t* out, size_t cap,
            size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    try {
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        if (nparts < 1) nparts = 1;
        std::vector<std::string> all;
        SeqOptions so = conv(*o);
        for (int p = 0; p < nparts; ++p) {
            size_t a = recs.size() * (size_t)p / (size_t)nparts, b = recs.size() * (size_t)(p + 1) / (size_t)nparts;
            std::vector<std::string_view> part(recs.begin() + a, recs.begin() + b);
            auto r = seq_call(part, so);
            all.insert(all.end(), r.begin(), r.end());
        }
        return emit(all, out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

int orc_grep(const uint8_t* buf, size_t n, int fastq, const orc_grep_opts* o, int nparts, uint8_t* out, size_t cap,
             size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    GrepOptions so = conv(*o);
    return run_parts(buf, n, fastq, so, nparts, grep_call, so.Count, out, cap, nout, nrec, err, errcap);
}

int orc_subseq(const uint8_t* buf, size_t n, int fastq, const orc_subseq_opts* o, int nparts, uint8_t* out, size_t cap,
               size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    SubseqOptions so = conv(*o);
    return run_parts(buf, n, fastq, so, nparts, subseq_call, false, out, cap, nout, nrec, err, errcap);
}

int orc_translate(const uint8_t* buf, size_t n, int fastq, const orc_translate_opts* o, int nparts, uint8_t* out,
                  size_t cap, size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    TranslateOptions so = conv(*o);
    return run_parts(buf, n, fastq, so, nparts, translate_call, false, out, cap, nout, nrec, err, errcap);
}

int orc_concat(const uint8_t* a, size_t na, const uint8_t* b, size_t nb, int fastq, const orc_kitconfig* cfg, int full,
               uint8_t* out, size_t cap, size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    try {
        auto ra = split_records(std::string_view((const char*)a, na), fastq != 0);
        auto rb = split_records(std::string_view((const char*)b, nb), fastq != 0);
        return emit(concat_call(ra, rb, conv(*cfg), full != 0), out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// common over nfiles files given back to back; ends[f] = offset one past file f
int orc_common(const uint8_t* buf, const uint64_t* ends, int nfiles, int fastq, const orc_kitconfig* cfg, int by_name, int by_seq,
               int ignore_case, int only_pos, uint8_t* out, size_t cap, size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    try {
        std::vector<std::vector<std::string_view>> files;
        uint64_t lo = 0;
        for (int f = 0; f < nfiles; ++f) {
            files.push_back(split_records(std::string_view((const char*)buf + lo, ends[f] - lo), fastq != 0));
            lo = ends[f];
        }
        CommonOptions o;
        o.Config = conv(*cfg);
        o.ByName = by_name != 0; o.BySeq = by_seq != 0; o.IgnoreCase = ignore_case != 0; o.OnlyPositiveStrand = only_pos != 0;
        return emit(common_call(files, o), out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// which: 0 paired.1, 1 paired.2, 2 unpaired.1, 3 unpaired.2
int orc_pair(const uint8_t* a, size_t na, const uint8_t* b, size_t nb, int fastq, const orc_kitconfig* cfg, int which,
             uint8_t* out, size_t cap, size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    try {
        auto ra = split_records(std::string_view((const char*)a, na), fastq != 0);
        auto rb = split_records(std::string_view((const char*)b, nb), fastq != 0);
        std::vector<std::string> outs[4];
        pair_call(ra, rb, conv(*cfg), outs);
        return emit(outs[which & 3], out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// queries: newline-joined region strings
int orc_faidx_query(const uint8_t* buf, size_t n, int fastq, const orc_kitconfig* cfg, const char* queries, int ignore_case,
                    int use_regexp, uint8_t* out, size_t cap, size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    try {
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        std::vector<std::string> qs;
        std::string all = queries ? queries : "";
        for (size_t a = 0; a < all.size();) {
            size_t b = all.find('\n', a);
            if (b == std::string::npos) b = all.size();
            if (b > a) qs.push_back(all.substr(a, b - a));
            a = b + 1;
        }
        return emit(faidx_query_call(recs, qs, ignore_case != 0, conv(*cfg), use_regexp != 0), out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// faidx index rows over nparts partitions (offsets = prefix sums of the partition sizes)
int orc_faidx(const uint8_t* buf, size_t n, int fastq, const orc_kitconfig* cfg, int full_head, int nparts, uint8_t* out,
              size_t cap, size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    try {
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        if (nparts < 1) nparts = 1;
        std::vector<std::string> all;
        uint64_t base = 0;
        for (int p = 0; p < nparts; ++p) {
            size_t a = recs.size() * (size_t)p / (size_t)nparts, b = recs.size() * (size_t)(p + 1) / (size_t)nparts;
            std::vector<std::string_view> part(recs.begin() + a, recs.begin() + b);
            uint64_t bytes = 0;
            auto r = faidx_call(part, base, full_head != 0, conv(*cfg), &bytes);
            base += bytes;
            all.insert(all.end(), r.begin(), r.end());
        }
        return emit(all, out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

typedef struct {
    orc_kitconfig Config;
    int InNaturalOrder, BySeq, ByName, ByLength, ByBases;
    const char* GapLetters;
    int Reverse, IgnoreCase;
    long long SeqPrefixLength;
} orc_sort_opts;

int orc_sort(const uint8_t* buf, size_t n, int fastq, const orc_sort_opts* o, uint8_t* out, size_t cap, size_t* nout,
             uint64_t* nrec, char* err, size_t errcap) {
    try {
        SortOptions so;
        so.Config = conv(o->Config);
        so.InNaturalOrder = o->InNaturalOrder != 0; so.BySeq = o->BySeq != 0; so.ByName = o->ByName != 0;
        so.ByLength = o->ByLength != 0; so.ByBases = o->ByBases != 0;
        if (o->GapLetters) so.GapLetters = o->GapLetters;
        so.Reverse = o->Reverse != 0; so.IgnoreCase = o->IgnoreCase != 0;
        so.SeqPrefixLength = o->SeqPrefixLength;
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        return emit(sort_call(recs, so), out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// rename is global (GroupByKey)
int orc_rename(const uint8_t* buf, size_t n, int fastq, const orc_kitconfig* cfg, int by_name, uint8_t* out, size_t cap,
               size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    try {
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        return emit(rename_call(recs, conv(*cfg), by_name != 0), out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// which: 0 fq2fa, 1 range (Range), 2 head (N), 3 duplicate (Times)
int orc_records(const uint8_t* buf, size_t n, int fastq, const orc_kitconfig* cfg, int which, const char* range,
                long long num, int nparts, uint8_t* out, size_t cap, size_t* nout, uint64_t* nrec, char* err,
                size_t errcap) {
    try {
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        if (nparts < 1) nparts = 1;
        std::vector<std::string> all;
        int64_t start = 0, end = 0;
        if (which == 1 || which == 2)
            range_bounds(which == 2 ? "1:" + std::to_string(num) : std::string(range ? range : ""), (int64_t)recs.size(),
                         &start, &end);
        for (int p = 0; p < nparts; ++p) {
            size_t a = recs.size() * (size_t)p / (size_t)nparts, b = recs.size() * (size_t)(p + 1) / (size_t)nparts;
            std::vector<std::string_view> part(recs.begin() + a, recs.begin() + b);
            std::vector<std::string> r;
            if (which == 0) r = fq2fa_call(part, conv(*cfg));
            else if (which == 3) r = duplicate_call(part, num);
            else r = range_call(part, (int64_t)a, start, end);
            all.insert(all.end(), r.begin(), r.end());
        }
        return emit(all, out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// rmdup is global (GroupByKey): nparts is ignored, the whole input is one group space
// which: 1 = text of the removed records (-d), 2 = duplicate-number lines (-D)
int orc_rmdup_side(const uint8_t* buf, size_t n, int fastq, const orc_rmdup_opts* o, int which, uint8_t* out, size_t cap,
                   size_t* nout, char* err, size_t errcap) {
    try {
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        std::string seqs, nums;
        rmdup_call_side(recs, conv(*o), &seqs, &nums);
        const std::string& r = which == 1 ? seqs : nums;
        *nout = r.size();
        if (r.size() > cap) return 2;
        memcpy(out, r.data(), r.size());
        return 0;
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

int orc_rmdup(const uint8_t* buf, size_t n, int fastq, const orc_rmdup_opts* o, int nparts, uint8_t* out, size_t cap,
              size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    (void)nparts;
    RmDupOptions so = conv(*o);
    return run_parts(buf, n, fastq, so, 1, rmdup_call, false, out, cap, nout, nrec, err, errcap);
}

// rmdup on `threads` host threads (bench.py's all-cores CPU baseline)
int orc_rmdup_mt(const uint8_t* buf, size_t n, int fastq, const orc_rmdup_opts* o, int threads, uint8_t* out, size_t cap,
                 size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    try {
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        return emit(rmdup_call_mt(recs, conv(*o), threads), out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

// Locate over nparts partitions: MapPartitionsWithIndex, the header row comes from partition 0
int orc_locate(const uint8_t* buf, size_t n, int fastq, const orc_locate_opts* o, int nparts, uint8_t* out, size_t cap,
               size_t* nout, uint64_t* nrec, char* err, size_t errcap) {
    try {
        auto recs = split_records(std::string_view((const char*)buf, n), fastq != 0);
        if (nparts < 1) nparts = 1;
        LocateOptions so = conv(*o);
        std::vector<std::string> all;
        for (int p = 0; p < nparts; ++p) {
            size_t a = recs.size() * (size_t)p / (size_t)nparts, b = recs.size() * (size_t)(p + 1) / (size_t)nparts;
            std::vector<std::string_view> part(recs.begin() + a, recs.begin() + b);
            auto r = locate_call(part, so, p);
            all.insert(all.end(), r.begin(), r.end());
        }
        return emit(all, out, cap, nout, nrec);
    } catch (const std::exception& e) { return fail(err, errcap, e); }
}

uint64_t orc_xxh64(const uint8_t* p, size_t n) { return xxh64(p, n, 0); }

// the 64-letter gc.prt line the oracle derived for a table (tests compare it with the product's table); 0 = unknown id
int orc_genetic_code(int id, int which, char* out65) {
    const char* s = genetic_code_strings(id, which);
    if (!s) return 0;
    memcpy(out65, s, 64);
    out65[64] = 0;
    return 1;
}

// every triple over the 15 IUPAC letters + one invalid byte, in both cases, for table `id`: number of triples on which the
// lookup table of translate_seq disagrees with codon_aa (0 expected); -1: unknown id
int orc_codon_table_mismatches(int id) {
    static const char L[] = "ACGTRYSWKMBDHVNacgturyswkmbdhvnU?";
    const int n = (int)sizeof(L) - 1;
    int bad = 0;
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
            for (int c = 0; c < n; ++c) {
                const char c3[3] = {L[a], L[b], L[c]};
                char s = 0, f = 0;
                if (codon_aa_pair(id, c3, &s, &f) != 0) return -1;
                bad += s != f;
            }
    return bad;
}

int orc_translate_seq(const char* seq, int table, int frame, int trim, int clean, int allow_unknown, int init_m,
                      char* out, size_t cap) {
    try {
        bool unknown = false;
        std::string aa = translate_seq(seq, table, frame, trim, clean, allow_unknown, init_m, &unknown);
        if (unknown) return 3;
        snprintf(out, cap, "%s", aa.c_str());
        return 0;
    } catch (const std::exception&) { return 1; }
}

// region table probe: 0-based [b, e)
int orc_sub_location(size_t length, int start, int end, size_t* b, size_t* e) {
    sub_location(length, start, end, b, e);
    return 0;
}

int orc_wrap(const uint8_t* s, size_t n, int width, uint8_t* out, size_t cap, size_t* nout) {
    std::string w = wrap_byte_slice(std::string_view((const char*)s, n), width);
    *nout = w.size();
    memcpy(out, w.data(), std::min(cap, w.size()));
    return 0;
}

int orc_parse_head(const char* head, const char* re, char* id, size_t idcap, char* desc, size_t desccap) {
    try {
        std::string i, d, r = re ? re : "";
        bool def = r.empty() || r == "^(\\S+)\\s?";
        parse_head_id_desc(head, def, r, i, d);
        snprintf(id, idcap, "%s", i.c_str());
        snprintf(desc, desccap, "%s", d.c_str());
        return 0;
    } catch (const std::exception&) { return 1; }
}

double orc_go_round(double f, int n) { return go_round(f, n); }

}  // extern "C"
