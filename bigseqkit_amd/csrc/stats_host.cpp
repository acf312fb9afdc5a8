// Driver-side half of `stats`: what the reference does on the IgnisHPC driver
// after the executors returned their maps.
//   StatsReduce.Call   /root/reference/bigseqkit-lib/stats.go:128-137
//   Stats()            bigseqkit/stats.go:75-166
//   StatsString()      bigseqkit/stats.go:168-288
// Third-party pieces restated from their published behaviour (not in tree):
// util.LengthStats (shenwei356/bio v0.7.0), math.Round (shenwei356/util v0.5.0),
// humanize.Comma/Commaf (go-humanize v1.0.0), go-prettytable.
#include "stats_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bsk {

// The reference rebuilds LengthStats by calling Add(k) v times (stats.go:134-138,
// O(#records)); we work on the (length,count) pairs directly: same multiset.
namespace {

struct Pairs {
    std::vector<std::pair<uint64_t, uint64_t>> c;  // ascending length
    uint64_t count = 0, sum = 0;

    // length at 0-based index `idx` of the sorted expanded multiset
    double at(uint64_t idx) const {
        uint64_t acc = 0;
        for (auto& p : c) {
            acc += p.second;
            if (idx < acc) return (double)p.first;
        }
        return c.empty() ? 0.0 : (double)c.back().first;
    }
    double mid(uint64_t n_items, uint64_t base) const {
        // median of n_items consecutive items starting at index base
        if (n_items % 2 == 0) return (at(base + n_items / 2 - 1) + at(base + n_items / 2)) / 2.0;
        return at(base + n_items / 2);
    }
};

double round_n(double f, int n) {  // util/math.Round
    const double p = std::pow(10.0, n);
    return std::trunc((f + 0.5 / p) * p) / p;
}

std::string comma_u64(uint64_t v) {
    std::string d = std::to_string(v), o;
    for (size_t i = 0; i < d.size(); ++i) {
        if (i && (d.size() - i) % 3 == 0) o.push_back(',');
        o.push_back(d[i]);
    }
    return o;
}

std::string shortest(double v) {  // strconv.FormatFloat(v, 'f', -1, 64)
    char b[400];
    for (int dec = 0; dec < 340; ++dec) {
        snprintf(b, sizeof b, "%.*f", dec, v);
        if (strtod(b, nullptr) == v) break;
    }
    return b;
}

std::string commaf(double v) {
    std::string s = shortest(std::fabs(v));
    size_t dot = s.find('.');
    std::string ip = s.substr(0, dot), fp = dot == std::string::npos ? std::string() : s.substr(dot);
    std::string o;
    for (size_t i = 0; i < ip.size(); ++i) {
        if (i && (ip.size() - i) % 3 == 0) o.push_back(',');
        o.push_back(ip[i]);
    }
    return (v < 0 ? "-" : "") + o + fp;
}

std::string fmt(const char* f, ...) {
    char b[2048];
    va_list ap;
    va_start(ap, f);
    vsnprintf(b, sizeof b, f, ap);
    va_end(ap);
    return b;
}

}  // namespace

StatsMap stats_merge(const StatsMap& a, const StatsMap& b) {
    StatsMap r = a;
    for (auto& kv : b) {
        if (kv.first == KEY_TYPE) {
            auto it = r.find(KEY_TYPE);
            if (it == r.end() || it->second == 'U') r[KEY_TYPE] = kv.second;  // lower partition wins
        } else {
            r[kv.first] += kv.second;  // PARITY.md Q2: sum, not overwrite
        }
    }
    return r;
}

void stats_finalize(const StatsMap& m, bool all, const std::string& type_if_F, bsk_statinfo* out) {
    memset(out, 0, sizeof(*out));
    auto get = [&](int64_t k) {
        auto it = m.find(k);
        return it == m.end() ? (int64_t)0 : it->second;
    };
    const int64_t q20 = get(KEY_Q20), q30 = get(KEY_Q30), ti = get(KEY_TYPE);
    const uint64_t gap = (uint64_t)get(KEY_GAP);
    std::string t;
    if (ti == 'D') t = "DNA";
    else if (ti == 'R') t = "RNA";
    else if (ti == 'U') t = "";
    else t = type_if_F;  // bigseqkit/stats.go:117-129
    snprintf(out->type, sizeof out->type, "%s", t.c_str());

    Pairs P;
    for (auto& kv : m) {
        if (kv.first < 0 || kv.second <= 0) continue;
        P.c.emplace_back((uint64_t)kv.first, (uint64_t)kv.second);
        P.count += (uint64_t)kv.second;
        P.sum += (uint64_t)kv.first * (uint64_t)kv.second;
    }
    if (P.count == 0) return;  // all-zero StatInfo (stats.go:148-152)
    if (all) {
        // N50 / L50: walk from the longest, one sequence at a time
        const double half = (double)P.sum / 2.0;
        double acc = 0;
        uint64_t nseq = 0;
        out->n50 = P.c.front().first;
        out->l50 = (int64_t)P.count;
        for (size_t i = P.c.size(); i-- > 0;) {
            const double len = (double)P.c[i].first, tot = len * (double)P.c[i].second;
            if (acc + tot >= half) {
                uint64_t k = len > 0 ? (uint64_t)std::ceil((half - acc) / len) : 1;
                k = std::max<uint64_t>(1, std::min<uint64_t>(k, P.c[i].second));
                out->n50 = P.c[i].first;
                out->l50 = (int64_t)(nseq + k);
                break;
            }
            acc += tot;
            nseq += P.c[i].second;
        }
        // quartiles: medians of the lower / whole / upper halves, the middle
        // element belonging to both halves when the count is odd
        if (P.count == 1) {
            out->q1 = out->q2 = out->q3 = (double)P.c[0].first;
        } else {
            const uint64_t h = (P.count % 2 == 0) ? P.count / 2 : (P.count + 1) / 2;
            const uint64_t upper0 = (P.count % 2 == 0) ? h : h - 1;
            out->q1 = P.mid(h, 0);
            out->q2 = P.mid(P.count, 0);
            out->q3 = P.mid(h, upper0);
        }
    }
    out->num = P.count;
    out->len_sum = P.sum;
    out->gap_sum = gap;
    out->len_min = P.c.front().first;
    out->len_max = P.c.back().first;
    out->len_avg = round_n((double)P.sum / (double)P.count, 1);
    out->q20 = round_n((double)q20 / (double)P.sum * 100.0, 2);
    out->q30 = round_n((double)q30 / (double)P.sum * 100.0, 2);
}

std::string stats_string(const std::string& name, const std::string& format, const bsk_statinfo& i, bool tabular,
                         bool all) {
    if (tabular) {
        std::string r = "file\tformat\ttype\tnum_seqs\tsum_len\tmin_len\tavg_len\tmax_len";
        if (all) r += "\tQ1\tQ2\tQ3\tsum_gap\tN50\tQ20(%)\tQ30(%)";
        r += "\n";
        r += fmt("%s\t%s\t%s\t%llu\t%llu\t%llu\t%.1f\t%llu", name.c_str(), format.c_str(), i.type,
                 (unsigned long long)i.num, (unsigned long long)i.len_sum, (unsigned long long)i.len_min, i.len_avg,
                 (unsigned long long)i.len_max);
        if (all)
            r += fmt("\t%.1f\t%.1f\t%.1f\t%llu\t%llu\t%.2f\t%.2f", i.q1, i.q2, i.q3, (unsigned long long)i.gap_sum,
                     (unsigned long long)i.n50, i.q20, i.q30);
        r += "\n";
        return r;
    }
    struct Col { std::string head, cell; bool right; };
    std::vector<Col> cols = {{"file", name, false},
                             {"format", format, false},
                             {"type", i.type, false},
                             {"num_seqs", comma_u64(i.num), true},
                             {"sum_len", comma_u64(i.len_sum), true},
                             {"min_len", comma_u64(i.len_min), true},
                             {"avg_len", commaf(i.len_avg), true},
                             {"max_len", comma_u64(i.len_max), true}};
    if (all) {
        cols.push_back({"Q1", commaf(i.q1), true});
        cols.push_back({"Q2", commaf(i.q2), true});
        cols.push_back({"Q3", commaf(i.q3), true});
        cols.push_back({"sum_gap", comma_u64(i.gap_sum), true});
        cols.push_back({"N50", comma_u64(i.n50), true});
        cols.push_back({"Q20(%)", commaf(i.q20), true});
        cols.push_back({"Q30(%)", commaf(i.q30), true});
    }
    std::string out;
    for (int row = 0; row < 2; ++row) {
        for (size_t c = 0; c < cols.size(); ++c) {
            const std::string& s = row == 0 ? cols[c].head : cols[c].cell;
            const size_t w = std::max(cols[c].head.size(), cols[c].cell.size());
            if (c) out.push_back(' ');
            if (cols[c].right) out.append(w - s.size(), ' ').append(s);
            else out.append(s).append(w - s.size(), ' ');
        }
        out.push_back('\n');
    }
    return out;
}

}  // namespace bsk
