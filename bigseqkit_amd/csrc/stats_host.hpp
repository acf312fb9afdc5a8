// Driver-side stats arithmetic (see stats_host.cpp)
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/bsk.h"

namespace bsk {

using StatsMap = std::map<int64_t, int64_t>;
// special keys of the reference's map[int64]int64 (bigseqkit-lib/stats.go:68-71)
constexpr int64_t KEY_Q20 = -1, KEY_Q30 = -2, KEY_GAP = -3, KEY_TYPE = -4;

StatsMap stats_merge(const StatsMap& a, const StatsMap& b);
// type_if_F: alphabet name guessed from the first record (used when key -4 == 'F')
void stats_finalize(const StatsMap& m, bool all, const std::string& type_if_F, bsk_statinfo* out);
std::string stats_string(const std::string& name, const std::string& format, const bsk_statinfo& info, bool tabular,
                         bool all);

}  // namespace bsk
