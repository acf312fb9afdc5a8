// Regular expressions for `grep -r` on the device: Go `regexp` (RE2) syntax subset -> Glushkov position automaton
// with at most 64 positions, simulated bit-parallel (one u64 of active positions per record).
//
// The reference calls re.Match(target) (bigseqkit-lib/grep.go:459-468): an UNANCHORED search whose only result is
// "matched or not", so leftmost-first / greedy-vs-lazy distinctions do not matter and the position automaton gives
// exactly RE2's answer for the supported syntax:
//   literals, escapes (\. \t \n \r \f \v \xHH \d \D \w \W \s \S), '.', classes [a-z0-9_] [^...] with POSIX names
//   ([:alpha:] ...), groups (...) (?:...) (?P<n>...), alternation, * + ? {m} {m,} {m,n} (lazy forms accepted),
//   anchors ^ $ \A \z, flags (?i) (?s) (?is) at the start of the expression.
// Not supported (bsk_create fails with an explicit message): \b \B, Unicode classes \p{..}, (?m), flags in the middle,
// more than 64 positions after expanding counted repetitions.  Bytes >= 0x80 are matched as bytes.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bsk {

constexpr int RE_SYM_BEGIN = 256;  // virtual symbol fed before the first byte (matches ^)
constexpr int RE_SYM_END = 257;    // virtual symbol fed after the last byte (matches $)
constexpr int RE_NSYM = 258;

struct RegexProgram {        // flat, device-friendly
    uint64_t first = 0;      // positions that can start a match (injected before every symbol: unanchored search)
    uint64_t last = 0;       // accepting positions
    uint32_t nullable = 0;   // the expression matches the empty string: every target matches
    uint32_t npos = 0;
    uint64_t accept[RE_NSYM];    // accept[sym]: positions whose symbol set contains sym
    uint64_t follow[8][256];     // follow[k][v]: union of follow(p) for the positions p = 8k + bit set in v
};

// throws OptError (message in the style of Go's "error parsing regexp: ...") on syntax errors and unsupported syntax
RegexProgram compile_regex(const std::string& expr);

// host-side simulation (used by bsk_create to vet a program and by CPU tests of the compiler)
bool regex_match(const RegexProgram& p, const uint8_t* text, size_t n);

}  // namespace bsk
