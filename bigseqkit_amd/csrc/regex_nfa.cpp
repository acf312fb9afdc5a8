// regex_nfa.cpp -- see regex_nfa.hpp.  Parser (recursive descent over RE2 syntax) + Glushkov construction.
#include "regex_nfa.hpp"
#include "regex_vm.hpp"

#include <array>
#include <cstring>
#include <memory>

#include "opts.hpp"  // OptError

namespace bsk {

namespace {

using ByteSet = std::array<uint32_t, 8>;
inline void bs_add(ByteSet& s, int b) { s[(b & 255) >> 5] |= 1u << (b & 31); }
inline bool bs_has(const ByteSet& s, int b) { return (s[(b & 255) >> 5] >> (b & 31)) & 1u; }
inline void bs_range(ByteSet& s, int lo, int hi) { for (int b = lo; b <= hi; ++b) bs_add(s, b); }
inline void bs_or(ByteSet& a, const ByteSet& b) { for (int i = 0; i < 8; ++i) a[i] |= b[i]; }
inline ByteSet bs_not(const ByteSet& a) { ByteSet r; for (int i = 0; i < 8; ++i) r[i] = ~a[i]; return r; }

enum class NT { Empty, Lit, Begin, End, WordB, NotWordB, Cat, Alt, Star, Plus, Quest, Repeat, Group };
struct Node {
    NT t = NT::Empty;
    ByteSet set{};
    std::vector<std::unique_ptr<Node>> kids;
    int lo = 0, hi = 0;  // Repeat: hi < 0 == unbounded
    int group = 0;       // Group: 1-based index of the capture group
    bool lazy = false;   // Star / Plus / Quest / Repeat: prefer the shorter match (matters only where positions are asked for)
};
using NodeP = std::unique_ptr<Node>;

[[noreturn]] void bad(const std::string& what, const std::string& expr) {
    throw OptError("error parsing regexp: " + what + ": `" + expr + "`");
}
[[noreturn]] void unsupported(const std::string& what, const std::string& expr) {
    throw OptError("libbsk: regexp syntax not supported by the HIP path (" + what + "): `" + expr + "`");
}

struct Parser {
    const std::string& e;
    size_t i = 0;
    bool icase = false, dotall = false, ungreedy = false;
    int ngroups = 0;
    explicit Parser(const std::string& s) : e(s) {}
    bool more() const { return i < e.size(); }
    char peek() const { return e[i]; }

    ByteSet fold(ByteSet s) const {
        if (!icase) return s;
        for (int c = 'a'; c <= 'z'; ++c) {
            if (bs_has(s, c)) bs_add(s, c - 32);
            if (bs_has(s, c - 32)) bs_add(s, c);
        }
        return s;
    }
    static ByteSet perl_class(char k) {  // \d \w \s
        ByteSet s{};
        switch (k) {
            case 'd': bs_range(s, '0', '9'); break;
            case 'w': bs_range(s, '0', '9'); bs_range(s, 'a', 'z'); bs_range(s, 'A', 'Z'); bs_add(s, '_'); break;
            case 's': for (char c : {'\t', '\n', '\f', '\r', ' '}) bs_add(s, c); break;
        }
        return s;
    }
    // escape after the backslash; returns true and fills `set` for a class escape, else a single byte in *lit
    bool escape(ByteSet* set, int* lit) {
        if (!more()) bad("trailing backslash at end of expression", e);
        const char c = e[i++];
        switch (c) {
            case 'd': case 'w': case 's': *set = perl_class(c); return true;
            case 'D': case 'W': case 'S': {
                ByteSet s = bs_not(perl_class((char)(c + 32)));
                // RE2 classes range over code points; as bytes: ASCII complement plus every byte >= 0x80
                *set = s;
                return true;
            }
            case 't': *lit = '\t'; return false;
            case 'n': *lit = '\n'; return false;
            case 'r': *lit = '\r'; return false;
            case 'f': *lit = '\f'; return false;
            case 'v': *lit = '\v'; return false;
            case 'a': *lit = 7; return false;
            case 'x': {
                auto hex = [&](char h) -> int {
                    if (h >= '0' && h <= '9') return h - '0';
                    if (h >= 'a' && h <= 'f') return h - 'a' + 10;
                    if (h >= 'A' && h <= 'F') return h - 'A' + 10;
                    return -1;
                };
                if (i + 1 < e.size() && hex(e[i]) >= 0 && hex(e[i + 1]) >= 0) {
                    *lit = hex(e[i]) * 16 + hex(e[i + 1]);
                    i += 2;
                    return false;
                }
                bad("invalid escape sequence", e);
            }
            case 'b': case 'B': bad("invalid escape sequence", e);  // (\\b inside a class; outside, atom() takes it)
            case 'p': case 'P': unsupported("Unicode class \\p", e);
            case 'A': case 'z': case 'Q': case 'E': case 'C': unsupported(std::string("\\") + c, e);
            default:
                if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9')) bad("invalid escape sequence", e);
                *lit = (unsigned char)c;  // escaped punctuation
                return false;
        }
    }
    NodeP lit(const ByteSet& s) {
        NodeP n(new Node);
        n->t = NT::Lit;
        n->set = fold(s);
        return n;
    }
    NodeP char_class() {  // after '['
        ByteSet s{};
        bool neg = false;
        if (more() && peek() == '^') { neg = true; ++i; }
        bool first = true;
        for (;;) {
            if (!more()) bad("missing closing ]", e);
            char c = e[i];
            if (c == ']' && !first) { ++i; break; }
            first = false;
            if (c == '[' && i + 1 < e.size() && e[i + 1] == ':') {
                const size_t j = e.find(":]", i + 2);
                if (j == std::string::npos) bad("missing closing ]", e);
                std::string name = e.substr(i + 2, j - i - 2);
                bool nn = false;
                if (!name.empty() && name[0] == '^') { nn = true; name.erase(0, 1); }
                ByteSet p{};
                if (name == "alpha") { bs_range(p, 'a', 'z'); bs_range(p, 'A', 'Z'); }
                else if (name == "digit") bs_range(p, '0', '9');
                else if (name == "alnum") { bs_range(p, 'a', 'z'); bs_range(p, 'A', 'Z'); bs_range(p, '0', '9'); }
                else if (name == "upper") bs_range(p, 'A', 'Z');
                else if (name == "lower") bs_range(p, 'a', 'z');
                else if (name == "space") { for (char w : {'\t', '\n', '\v', '\f', '\r', ' '}) bs_add(p, w); }
                else if (name == "punct") { bs_range(p, '!', '/'); bs_range(p, ':', '@'); bs_range(p, '[', '`'); bs_range(p, '{', '~'); }
                else if (name == "xdigit") { bs_range(p, '0', '9'); bs_range(p, 'a', 'f'); bs_range(p, 'A', 'F'); }
                else if (name == "word") { bs_range(p, 'a', 'z'); bs_range(p, 'A', 'Z'); bs_range(p, '0', '9'); bs_add(p, '_'); }
                else if (name == "blank") { bs_add(p, ' '); bs_add(p, '\t'); }
                else bad("invalid character class range", e);
                bs_or(s, nn ? bs_not(p) : p);
                i = j + 2;
                continue;
            }
            int lo;
            ++i;
            if (c == '\\') {
                ByteSet cls{};
                if (escape(&cls, &lo)) { bs_or(s, cls); continue; }
            } else {
                lo = (unsigned char)c;
            }
            int hi = lo;
            if (i + 1 < e.size() && e[i] == '-' && e[i + 1] != ']') {
                ++i;
                char d = e[i++];
                if (d == '\\') {
                    ByteSet cls{};
                    if (escape(&cls, &hi)) bad("invalid character class range", e);
                } else {
                    hi = (unsigned char)d;
                }
                if (hi < lo) bad("invalid character class range", e);
            }
            bs_range(s, lo, hi);
        }
        // fold before negating: (?i)[^a] excludes both cases
        s = fold(s);
        NodeP n(new Node);
        n->t = NT::Lit;
        n->set = neg ? bs_not(s) : s;
        return n;
    }
    NodeP atom() {
        const char c = e[i++];
        switch (c) {
            case '(': {
                bool capture = true;
                if (more() && peek() == '?') {
                    if (e.compare(i, 2, "?:") == 0) { i += 2; capture = false; }
                    else if (e.compare(i, 3, "?P<") == 0) {
                        const size_t j = e.find('>', i);
                        if (j == std::string::npos) bad("invalid named capture", e);
                        i = j + 1;
                    } else unsupported("flags / look-around inside the expression", e);
                }
                const int idx = capture ? ++ngroups : 0;  // groups are numbered by their opening parenthesis
                NodeP n = alt();
                if (!more() || peek() != ')') bad("missing closing )", e);
                ++i;
                if (!capture) return n;
                NodeP g(new Node);
                g->t = NT::Group;
                g->group = idx;
                g->kids.push_back(std::move(n));
                return g;
            }
            case '[': return char_class();
            case '.': {
                ByteSet s{};
                bs_range(s, 0, 255);
                if (!dotall) s['\n' >> 5] &= ~(1u << ('\n' & 31));
                NodeP n(new Node);
                n->t = NT::Lit;
                n->set = s;
                return n;
            }
            case '^': { NodeP n(new Node); n->t = NT::Begin; return n; }
            case '$': { NodeP n(new Node); n->t = NT::End; return n; }
            case '\\': {
                if (more() && (peek() == 'b' || peek() == 'B')) {  // ASCII word boundary / not a boundary (RE2: \b \B)
                    NodeP n(new Node);
                    n->t = e[i++] == 'b' ? NT::WordB : NT::NotWordB;
                    return n;
                }
                ByteSet cls{};
                int b = 0;
                if (escape(&cls, &b)) { NodeP n(new Node); n->t = NT::Lit; n->set = cls; return n; }
                ByteSet s{};
                bs_add(s, b);
                return lit(s);
            }
            case '*': case '+': case '?': bad("missing argument to repetition operator", e);
            case ')': bad("unexpected )", e);
            default: {
                ByteSet s{};
                bs_add(s, (unsigned char)c);
                return lit(s);
            }
        }
    }
    NodeP repeat() {
        NodeP a = atom();
        for (;;) {
            if (!more()) return a;
            const char c = peek();
            NodeP n(new Node);
            if (c == '*') { n->t = NT::Star; ++i; }
            else if (c == '+') { n->t = NT::Plus; ++i; }
            else if (c == '?') { n->t = NT::Quest; ++i; }
            else if (c == '{') {
                // {m} {m,} {m,n}; anything else is a literal '{' (RE2)
                size_t j = i + 1;
                auto num = [&](int* v) { size_t k = j; long x = 0; while (j < e.size() && isdigit((unsigned char)e[j])) { x = x * 10 + (e[j] - '0'); if (x > 1000) x = 1001; ++j; } *v = (int)x; return j > k; };
                int lo = 0, hi = 0;
                if (!num(&lo)) return a;
                if (j < e.size() && e[j] == '}') hi = lo;
                else if (j < e.size() && e[j] == ',') {
                    ++j;
                    if (j < e.size() && e[j] == '}') hi = -1;
                    else if (!num(&hi) || j >= e.size() || e[j] != '}') return a;
                } else return a;
                if (lo > 1000 || hi > 1000 || (hi >= 0 && hi < lo)) bad("invalid repeat count", e);
                i = j + 1;
                n->t = NT::Repeat; n->lo = lo; n->hi = hi;
            } else return a;
            n->lazy = ungreedy;
            if (more() && peek() == '?') { ++i; n->lazy = !ungreedy; }  // lazy form: same set of matching targets, other positions
            n->kids.push_back(std::move(a));
            a = std::move(n);
        }
    }
    NodeP concat() {
        NodeP n(new Node);
        n->t = NT::Cat;
        while (more() && peek() != '|' && peek() != ')') n->kids.push_back(repeat());
        if (n->kids.empty()) { n->t = NT::Empty; }
        return n;
    }
    NodeP alt() {
        NodeP first = concat();
        if (!more() || peek() != '|') return first;
        NodeP n(new Node);
        n->t = NT::Alt;
        n->kids.push_back(std::move(first));
        while (more() && peek() == '|') { ++i; n->kids.push_back(concat()); }
        return n;
    }
    NodeP parse() {
        // leading flag group (?i) (?s) (?is) ...
        while (e.compare(i, 2, "(?") == 0) {
            size_t j = i + 2;
            bool ok = j < e.size() && e[j] != ')';
            bool ic = icase, ds = dotall;
            for (; j < e.size() && e[j] != ')'; ++j) {
                if (e[j] == 'i') ic = true;
                else if (e[j] == 's') ds = true;
                else if (e[j] == 'U') ungreedy = true;  // swap greediness: irrelevant for a boolean match, kept for the VM
                else if (e[j] == 'm') {}  // ^ $ also at line breaks: the targets (ID, name, bases of ONE record) hold none -- the same matches
                else { ok = false; break; }
            }
            if (!ok || j >= e.size()) break;
            icase = ic; dotall = ds;
            i = j + 1;
        }
        NodeP n = alt();
        if (more()) bad("unexpected )", e);
        return n;
    }
};

struct Frag { bool nullable; uint64_t first, last; };

struct Builder {
    const std::string& expr;
    std::vector<std::array<uint32_t, 9>> sym;  // per position: 256-bit byte set + bit0 BEGIN / bit1 END in word 8
    std::vector<uint64_t> follow;
    explicit Builder(const std::string& e) : expr(e) {}
    uint64_t newpos(const ByteSet& s, bool begin, bool end) {
        if (sym.size() >= 64) unsupported("more than 64 positions after expanding repetitions", expr);
        std::array<uint32_t, 9> a{};
        for (int i = 0; i < 8; ++i) a[i] = s[i];
        a[8] = (begin ? 1u : 0u) | (end ? 2u : 0u);
        sym.push_back(a);
        follow.push_back(0);
        return 1ull << (sym.size() - 1);
    }
    void link(uint64_t from, uint64_t to) {
        for (size_t p = 0; p < follow.size(); ++p) if ((from >> p) & 1) follow[p] |= to;
    }
    Frag cat(Frag a, Frag b) {
        link(a.last, b.first);
        return {a.nullable && b.nullable, a.first | (a.nullable ? b.first : 0), b.last | (b.nullable ? a.last : 0)};
    }
    Frag build(const Node& n) {
        switch (n.t) {
            case NT::Empty: return {true, 0, 0};
            case NT::Lit: { uint64_t b = newpos(n.set, false, false); return {false, b, b}; }
            case NT::Begin: { uint64_t b = newpos(ByteSet{}, true, false); return {false, b, b}; }
            case NT::End: { uint64_t b = newpos(ByteSet{}, false, true); return {false, b, b}; }
            // (an assertion over TWO neighbouring bytes is no position of a Glushkov automaton: the callers that only need
            // "matches or not" take the thread-list matcher of regex_vm.hpp for such an expression)
            case NT::WordB: case NT::NotWordB: unsupported("word boundary \\b in the bit-parallel automaton", expr);
            case NT::Cat: {
                Frag f{true, 0, 0};
                for (auto& k : n.kids) f = cat(f, build(*k));
                return f;
            }
            case NT::Alt: {
                Frag f{false, 0, 0};
                for (auto& k : n.kids) { Frag g = build(*k); f.nullable |= g.nullable; f.first |= g.first; f.last |= g.last; }
                return f;
            }
            case NT::Star: { Frag f = build(*n.kids[0]); link(f.last, f.first); f.nullable = true; return f; }
            case NT::Plus: { Frag f = build(*n.kids[0]); link(f.last, f.first); return f; }
            case NT::Quest: { Frag f = build(*n.kids[0]); f.nullable = true; return f; }
            case NT::Group: return build(*n.kids[0]);
            case NT::Repeat: {
                Frag f{true, 0, 0};
                for (int k = 0; k < n.lo; ++k) f = cat(f, build(*n.kids[0]));
                if (n.hi < 0) { Frag s = build(*n.kids[0]); link(s.last, s.first); s.nullable = true; f = cat(f, s); }
                else for (int k = n.lo; k < n.hi; ++k) { Frag q = build(*n.kids[0]); q.nullable = true; f = cat(f, q); }
                return f;
            }
        }
        return {true, 0, 0};
    }
};

}  // namespace

RegexProgram compile_regex(const std::string& expr) {
    Parser ps(expr);
    NodeP root = ps.parse();
    Builder b(expr);
    const Frag f = b.build(*root);
    RegexProgram p{};
    p.first = f.first;
    p.last = f.last;
    p.nullable = f.nullable ? 1u : 0u;
    p.npos = (uint32_t)b.sym.size();
    for (size_t pos = 0; pos < b.sym.size(); ++pos) {
        for (int s = 0; s < 256; ++s)
            if ((b.sym[pos][s >> 5] >> (s & 31)) & 1u) p.accept[s] |= 1ull << pos;
        if (b.sym[pos][8] & 1u) p.accept[RE_SYM_BEGIN] |= 1ull << pos;
        if (b.sym[pos][8] & 2u) p.accept[RE_SYM_END] |= 1ull << pos;
    }
    for (int k = 0; k < 8; ++k)
        for (int v = 0; v < 256; ++v) {
            uint64_t u = 0;
            for (int bit = 0; bit < 8; ++bit)
                if (((v >> bit) & 1) && (size_t)(8 * k + bit) < b.follow.size()) u |= b.follow[8 * k + bit];
            p.follow[k][v] = u;
        }
    return p;
}

bool regex_match(const RegexProgram& p, const uint8_t* text, size_t n) {
    if (p.nullable) return true;
    uint64_t S = 0;
    auto step = [&](int sym) {
        uint64_t f = p.first;
        for (int k = 0; k < 8; ++k) f |= p.follow[k][(S >> (8 * k)) & 255];
        S = f & p.accept[sym];
        return (S & p.last) != 0;
    };
    if (step(RE_SYM_BEGIN)) return true;
    for (size_t i = 0; i < n; ++i)
        if (step(text[i])) return true;
    return step(RE_SYM_END);
}


// ---------------------------------------------------------------------------
// Thompson program for the Pike VM (regex_vm.hpp)
// ---------------------------------------------------------------------------
namespace {

struct VmBuilder {
    const std::string& expr;
    VmProgram P;
    uint32_t nsets = 0;
    explicit VmBuilder(const std::string& e) : expr(e) {}
    uint32_t emit(uint8_t op, uint8_t arg = 0, uint8_t x = 0, uint8_t y = 0) {
        if (P.n >= (uint32_t)VM_MAX_INST) unsupported("expression too large for the position-reporting matcher", expr);
        P.inst[P.n] = VmInst{op, arg, x, y};
        return P.n++;
    }
    uint8_t set_of(const ByteSet& s) {
        for (uint32_t k = 0; k < nsets; ++k) {
            bool same = true;
            for (int i = 0; i < 8; ++i) same &= P.sets[k][i] == s[i];
            if (same) return (uint8_t)k;
        }
        if (nsets >= (uint32_t)VM_MAX_SETS) unsupported("too many distinct character classes", expr);
        for (int i = 0; i < 8; ++i) P.sets[nsets][i] = s[i];
        return (uint8_t)nsets++;
    }
    void star(const Node& kid, bool lazy) {  // L1: split L2, L3; L2: e; jmp L1; L3:
        const uint32_t l1 = emit(VM_SPLIT);
        gen(kid);
        emit(VM_JMP, 0, (uint8_t)l1);
        const uint32_t l3 = P.n;
        P.inst[l1].x = (uint8_t)(lazy ? l3 : l1 + 1);
        P.inst[l1].y = (uint8_t)(lazy ? l1 + 1 : l3);
    }
    void quest(const Node& kid, bool lazy) {  // split L1, L2; L1: e; L2:
        const uint32_t l0 = emit(VM_SPLIT);
        gen(kid);
        const uint32_t l2 = P.n;
        P.inst[l0].x = (uint8_t)(lazy ? l2 : l0 + 1);
        P.inst[l0].y = (uint8_t)(lazy ? l0 + 1 : l2);
    }
    void gen(const Node& n) {
        switch (n.t) {
            case NT::Empty: return;
            case NT::Lit: emit(VM_CHAR, set_of(n.set)); return;
            case NT::Begin: emit(VM_BEGIN); return;
            case NT::End: emit(VM_END); return;
            case NT::WordB: emit(VM_WORDB); return;
            case NT::NotWordB: emit(VM_NWORDB); return;
            case NT::Cat: for (auto& k : n.kids) gen(*k); return;
            case NT::Alt: {  // split L1, L2; L1: a; jmp END; L2: split ... (the first alternative is preferred)
                std::vector<uint32_t> jumps;
                for (size_t k = 0; k < n.kids.size(); ++k) {
                    if (k + 1 < n.kids.size()) {
                        const uint32_t sp = emit(VM_SPLIT);
                        gen(*n.kids[k]);
                        jumps.push_back(emit(VM_JMP));
                        P.inst[sp].x = (uint8_t)(sp + 1);
                        P.inst[sp].y = (uint8_t)P.n;
                    } else {
                        gen(*n.kids[k]);
                    }
                }
                for (uint32_t j : jumps) P.inst[j].x = (uint8_t)P.n;
                return;
            }
            case NT::Star: star(*n.kids[0], n.lazy); return;
            case NT::Plus: {  // L1: e; split L1, L3
                const uint32_t l1 = P.n;
                gen(*n.kids[0]);
                const uint32_t sp = emit(VM_SPLIT);
                P.inst[sp].x = (uint8_t)(n.lazy ? sp + 1 : l1);
                P.inst[sp].y = (uint8_t)(n.lazy ? l1 : sp + 1);
                return;
            }
            case NT::Quest: quest(*n.kids[0], n.lazy); return;
            case NT::Repeat: {
                for (int k = 0; k < n.lo; ++k) gen(*n.kids[0]);
                if (n.hi < 0) star(*n.kids[0], n.lazy);
                else {
                    // (e(e(e)?)?)? : every optional copy may end the repetition
                    std::vector<uint32_t> splits;
                    for (int k = n.lo; k < n.hi; ++k) { splits.push_back(emit(VM_SPLIT)); gen(*n.kids[0]); }
                    for (uint32_t sp : splits) {
                        P.inst[sp].x = (uint8_t)(n.lazy ? P.n : sp + 1);
                        P.inst[sp].y = (uint8_t)(n.lazy ? sp + 1 : P.n);
                    }
                }
                return;
            }
            case NT::Group:
                if (n.group == 1) emit(VM_SAVE, 2);
                gen(*n.kids[0]);
                if (n.group == 1) emit(VM_SAVE, 3);
                return;
        }
    }
};

}  // namespace

VmProgram compile_vm(const std::string& expr) {
    Parser ps(expr);
    NodeP root = ps.parse();
    VmBuilder b(expr);
    memset(&b.P, 0, sizeof b.P);
    b.emit(VM_SAVE, 0);
    b.gen(*root);
    b.emit(VM_SAVE, 1);
    b.emit(VM_MATCH);
    b.P.ngroups = (uint32_t)ps.ngroups;
    return b.P;
}

}  // namespace bsk
