// Regular expressions WITH positions: a Pike VM (Thompson program, thread lists in priority order) that finds the
// leftmost-first match and the bounds of capture group 1 -- what Go's regexp.FindSubmatch / FindSubmatchIndex return.
// Used where the reference needs more than "matches or not":
//   * custom --id-regexp: ID = FindSubmatch(head)[1]        (/root/reference/bigseqkit-lib/helper.go:362-368)
//   * locate -r with matches of variable length: FindSubmatchIndex in a loop (bigseqkit-lib/locate.go:583-667)
// The search routine is compiled for the host (bsk_create vets programs, CPU tests compare it with std::regex) and for
// the device (one lane per record; thread lists in private memory -- a rare path, not a fast one).
// Syntax: what regex_nfa.hpp parses (RE2 subset), plus capture groups and lazy quantifiers, which matter here.
#pragma once
#include <cstdint>
#include <string>

#if defined(__HIPCC__)
#define BSK_VM_HD __host__ __device__
#else
#define BSK_VM_HD
#endif

namespace bsk {

constexpr int VM_MAX_INST = 64;
constexpr int VM_MAX_SETS = 24;
enum : uint8_t { VM_CHAR = 0, VM_SPLIT = 1, VM_JMP = 2, VM_SAVE = 3, VM_BEGIN = 4, VM_END = 5, VM_MATCH = 6, VM_WORDB = 7, VM_NWORDB = 8 };

struct VmInst { uint8_t op, arg; uint8_t x, y; };  // CHAR: arg = set; SPLIT: x first (preferred), y second; JMP: x; SAVE: arg = slot (0..3)
struct VmProgram {
    uint32_t n = 0;
    uint32_t ngroups = 0;  // capture groups in the expression (group 1 is the only one whose bounds are kept)
    VmInst inst[VM_MAX_INST];
    uint32_t sets[VM_MAX_SETS][8];
};

// throws OptError (unsupported syntax, too many instructions / classes)
VmProgram compile_vm(const std::string& expr);

// leftmost-first match of the program in text[0, n), searching from `from` (^ matches at 0 only, $ at n only).
// caps[0..1] = bounds of the match, caps[2..3] = bounds of group 1 (0xFFFFFFFF when it did not take part).
// (text(i): byte i of the target -- a plain pointer, a wrapped FASTA record, or the reverse complement read backwards)
template <class TextFn>
BSK_VM_HD inline bool vm_search_fn(const VmProgram& P, const TextFn& text, uint32_t n, uint32_t from, uint32_t* caps) {
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    struct Th { uint8_t pc; uint32_t c[4]; };
    Th la[VM_MAX_INST], lb[VM_MAX_INST];
    Th* cl = la;
    Th* nl = lb;
    uint32_t ncl = 0, nnl = 0;
    uint32_t mark[VM_MAX_INST];
    for (uint32_t i = 0; i < P.n; ++i) mark[i] = NONE;
    bool matched = false;
    Th stack[VM_MAX_INST];
    // addthread with an explicit stack: follows JMP / SPLIT / SAVE / assertions in priority order
    auto add = [&](Th* list, uint32_t& cnt, uint8_t pc0, const uint32_t* c0, uint32_t sp) {
        uint32_t top = 0;
        stack[top].pc = pc0;
        for (int k = 0; k < 4; ++k) stack[top].c[k] = c0[k];
        ++top;
        while (top) {
            Th t = stack[--top];
            for (;;) {
                if (mark[t.pc] == sp) break;
                mark[t.pc] = sp;
                const VmInst I = P.inst[t.pc];
                if (I.op == VM_JMP) { t.pc = I.x; continue; }
                if (I.op == VM_SPLIT) {
                    if (top < (uint32_t)VM_MAX_INST) { stack[top] = t; stack[top].pc = I.y; ++top; }  // the second branch waits
                    t.pc = I.x;
                    continue;
                }
                if (I.op == VM_SAVE) { t.c[I.arg] = sp; ++t.pc; continue; }
                if (I.op == VM_BEGIN) { if (sp != 0) break; ++t.pc; continue; }
                if (I.op == VM_END) { if (sp != n) break; ++t.pc; continue; }
                if (I.op == VM_WORDB || I.op == VM_NWORDB) {  // \b / \B: ASCII word characters [0-9A-Za-z_] (RE2)
                    auto word = [](uint8_t ch) { return (ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || ch == '_'; };
                    const bool before = sp > 0 && word(text(sp - 1)), after = sp < n && word(text(sp));
                    if ((before != after) != (I.op == VM_WORDB)) break;
                    ++t.pc;
                    continue;
                }
                list[cnt++] = t;  // CHAR or MATCH
                break;
            }
        }
    };
    const uint32_t fresh[4] = {NONE, NONE, NONE, NONE};
    for (uint32_t sp = from;; ++sp) {
        if (!matched) add(cl, ncl, 0, fresh, sp);  // a new attempt starts here, below every running one
        if (ncl == 0) {
            if (matched || sp >= n) break;
            continue;  // (an expression that can only begin further on, e.g. at the end of the text)
        }
        nnl = 0;
        for (uint32_t i = 0; i < ncl; ++i) {
            const Th t = cl[i];
            const VmInst I = P.inst[t.pc];
            if (I.op == VM_MATCH) {
                for (int k = 0; k < 4; ++k) caps[k] = t.c[k];
                matched = true;
                break;  // threads below this one are cut off
            }
            if (sp < n) {
                const uint8_t ch = text(sp);
                if ((P.sets[I.arg][ch >> 5] >> (ch & 31)) & 1u) {
                    // marks of step sp + 1: distinct from step sp because `mark` holds the step number
                    add(nl, nnl, (uint8_t)(t.pc + 1), t.c, sp + 1);
                }
            }
        }
        Th* tmp = cl; cl = nl; nl = tmp;
        ncl = nnl;
        if (sp >= n) break;
    }
    return matched;
}

BSK_VM_HD inline bool vm_search(const VmProgram& P, const uint8_t* text, uint32_t n, uint32_t from, uint32_t* caps) {
    return vm_search_fn(P, [text](uint32_t i) { return text[i]; }, n, from, caps);
}

}  // namespace bsk
