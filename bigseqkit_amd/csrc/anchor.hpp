// Record-boundary repair: find the first record start at or after a byte offset.
// This is the job IgnisHPC's PlainFile(path, delim) + ReadFixer do for the
// reference (/root/reference/bigseqkit/helper.go:148-178,
// bigseqkit-lib/helper.go:41-66): every partition must begin on a record.
// One rule, compiled for host (shard cutting) and device (range anchors).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define BSK_HD __host__ __device__ __forceinline__
#else
#ifndef BSK_HD
#define BSK_HD inline
#endif
#endif

namespace bsk {

// first index >= from with buf[idx] == c, else n
BSK_HD uint64_t find_byte(const uint8_t* buf, uint64_t n, uint64_t from, uint8_t c) {
    uint64_t i = from;
    // head: up to 8-byte alignment
    while (i < n && ((uintptr_t)(buf + i) & 7)) {
        if (buf[i] == c) return i;
        ++i;
    }
    const uint64_t rep = 0x0101010101010101ull * c;
    while (i + 8 <= n) {
        uint64_t x = *(const uint64_t*)(buf + i) ^ rep;
        // exact zero-byte detector
        uint64_t z = ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);
        if (z) {
#if defined(__HIP_DEVICE_COMPILE__)
            return i + ((uint64_t)(__ffsll((long long)z) - 1) >> 3);
#else
            return i + ((uint64_t)__builtin_ctzll(z) >> 3);
#endif
        }
        i += 8;
    }
    while (i < n) {
        if (buf[i] == c) return i;
        ++i;
    }
    return n;
}

// first line start at or after `from` (n if none): where a FASTA range may begin even inside a record
BSK_HD uint64_t find_line_start(const uint8_t* buf, uint64_t n, uint64_t from) {
    if (from == 0) return 0;
    if (from >= n) return n;
    const uint64_t j = find_byte(buf, n, from - 1, '\n');
    return j < n ? j + 1 : n;
}

// FASTA: a record starts at a '>' that is the first byte of a line (PARITY.md SPLIT).
BSK_HD uint64_t find_fasta_start(const uint8_t* buf, uint64_t n, uint64_t from) {
    if (from >= n) return n;
    if (from == 0 && buf[0] == '>') return 0;
    uint64_t j = from == 0 ? find_byte(buf, n, 0, '\n') : find_byte(buf, n, from - 1, '\n');
    while (j < n) {
        if (j + 1 < n && buf[j + 1] == '>') return j + 1;
        j = find_byte(buf, n, j + 1, '\n');
    }
    return n;
}

// FASTQ (strict 4-line layout): line p is a record start iff it begins with
// '@', the line two below begins with '+', the sequence line does not begin
// with '+', and len(line p+1) == len(line p+3).  A quality line that begins
// with '@' fails the test because the line two below it is a sequence line.
BSK_HD bool fastq_record_at(const uint8_t* buf, uint64_t n, uint64_t p) {
    if (p >= n || buf[p] != '@') return false;
    uint64_t e0 = find_byte(buf, n, p, '\n');
    if (e0 >= n) return false;
    uint64_t s0 = e0 + 1;
    uint64_t e1 = find_byte(buf, n, s0, '\n');
    if (e1 >= n) return false;
    if (e1 > s0 && buf[s0] == '+') return false;
    uint64_t p0 = e1 + 1;
    if (p0 >= n || buf[p0] != '+') return false;
    uint64_t e2 = find_byte(buf, n, p0, '\n');
    if (e2 >= n) return false;
    uint64_t q0 = e2 + 1;
    uint64_t e3 = find_byte(buf, n, q0, '\n');  // may be n: last line without '\n'
    if (e3 - q0 != e1 - s0) return false;
    if (e3 + 1 < n) {
        uint8_t c = buf[e3 + 1];
        if (c != '@' && c != '\n') return false;
    }
    return true;
}

// `limit` (device side: ANCHOR_SEARCH_BYTES past `from`) bounds the search: text that is not FASTQ at all -- FASTA handed
// over with the FASTQ flag -- holds no record start, and every range boundary of a 10 GB shard walking to the end of the
// buffer kept the GPU busy for a quarter of an hour.  ANCHOR_NONE = nothing found before the limit.
constexpr uint64_t ANCHOR_NONE = ~0ull;
constexpr uint64_t ANCHOR_SEARCH_BYTES = 32ull << 20;  // (several of the longest reads there are)
BSK_HD uint64_t find_fastq_start(const uint8_t* buf, uint64_t n, uint64_t from, uint64_t limit = ~0ull) {
    if (from >= n) return n;
    if (from == 0 && fastq_record_at(buf, n, 0)) return 0;
    uint64_t j = from == 0 ? find_byte(buf, n, 0, '\n') : find_byte(buf, n, from - 1, '\n');
    while (j < n) {
        if (fastq_record_at(buf, n, j + 1)) return j + 1;
        if (j >= limit) return ANCHOR_NONE;
        j = find_byte(buf, n, j + 1, '\n');
    }
    return n;
}

// ---- FASTQ whose sequence / quality text is wrapped over several lines (SeqParser reads it: /root/reference/
// bigseqkit-lib/helper.go:252-269) -- HOST side only: where a file may be cut (bsk_find_record_start, the staging
// pipelines).  The shards themselves are rewritten to four lines per record on the device (capi.cpp).
// A record starts at line p iff p begins with '@', the lines below it up to the first line that begins with '+' hold S > 0
// bytes of bases, and the lines below the '+' line reach EXACTLY S bytes of qualities at a line end -- after which the
// text ends or a line that begins with '@' follows.  A quality line that begins with '@' passes this for one record only
// by accident; the test is therefore repeated on the records that follow (fastq_multiline_record_at).
// Returns the byte after the record's last quality line (n when the text ends without a final line break), 0 = no record.
inline uint64_t fastq_multiline_record_end(const uint8_t* buf, uint64_t n, uint64_t p) {
    if (p >= n || buf[p] != '@') return 0;
    uint64_t e = find_byte(buf, n, p, '\n');
    if (e >= n) return 0;
    uint64_t s = e + 1, S = 0;
    for (;;) {  // sequence lines (none of a real record begins with '@': a candidate that needs one to be read as bases is a
                // quality line followed by the next record's header -- as a place to cut it is passed over)
        if (s >= n) return 0;
        if (buf[s] == '+') break;
        if (buf[s] == '@') return 0;
        e = find_byte(buf, n, s, '\n');
        if (e >= n) return 0;
        S += e - s;
        s = e + 1;
    }
    e = find_byte(buf, n, s, '\n');  // the '+' line
    if (e >= n) return 0;
    s = e + 1;
    uint64_t Q = 0;
    if (S == 0) {  // an empty record: one EMPTY quality line (the grammar asks for a quality line), or the end of the text
        if (s >= n) return n;
        if (buf[s] != '\n') return 0;
        s += 1;
        if (s < n && buf[s] != '@' && buf[s] != '\n') return 0;
        return s;
    }
    while (Q < S) {  // quality lines
        if (s >= n) return 0;
        e = find_byte(buf, n, s, '\n');
        Q += (e < n ? e : n) - s;
        if (e >= n) return Q == S ? n : 0;
        s = e + 1;
    }
    if (Q != S) return 0;
    if (s < n && buf[s] != '@' && buf[s] != '\n') return 0;
    return s;
}
inline bool fastq_multiline_record_at(const uint8_t* buf, uint64_t n, uint64_t p) {
    // three records in a row (or fewer and the end of the text): a quality line that begins with '@' reads as a record of its
    // own only by accident, and the accident has to repeat.  (Text wrapped at one or two bytes per line defeats any such
    // test -- the lines carry no structure; files are wrapped at 50 - 80.)
    uint64_t q = p;
    for (int k = 0; k < 3; ++k) {
        const uint64_t e = fastq_multiline_record_end(buf, n, q);
        if (e == 0) return false;
        q = e;
        while (q < n && buf[q] == '\n') ++q;  // (blank lines at the end of the text)
        if (q >= n) return true;
    }
    return true;
}
// ... and the text BEFORE the candidate must agree: only the first of the three records can be the accidental
// one -- a quality line that begins with '@', then lines read as bases, a quality line that begins with '+', and quality
// lines that happen to add up, followed by a real header; the two records behind it are genuine and confirm nothing.  A
// genuine start is where the record before it ENDS.  Walking back over the lines that begin with '@' (at most `back`
// bytes): one whose record ends exactly at p confirms the candidate; one that reads as records in a row and whose first
// record reaches beyond p refutes it (p lies inside that record).  Text that begins too late to tell (a window into the
// file) leaves the candidate standing, as before.
inline bool fastq_multiline_prev_agrees(const uint8_t* buf, uint64_t n, uint64_t p, uint64_t back = 8ull << 20) {
    if (p == 0 || buf[p - 1] != '\n') return true;
    const uint64_t stop = p > back ? p - back : 0;
    uint64_t q = p - 1;  // the line feed that ends the line before
    while (q > stop && buf[q - 1] == '\n') --q;  // (blank lines before the candidate: the end of the text before it)
    while (q > stop) {
        uint64_t s = q;  // start of the line that ends at q
        while (s > stop && buf[s - 1] != '\n') --s;
        if (s == stop && stop > 0) return true;  // out of look-behind: cannot tell
        if (s == 0 && buf[0] != '@') return true;  // the text begins inside a record (a window): cannot tell
        if (buf[s] == '@') {
            const uint64_t e = fastq_multiline_record_end(buf, n, s);
            if (e != 0 && e <= p) {
                uint64_t g = e;
                while (g < p && buf[g] == '\n') ++g;
                if (g == p) return true;  // the record before ends here
            } else if (e > p && fastq_multiline_record_at(buf, n, s)) {
                return false;  // p lies inside a record that is confirmed by the ones behind it
            }
        }
        if (s == 0) return true;  // (reached the beginning without a verdict)
        q = s - 1;
    }
    return true;
}
// first record start at or after `from`, looking at most `limit` bytes ahead (n = none found)
inline uint64_t find_fastq_start_multiline(const uint8_t* buf, uint64_t n, uint64_t from, uint64_t limit = 64ull << 20) {
    if (from >= n) return n;
    if (from == 0 && fastq_multiline_record_at(buf, n, 0)) return 0;
    uint64_t j = from == 0 ? find_byte(buf, n, 0, '\n') : find_byte(buf, n, from - 1, '\n');
    while (j < n) {
        if (j + 1 < n && buf[j + 1] == '@' && fastq_multiline_record_at(buf, n, j + 1) && fastq_multiline_prev_agrees(buf, n, j + 1)) return j + 1;
        if (j + 1 > from + limit) return n;
        j = find_byte(buf, n, j + 1, '\n');
    }
    return n;
}
// where a host-resident FASTQ text may be cut.  The wrapped reading goes first: on a four-line file it names the same
// record starts as the strict rule (a four-line record is a wrapped one), while the strict rule on a WRAPPED file can take
// four full-width quality lines for a record ('@...', line, '+...', line of the same width, then a header).  The strict
// rule still answers where the wrapped reading finds nothing it can confirm with a second record (a damaged tail).
inline uint64_t find_fastq_cut(const uint8_t* buf, uint64_t n, uint64_t from) {
    if (from >= n) return n;
    const uint64_t wrapped = find_fastq_start_multiline(buf, n, from);
    if (wrapped < n) return wrapped;
    const uint64_t strict = find_fastq_start(buf, n, from);
    return strict == ANCHOR_NONE ? n : strict;
}

// end of the shard once trailing blank lines are dropped: at most one '\n'
// is kept after the last non-newline byte; a buffer of only newlines is empty.
BSK_HD uint64_t effective_end(const uint8_t* buf, uint64_t n) {
    uint64_t e = n;
    while (e >= 2 && buf[e - 1] == '\n' && buf[e - 2] == '\n') --e;
    if (e == 1 && buf[0] == '\n') e = 0;
    return e;
}

}  // namespace bsk
