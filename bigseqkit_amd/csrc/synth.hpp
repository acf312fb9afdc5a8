// Deterministic synthetic FASTA/FASTQ.  Byte k of record
// i is a pure function of (seed, flags, i, k), so the same file can be produced
// on the host (tests, cpu baseline) and directly in HBM (bench) -- compiled for
// both sides from this one header.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define BSK_HD __host__ __device__ __forceinline__
#else
#define BSK_HD inline
#endif

namespace bsk {
namespace synth {

constexpr int KIND_FASTQ150 = 0, KIND_FASTA1K = 1, KIND_FASTA5K_CDS = 2, KIND_FASTA5K_VAR = 3;
constexpr unsigned FLAG_MOTIF = 1u, FLAG_DUPS = 2u;
constexpr uint32_t REC_FASTQ150 = 317, REC_FASTA1K = 1027, REC_FASTA5K = 5107;

BSK_HD uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
BSK_HD uint64_t h3(uint64_t seed, uint64_t i, uint64_t j) { return mix64(mix64(seed ^ (i * 0xD6E8FEB86659FD93ull)) + j); }

// (KIND_FASTA5K_VAR: records differ in size -- 0 here, var_record_bytes(i) / var_offset(i) below)
BSK_HD uint32_t record_bytes(int kind) {
    return kind == KIND_FASTQ150 ? REC_FASTQ150 : kind == KIND_FASTA1K ? REC_FASTA1K : kind == KIND_FASTA5K_CDS ? REC_FASTA5K : 0u;
}

BSK_HD uint8_t digit(uint64_t v, int pos_from_right) {
    for (int k = 0; k < pos_from_right; ++k) v /= 10;
    return (uint8_t)('0' + (v % 10));
}

BSK_HD uint8_t base_of(uint64_t seed, uint64_t i, uint32_t b) {
    const uint64_t h = h3(seed, i, 0x100 + (b >> 5));
    return (uint8_t)("ACGT"[(h >> ((b & 31) * 2)) & 3]);
}

// C5: record i with i % 5 == 4 copies the sequence of record i-1-(h(i)%1000)
// (clamped to >= 0 and never itself a copy: sources with src % 5 == 4 step back once).
BSK_HD uint64_t seq_source(uint64_t seed, unsigned flags, uint64_t i) {
    if (!(flags & FLAG_DUPS) || i % 5 != 4) return i;
    uint64_t back = 1 + (h3(seed, i, 0x300) % 1000);
    uint64_t src = i >= back ? i - back : 0;
    if (src % 5 == 4) src -= 1;  // src >= 4 here, so no underflow
    return src;
}

// FASTQ-150:  "@S%010d\n" (13) + 150 bases + "\n+\n" + 150 quals + "\n" = 317 B
BSK_HD uint8_t fastq150_byte(uint64_t seed, unsigned flags, uint64_t i, uint32_t k) {
    if (k < 13) {
        if (k == 0) return '@';
        if (k == 1) return 'S';
        if (k == 12) return '\n';
        return digit(i, 11 - (int)k);
    }
    if (k < 163) {
        uint32_t b = k - 13;
        if (flags & FLAG_MOTIF) {
            // C3: motif ACGTTGCAAGCT on '+' strand when i%100==0, its reverse
            // complement AGCTTGCAACGT when i%100==50, at a uniform position.
            uint64_t m = i % 100;
            if (m == 0 || m == 50) {
                uint32_t pos = (uint32_t)(h3(seed, i, 0x400) % 139);  // 150-12+1
                if (b >= pos && b < pos + 12) return (uint8_t)((m == 0 ? "ACGTTGCAAGCT" : "AGCTTGCAACGT")[b - pos]);
            }
        }
        return base_of(seed, seq_source(seed, flags, i), b);
    }
    if (k == 163) return '\n';
    if (k == 164) return '+';
    if (k == 165) return '\n';
    if (k < 316) {
        uint32_t q = k - 166;
        const uint64_t h = h3(seed, i, 0x200 + (q >> 3));
        uint32_t byte = (uint32_t)(h >> ((q & 7) * 8)) & 0xFF;
        return (uint8_t)('#' + ((byte * 39) >> 8));  // '#'..'I' == Phred 2..40
    }
    return '\n';
}

// FASTA-1k: ">r%07d\n" (10) + 1000 bases in 60-column lines (17 lines) = 1027 B
// (BASELINE.md quotes 1026 B; its header count is off by one.)
BSK_HD uint8_t fasta1k_byte(uint64_t seed, uint64_t i, uint32_t k) {
    if (k < 10) {
        if (k == 0) return '>';
        if (k == 1) return 'r';
        if (k == 9) return '\n';
        return digit(i, 8 - (int)k);
    }
    uint32_t r = k - 10, line = r / 61, col = r % 61;
    if (line < 16) {
        if (col == 60) return '\n';
        return base_of(seed, i, line * 60 + col);
    }
    // last line: 40 bases + '\n'
    if (col == 40) return '\n';
    return base_of(seed, i, 960 + col);
}

// FASTA-5k CDS: ">cds%08d len=5001\n" (22) + ATG + 1665 sense codons + TAA in
// 60-column lines (83 full + 21) = 22 + 5001 + 84 = 5107 B
BSK_HD uint8_t cds_sense(uint64_t seed, uint64_t i, uint32_t b);
BSK_HD uint8_t cds_base(uint64_t seed, uint64_t i, uint32_t b) {
    if (b < 3) return (uint8_t)"ATG"[b];
    if (b >= 4998) return (uint8_t)"TAA"[b - 4998];
    return cds_sense(seed, i, b);
}
BSK_HD uint8_t cds_sense(uint64_t seed, uint64_t i, uint32_t b) {  // base b >= 3 of the run of sense codons behind ATG
    uint32_t c = (b - 3) / 3, ph = (b - 3) % 3;
    const uint64_t h = h3(seed, i, 0x500 + (c / 10));
    uint32_t cod = (uint32_t)(h >> ((c % 10) * 6)) & 63;  // b1*16+b2*4+b3 over "ACGT"
    uint32_t b1 = cod >> 4, b2 = (cod >> 2) & 3, b3 = cod & 3;
    // stops TAA(3,0,0) TAG(3,0,2) TGA(3,2,0) -> first base becomes 'C'
    if (b1 == 3 && ((b2 == 0 && (b3 == 0 || b3 == 2)) || (b2 == 2 && b3 == 0))) b1 = 1;
    return (uint8_t)"ACGT"[ph == 0 ? b1 : ph == 1 ? b2 : b3];
}
BSK_HD uint8_t fasta5k_byte(uint64_t seed, uint64_t i, uint32_t k) {
    if (k < 22) {
        const char* pre = ">cds";
        if (k < 4) return (uint8_t)pre[k];
        if (k < 12) return digit(i, 11 - (int)k);
        const char* suf = " len=5001\n";
        return (uint8_t)suf[k - 12];
    }
    uint32_t r = k - 22, line = r / 61, col = r % 61;
    if (line < 83) {
        if (col == 60) return '\n';
        return cds_base(seed, i, line * 60 + col);
    }
    if (col == 21) return '\n';
    return cds_base(seed, i, 4980 + col);
}

// FASTA-5k CDS with records that do NOT all look alike (round 5: the second C4 leg of bench.py -- `translate` off its
// uniform-layout path): ">cds%d len=%d\n" with the record number UNPADDED (header lengths 15 .. 22), and one record in a
// hundred three bases shorter (i % 100 == 37: 4 998) or longer (i % 100 == 73: 5 004) than the 5 001 of the others; ATG +
// sense codons + TAA in 60-column lines (84 lines in all three cases).  Record sizes vary, so the file offset of record
// i is a sum -- in closed form (digits of the numbers below i by decades, the two residue classes by division), which
// keeps "byte x of the file is a function of (seed, x)" and lets any slice be produced anywhere.
BSK_HD uint32_t ndigits(uint64_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; ++d; } return d; }
BSK_HD uint64_t digits_below(uint64_t i) {  // sum of ndigits(j) over j < i
    uint64_t s = 0, lo = 0, p = 10;
    uint32_t d = 1;
    for (;;) {
        const uint64_t hi = i < p ? i : p;
        if (hi > lo) s += (hi - lo) * d;
        if (i <= p) break;
        lo = p; p *= 10; ++d;
    }
    return s;
}
BSK_HD uint32_t var_len(uint64_t i) { const uint64_t m = i % 100; return m == 37 ? 4998u : (m == 73 ? 5004u : 5001u); }
BSK_HD uint32_t var_record_bytes(uint64_t i) { return 14u + ndigits(i) + var_len(i) + 84u; }
BSK_HD uint64_t var_offset(uint64_t i) {  // file offset of record i == bytes of the records below it
    const uint64_t c37 = (i + 62) / 100, c73 = (i + 26) / 100;  // j < i with j % 100 == 37 / 73
    return i * (uint64_t)(14 + 5001 + 84) + digits_below(i) + 3 * c73 - 3 * c37;
}
BSK_HD uint64_t var_record_at(uint64_t x) {  // the record that holds file byte x
    uint64_t lo = x / 5121, hi = x / 5099 + 1;  // (a record has 5 096 + digits .. 5 102 + digits bytes, digits <= 19)
    while (lo < hi) {  // last i with var_offset(i) <= x
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (var_offset(mid) <= x) lo = mid; else hi = mid - 1;
    }
    return lo;
}
BSK_HD uint8_t cds_base_len(uint64_t seed, uint64_t i, uint32_t b, uint32_t L) {
    if (b < 3) return (uint8_t)"ATG"[b];
    if (b >= L - 3) return (uint8_t)"TAA"[b - (L - 3)];
    return cds_sense(seed, i, b);
}
BSK_HD uint8_t fasta5k_var_byte(uint64_t seed, uint64_t i, uint32_t k) {
    const uint32_t d = ndigits(i), L = var_len(i);
    if (k < 14u + d) {
        if (k < 4) return (uint8_t)">cds"[k];
        if (k < 4 + d) return digit(i, (int)(d - 1 - (k - 4)));
        const uint32_t r = k - 4 - d;
        if (r < 5) return (uint8_t)" len="[r];
        if (r < 9) return digit(L, (int)(8 - r));
        return '\n';
    }
    const uint32_t r = k - 14u - d, line = r / 61, col = r % 61;
    if (line < 83) {
        if (col == 60) return '\n';
        return cds_base_len(seed, i, line * 60 + col, L);
    }
    if (col == L - 4980) return '\n';
    return cds_base_len(seed, i, 4980 + col, L);
}

BSK_HD uint8_t byte_at(int kind, uint64_t seed, unsigned flags, uint64_t i, uint32_t k) {
    if (kind == KIND_FASTQ150) return fastq150_byte(seed, flags, i, k);
    if (kind == KIND_FASTA1K) return fasta1k_byte(seed, i, k);
    if (kind == KIND_FASTA5K_VAR) return fasta5k_var_byte(seed, i, k);
    return fasta5k_byte(seed, i, k);
}

}  // namespace synth
}  // namespace bsk
