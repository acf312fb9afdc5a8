// ============================================================================
// ops_translate.hip -- Translate.Call (/root/reference/bigseqkit-lib/translate.go:66-145)
// with CodonTable.Translate (shenwei356/bio, not in tree) on the record table.
// One element per (record, frame) (PARITY.md Q5).  Bases are mapped to 4-bit IUPAC
// codes (A=1 C=2 G=4 T/U=8, unions for ambiguity codes); complement is a 4-bit
// reversal; the amino acid of any (possibly ambiguous) codon is ONE lookup in a
// 4 KiB table that every block stages in LDS.  Negative frames read the forward
// text backwards -- the reverse complement is never materialised.
// ============================================================================
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "ops_translate.hpp"
#include "text_dev.hpp"

namespace bsk {

namespace {

constexpr int GROUP = 16;

__device__ __forceinline__ uint32_t iupac_code(uint8_t c) {
    switch (c | 0x20) {  // ASCII lower-case fold
        case 'a': return 1; case 'c': return 2; case 'g': return 4; case 't': case 'u': return 8;
        case 'r': return 5; case 'y': return 10; case 's': return 6; case 'w': return 9;
        case 'k': return 12; case 'm': return 3; case 'b': return 14; case 'd': return 13;
        case 'h': return 11; case 'v': return 7; case 'n': return 15;
        default: return 0;
    }
}
__device__ __forceinline__ uint32_t comp_code(uint32_t c) {  // A<->T, C<->G on the bit set
    return ((c & 1u) << 3) | ((c & 2u) << 1) | ((c & 4u) >> 1) | ((c & 8u) >> 3);
}

// IUPAC-coded codon index of amino acid j of `frame`
__device__ __forceinline__ uint32_t codon_index(const Text& T, int frame, uint32_t j) {
    uint32_t c1, c2, c3;
    if (frame > 0) {
        const uint32_t p = (uint32_t)(frame - 1) + 3u * j;
        c1 = iupac_code(T.at(p)); c2 = iupac_code(T.at(p + 1)); c3 = iupac_code(T.at(p + 2));
    } else {
        const uint32_t p = (uint32_t)(-frame - 1) + 3u * j;  // position in the reverse complement
        c1 = comp_code(iupac_code(T.at(T.L - 1 - p)));
        c2 = comp_code(iupac_code(T.at(T.L - 2 - p)));
        c3 = comp_code(iupac_code(T.at(T.L - 3 - p)));
    }
    if (c1 == 0 || c2 == 0 || c3 == 0) return 0xFFFFFFFFu;  // not an IUPAC codon
    return (c1 << 8) | (c2 << 4) | c3;
}

__device__ __forceinline__ uint32_t num_aa(uint32_t L, int frame) {
    const uint32_t f = (uint32_t)(frame > 0 ? frame : -frame) - 1u;
    return L >= f + 3u ? (L - f) / 3u : 0u;
}

// amino acid j after the unknown / init-codon / clean rules; 0 = unknown codon without -x
__device__ __forceinline__ uint8_t aa_at(const Text& T, const TranslateParams& P, const uint8_t* s_codon, int frame,
                                         uint32_t j) {
    const uint32_t ci = codon_index(T, frame, j);
    uint8_t aa = ci == 0xFFFFFFFFu ? 0 : s_codon[ci];
    if (aa == 0) {
        if (!P.allow_unknown) return 0;
        aa = 'X';
    }
    if (j == 0 && P.init_m && ci != 0xFFFFFFFFu && P.start[ci]) aa = 'M';
    if (P.clean && aa == '*') aa = 'X';
    return aa;
}

__device__ __forceinline__ uint32_t dec_len(int v) {
    uint32_t n = v < 0 ? 1u : 0u;
    uint32_t a = (uint32_t)(v < 0 ? -v : v);
    do { ++n; a /= 10; } while (a);
    return n;
}

// header of an element: ">Name" or ">ID_frame=N Desc" (translate.go:133-137), without the '\n'
__device__ uint32_t header_len(const RecordTable& t, uint64_t rec, const uint8_t* h, uint32_t hl, const TranslateParams& P, int frame) {
    if (!P.append_frame) return 1 + hl;
    uint32_t ioff, doff;
    const uint32_t il = id_span_rec(t, rec, h, hl, P.id_mode, &ioff);
    const uint32_t dl = desc_of(h, hl, P.id_mode, il, &doff);
    return 1 + il + 7 + dec_len(frame) + 1 + dl;  // '>' id "_frame=" N ' ' desc
}

__global__ __launch_bounds__(256) void k_translate_size(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt,
                                                        TranslateParams P, uint32_t* __restrict__ out_len,
                                                        uint64_t* __restrict__ status) {
    __shared__ uint8_t s_codon[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_codon[i] = P.codon[i];
    __syncthreads();
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.n * (uint64_t)P.nframes) return;
    const uint64_t i = e / (uint64_t)P.nframes;
    const int frame = P.frames[e % (uint64_t)P.nframes];
    const Text T = text_of(buf, t, tt, i);
    uint32_t naa = num_aa(T.L, frame);
    if (P.trim) {  // bytes.TrimRight(aas, "X*"): walk back over the tail
        while (naa > 0) {
            const uint8_t aa = aa_at(T, P, s_codon, frame, naa - 1);
            if (aa == 'X' || aa == '*') --naa;
            else break;  // (an unknown codon is reported by the emit pass)
        }
    }
    const uint32_t lh = t.l_head[i];
    uint32_t n = header_len(t, i, buf + t.start[i] + 1, lh > 0 ? lh - 1 : 0, P, frame) + 1;
    n += naa + ((P.line_width > 0 && naa > 0) ? (naa - 1) / (uint32_t)P.line_width : 0u);
    n += 1;  // FileStore's newline after the element
    out_len[e] = n;
    (void)status;
}

__device__ __forceinline__ uint32_t put_dec(uint8_t* o, int v) {
    uint32_t n = 0;
    if (v < 0) { o[n++] = '-'; v = -v; }
    char tmp[12];
    int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) o[n++] = (uint8_t)tmp[--k];
    return n;
}

__global__ __launch_bounds__(256) void k_translate_emit(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt,
                                                        TranslateParams P, const uint32_t* __restrict__ out_len,
                                                        const uint64_t* __restrict__ out_off,
                                                        uint8_t* __restrict__ out, uint64_t* __restrict__ status) {
    __shared__ uint8_t s_codon[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_codon[i] = P.codon[i];
    __syncthreads();
    const uint64_t e = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
    const uint32_t gl = threadIdx.x % GROUP;
    if (e >= t.n * (uint64_t)P.nframes) return;
    const uint64_t i = e / (uint64_t)P.nframes;
    const int frame = P.frames[e % (uint64_t)P.nframes];
    const Text T = text_of(buf, t, tt, i);
    const uint32_t n = out_len[e];
    uint8_t* o = out + out_off[e];
    const uint8_t* h = buf + t.start[i] + 1;
    const uint32_t lh = t.l_head[i];
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    uint32_t hdr = 0;
    if (gl == 0) {
        o[hdr++] = '>';
        if (!P.append_frame) {
            for (uint32_t k = 0; k < hl; ++k) o[hdr++] = h[k];
        } else {
            uint32_t ioff, doff;
            const uint32_t il = id_span_rec(t, i, h, hl, P.id_mode, &ioff);
            const uint32_t dl = desc_of(h, hl, P.id_mode, il, &doff);
            for (uint32_t k = 0; k < il; ++k) o[hdr++] = h[ioff + k];
            const char* f = "_frame=";
            for (int k = 0; k < 7; ++k) o[hdr++] = (uint8_t)f[k];
            hdr += put_dec(o + hdr, frame);
            o[hdr++] = ' ';
            for (uint32_t k = 0; k < dl; ++k) o[hdr++] = h[doff + k];
        }
        o[hdr++] = '\n';
        o[n - 1] = '\n';
    }
    const uint32_t H = header_len(t, i, h, hl, P, frame) + 1;
    // body: wrapped amino acids; the trimmed length follows from the element size
    const uint32_t body = n - H - 1;
    const uint32_t w1 = P.line_width > 0 ? (uint32_t)P.line_width + 1u : 0u;
    uint32_t err = 0;
    for (uint32_t x = gl; x < body; x += GROUP) {
        uint32_t j = x;
        bool nl = false;
        if (w1) {
            const uint32_t line = x / w1, col = x - line * w1;
            if (col == w1 - 1) nl = true;
            j = line * (w1 - 1) + col;
        }
        uint8_t c = '\n';
        if (!nl) {
            c = aa_at(T, P, s_codon, frame, j);
            if (c == 0) { err = ERR_UNKNOWN_CODON; c = 'X'; }
        }
        o[H + x] = c;
    }
    // codons dropped by --trim are still translated by the reference (errors included)
    if (P.trim && !P.allow_unknown) {
        const uint32_t total = num_aa(T.L, frame);
        const uint32_t kept = body - ((w1 && body) ? (body - 1) / w1 : 0u);
        for (uint32_t j = kept + gl; j < total; j += GROUP)
            if (aa_at(T, P, s_codon, frame, j) == 0) err = ERR_UNKNOWN_CODON;
    }
    if (err) atomicOr((unsigned long long*)&status[0], (unsigned long long)err);
}


// Chromosome-sized records: every output byte of an element is computed from its position (aa_at), so an element's
// body is cut into chunks of LONG_BODY bytes, one block each: grid = (chunks, long records x frames).
constexpr uint32_t LONG_BODY = 16u * 1024u;
__global__ __launch_bounds__(256) void k_translate_long(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt,
                                                        TranslateParams P, const uint32_t* __restrict__ out_len,
                                                        const uint64_t* __restrict__ out_off,
                                                        uint8_t* __restrict__ out, uint64_t* __restrict__ status) {
    __shared__ uint8_t s_codon[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_codon[i] = P.codon[i];
    __syncthreads();
    const uint64_t i = P.long_list[blockIdx.y / (uint32_t)P.nframes];
    const int k = (int)(blockIdx.y % (uint32_t)P.nframes);
    const int frame = P.frames[k];
    const uint64_t e = i * (uint64_t)P.nframes + (uint64_t)k;
    const Text T = text_of(buf, t, tt, i);
    const uint32_t n = out_len[e];
    uint8_t* o = out + out_off[e];
    const uint8_t* h = buf + t.start[i] + 1;
    const uint32_t lh = t.l_head[i];
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    const uint32_t H = header_len(t, i, h, hl, P, frame) + 1;
    const uint32_t body = n - H - 1;
    const uint32_t x_lo = blockIdx.x * LONG_BODY;
    if (x_lo >= body && blockIdx.x != 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // header and the element's final newline
        uint32_t hdr = 0;
        o[hdr++] = '>';
        if (!P.append_frame) {
            for (uint32_t q = 0; q < hl; ++q) o[hdr++] = h[q];
        } else {
            uint32_t ioff, doff;
            const uint32_t il = id_span_rec(t, i, h, hl, P.id_mode, &ioff);
            const uint32_t dl = desc_of(h, hl, P.id_mode, il, &doff);
            for (uint32_t q = 0; q < il; ++q) o[hdr++] = h[ioff + q];
            const char* f = "_frame=";
            for (int q = 0; q < 7; ++q) o[hdr++] = (uint8_t)f[q];
            hdr += put_dec(o + hdr, frame);
            o[hdr++] = ' ';
            for (uint32_t q = 0; q < dl; ++q) o[hdr++] = h[doff + q];
        }
        o[hdr++] = '\n';
        o[n - 1] = '\n';
    }
    const uint32_t w1 = P.line_width > 0 ? (uint32_t)P.line_width + 1u : 0u;
    const uint32_t x_hi = x_lo + LONG_BODY < body ? x_lo + LONG_BODY : body;
    uint32_t err = 0;
    for (uint32_t x = x_lo + threadIdx.x; x < x_hi; x += blockDim.x) {
        uint32_t j = x;
        bool nl = false;
        if (w1) {
            const uint32_t line = x / w1, col = x - line * w1;
            if (col == w1 - 1) nl = true;
            j = line * (w1 - 1) + col;
        }
        uint8_t c = '\n';
        if (!nl) {
            c = aa_at(T, P, s_codon, frame, j);
            if (c == 0) { err = ERR_UNKNOWN_CODON; c = 'X'; }
        }
        o[H + x] = c;
    }
    // codons dropped by --trim are still translated by the reference (errors included)
    if (blockIdx.x == 0 && P.trim && !P.allow_unknown) {
        const uint32_t total = num_aa(T.L, frame);
        const uint32_t kept = body - ((w1 && body) ? (body - 1) / w1 : 0u);
        for (uint32_t j = kept + threadIdx.x; j < total; j += blockDim.x)
            if (aa_at(T, P, s_codon, frame, j) == 0) err = ERR_UNKNOWN_CODON;
    }
    if (err) atomicOr((unsigned long long*)&status[0], (unsigned long long)err);
}

// ---------------------------------------------------------------------------
// k_translate_frames<G>: G lanes per record, every base read once.
// A window is G x 48 bases (a multiple of 3, so base k of a lane always belongs to forward
// frame (k % 3) + 1).  The window's RAW text (bases + the newlines of a wrapped FASTA record)
// is copied once into LDS with coalesced 16-byte loads; each lane then walks its 48 bases,
// skipping newlines by column counting, mapping bytes to 4-bit IUPAC codes through an LDS
// table.  Every position q yields two amino acids from the same three codes:
//   forward frame (q % 3) + 1, index q / 3              -> s_fw[c0 c1 c2]
//   reverse frame -(((L-3-q) % 3) + 1), index (L-3-q)/3 -> s_rc[c0 c1 c2]  (table of the
//   reverse-complemented codon, so the reverse complement is never formed)
// Output offsets (index + line breaks) are advanced incrementally: no division per residue.
// ---------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void k_translate_frames(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt,
                                                          TranslateParams P, const uint32_t* __restrict__ out_len,
                                                          const uint64_t* __restrict__ out_off,
                                                          uint8_t* __restrict__ out, uint64_t* __restrict__ status) {
    constexpr int WIN = G * 48;
    constexpr int RAWCAP = WIN + WIN / 16 + 64;  // line width >= 16 (narrower records are linearised)
    constexpr int NG = 256 / G;
    __shared__ uint8_t s_fw[4096];
    __shared__ uint8_t s_rc[4096];
    __shared__ uint8_t s_iu[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[NG][RAWCAP];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) { s_fw[i] = P.codon[i]; s_rc[i] = P.codon_rc[i]; }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_iu[i] = P.iupac[i];
    __syncthreads();
    const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const uint32_t gl = threadIdx.x % G;
    if (g >= t.n) return;  // no block-level barrier below
    if (P.long_thresh && t.l_seq[g] >= P.long_thresh) return;
    if (P.long_thresh && t.l_seq[g] >= P.long_thresh) return;  // launch_translate_long handles this record
    uint8_t* raw = s_raw[threadIdx.x / G];
    const Text T = text_of(buf, t, tt, g);
    const uint32_t L = T.L;
    const uint32_t W = T.W;
    // bytes of text that may be read: bases plus one newline per full line
    const uint32_t raw_total = W ? L + (L ? (L - 1) / W : 0u) : L;
    const uint8_t* h = buf + t.start[g] + 1;
    const uint32_t lh = t.l_head[g];
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    const uint32_t lw = P.line_width > 0 ? (uint32_t)P.line_width : 0u;

    // element slots: forward frames 1..3 -> fb[0..2], reverse frames -1..-3 -> rbq[0..2]
    uint8_t* fb[3] = {nullptr, nullptr, nullptr};
    uint8_t* rbq[3] = {nullptr, nullptr, nullptr};
    uint32_t fk[3] = {0, 0, 0}, rkq[3] = {0, 0, 0};  // amino acids kept (after --trim) per slot
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (k >= P.nframes) break;
        const int frame = P.frames[k];
        const uint64_t e = g * (uint64_t)P.nframes + (uint64_t)k;
        const uint32_t n = out_len[e];
        uint8_t* o = out + out_off[e];
        const uint32_t H = header_len(t, g, h, hl, P, frame) + 1;
        const uint32_t body = n - H - 1;
        const uint32_t kept = body - (lw ? body / (lw + 1) : 0u);
        // header and the element's final newline
        if (!P.append_frame) {
            for (uint32_t x = gl; x < H; x += G) o[x] = x == 0 ? (uint8_t)'>' : (x == H - 1 ? (uint8_t)'\n' : h[x - 1]);
        } else if (gl == 0) {
            uint32_t hdr = 0, ioff, doff;
            o[hdr++] = '>';
            const uint32_t il = id_span_rec(t, g, h, hl, P.id_mode, &ioff);
            const uint32_t dl = desc_of(h, hl, P.id_mode, il, &doff);
            for (uint32_t q = 0; q < il; ++q) o[hdr++] = h[ioff + q];
            const char* fs = "_frame=";
            for (int q = 0; q < 7; ++q) o[hdr++] = (uint8_t)fs[q];
            hdr += put_dec(o + hdr, frame);
            o[hdr++] = ' ';
            for (uint32_t q = 0; q < dl; ++q) o[hdr++] = h[doff + q];
            o[hdr++] = '\n';
        }
        if (gl == 0) o[n - 1] = '\n';
        // line breaks inside the body: one every lw amino acids
        if (lw) for (uint32_t x = lw + gl * (lw + 1); x < body; x += G * (lw + 1)) o[H + x] = '\n';
        uint8_t* bodyp = o + H;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (frame == c + 1) { fb[c] = bodyp; fk[c] = kept; }
            if (frame == -(c + 1)) { rbq[c] = bodyp; rkq[c] = kept; }
        }
    }
    // a base at position q (class c = q % 3) feeds reverse frame -(((L - 3 - q) % 3) + 1):
    // rotate the reverse slots once per record so that the class index is static in the loop
    const uint32_t Lm = (L % 3u);  // (L - 3) mod 3 == L mod 3
    uint8_t* rb[3];
    uint32_t rk[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t s = (Lm + 3u - (uint32_t)c) % 3u;  // reverse slot of class c
        rb[c] = s == 0 ? rbq[0] : (s == 1 ? rbq[1] : rbq[2]);
        rk[c] = s == 0 ? rkq[0] : (s == 1 ? rkq[1] : rkq[2]);
    }
    uint32_t err = 0;
    // Lane l of the group owns codon slot l of every 3G-base step: bases q, q+1, q+2 with
    // q = step_base + 3l give one residue to each forward frame (index step_base/3 + l) and one
    // to each reverse slot, so consecutive lanes write consecutive output bytes.
    // Per-lane constants for walking a wrapped FASTA line without divisions in the loop:
    const uint32_t l3 = gl * 3u;
    const uint32_t la = W ? l3 / W : 0u, lb = W ? l3 % W : 0u;  // 3l = la * W + lb
    const uint32_t stepb = 3u * G;                                  // bases per step
    const uint32_t sd = W ? stepb / W : 0u, sm = W ? stepb % W : 0u;
    // forward cursor of this lane: residue index, line-break offset, column (shared by the 3 frames)
    uint32_t fj = gl, fo = 0, fc = gl;
    if (lw) { fo = fj / lw; fc = fj - fo * lw; }
    const uint32_t gd = lw ? (uint32_t)G / lw : 0u, gm = lw ? (uint32_t)G % lw : 0u;
    // reverse cursors (one per class): residue index Rc - l - G * step; valid while >= 0
    int64_t rj[3];
    uint32_t ro[3], rcc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rj[c] = L >= 3u + (uint32_t)c ? (int64_t)((L - 3u - (uint32_t)c) / 3u) - (int64_t)gl : -1;
        ro[c] = 0; rcc[c] = 0;
        if (lw && rj[c] >= 0) { ro[c] = (uint32_t)rj[c] / lw; rcc[c] = (uint32_t)rj[c] - ro[c] * lw; }
    }
    for (uint32_t q0 = 0; q0 < L; q0 += WIN) {
        // ---- stage the window's raw text in LDS (coalesced 16-byte loads)
        const uint32_t rawbase = W ? q0 + q0 / W : q0;
        uint32_t span = raw_total - rawbase;
        if (span > (uint32_t)RAWCAP) span = RAWCAP;
        for (uint32_t off = gl * 16u; off < span; off += G * 16u) {
            if (off + 16u <= span) {
                uint4 v;
                __builtin_memcpy(&v, T.p + rawbase + off, 16);
                *reinterpret_cast<uint4*>(raw + off) = v;
            } else {
                for (uint32_t bq = off; bq < span; ++bq) raw[bq] = T.p[rawbase + bq];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- 16 steps of 3G bases
        uint32_t colbase = W ? q0 % W : 0u, lines = 0;  // column of the step's first base, newlines since q0
        for (uint32_t it = 0; it < 16u; ++it) {
            const uint32_t sb = q0 + it * stepb;  // first base of the step
            if (sb >= L) break;
            const uint32_t q = sb + l3;
            // raw position and column of base q
            uint32_t col = colbase + lb, rp = (sb - q0) + lines + l3 + la;
            if (W && col >= W) { col -= W; ++rp; }
            uint32_t cd[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint32_t adj = (W && col + (uint32_t)k >= W) ? 1u : 0u;  // W >= 16 > 5: at most one break
                cd[k] = (q + (uint32_t)k < L) ? (uint32_t)s_iu[raw[rp + (uint32_t)k + adj]] : 0u;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (q + (uint32_t)c + 2u < L) {
                    const uint32_t c0 = cd[c], c1 = cd[c + 1], c2 = cd[c + 2];
                    const bool bad = c0 == 0 || c1 == 0 || c2 == 0;
                    const uint32_t idx = (c0 << 8) | (c1 << 4) | c2;
                    if (fb[c]) {  // forward frame c + 1, residue fj
                        uint8_t aa = bad ? (uint8_t)0 : s_fw[idx];
                        if (aa == 0) { if (P.allow_unknown) aa = 'X'; else err = ERR_UNKNOWN_CODON; }
                        if (fj == 0 && P.init_m && !bad && P.start[idx]) aa = 'M';
                        if (P.clean && aa == '*') aa = 'X';
                        if (fj < fk[c]) fb[c][fj + fo] = aa;
                    }
                    if (rb[c]) {  // reverse slot of class c, residue rj[c]
                        const uint32_t j = (uint32_t)rj[c];
                        uint8_t aa = bad ? (uint8_t)0 : s_rc[idx];
                        if (aa == 0) { if (P.allow_unknown) aa = 'X'; else err = ERR_UNKNOWN_CODON; }
                        if (j == 0 && P.init_m && !bad && P.start_rc[idx]) aa = 'M';
                        if (P.clean && aa == '*') aa = 'X';
                        if (j < rk[c]) rb[c][j + ro[c]] = aa;
                    }
                }
            }
            // advance the cursors by one step (G residues per frame)
            fj += G;
            if (lw) { fo += gd; fc += gm; if (fc >= lw) { fc -= lw; ++fo; } }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rj[c] -= G;
                if (lw && rj[c] >= 0) {
                    ro[c] -= gd;
                    if (rcc[c] < gm) { rcc[c] += lw - gm; --ro[c]; } else rcc[c] -= gm;
                }
            }
            if (W) { lines += sd; colbase += sm; if (colbase >= W) { colbase -= W; ++lines; } }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (err) atomicOr((unsigned long long*)&status[0], (unsigned long long)err);
}

// ---------------------------------------------------------------------------
// k_translate_frames4<G>: like k_translate_frames, but a lane owns FOUR consecutive codon slots (12 bases) of a
// step, so every frame gets four consecutive residues per lane and step.  The previous kernel was VALU-bound
// (~36 vector instructions per residue, profiles/r01d_ops_kernel_breakdown.json + SQ counters); here
//   * the lane fetches its 16 raw bytes with five LDS dword reads + v_alignbyte, drops the (at most one) newline of a
//     wrapped FASTA line with four v_bfi merges, and maps 14 bytes to IUPAC codes through the LDS table;
//   * four residues are packed into one dword and leave as ONE store unless a line break falls between them
//     (then four byte stores) -- interior steps carry no bounds checks at all;
//   * --clean and -x are folded into the LDS copies of the codon tables, an unknown codon is detected from the AND of
//     all packed dwords once per record, -M patches residue 0 of each frame after the loop.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t byte_of(const uint32_t (&w)[4], int i) { return (w[i >> 2] >> (8 * (i & 3))) & 0xFFu; }

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

// four residues at `at` = position of the first one; `first` = residues before the line break (>= 4: none inside).
// A break inside the four is written WITH them: v_perm splices the '\n' into the dword and only the fourth residue needs
// a byte store of its own (the first version fell back to four byte stores, each with its 64-bit address: with 60-column
// lines one lane in fifteen straddles a break, so every wave paid for both paths -- 150 of its ~500 vector instructions
// per step).
__device__ __forceinline__ void put4(uint8_t* at, uint32_t packed, uint32_t first) {
    // selector over {bytes 4..7 = '\n', bytes 0..3 = packed}
    const uint32_t sel = first == 1u ? 0x02010400u : (first == 2u ? 0x02040100u : (first == 3u ? 0x04020100u : 0x03020100u));
    *reinterpret_cast<u32_unaligned*>(at) = __builtin_amdgcn_perm(0x0A0A0A0Au, packed, sel);
    if (first < 4u) at[4] = (uint8_t)(packed >> 24);
}

// the same, cut by the start / end of the element (first and last step of a record).  o / col: line-break offset and
// column of residue jlow when jlow >= 0; a group that starts before residue 0 lies on the first line
__device__ __forceinline__ void put4_edge(uint8_t* body, int64_t jlow, uint32_t packed, uint32_t lw, uint32_t kept,
                                          uint32_t o, uint32_t col) {
    if (jlow + 3 < 0 || jlow >= (int64_t)kept) return;
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
        const int64_t jj = jlow + tq;
        if (jj >= 0 && (uint64_t)jj < kept) {
            uint32_t oo = 0;
            if (lw >= 4u) oo = jlow >= 0 ? o + (col + (uint32_t)tq >= lw ? 1u : 0u) : 0u;  // at most one break inside
            else if (lw) oo = (uint32_t)jj / lw;
            body[jj + oo] = (uint8_t)(packed >> (8 * tq));
        }
    }
}

#ifndef BSK_TR_WAVES
#define BSK_TR_WAVES 0
#endif
#ifndef BSK_TR_FAST
#define BSK_TR_FAST 1  // 0: IUPAC path only (measurement knob)
#endif
#if BSK_TR_WAVES
#define BSK_TR_ATTR __attribute__((amdgpu_waves_per_eu(BSK_TR_WAVES, 8)))
#else
#define BSK_TR_ATTR
#endif

template <int G>
__global__ __launch_bounds__(256) BSK_TR_ATTR void k_translate_frames4(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt,
                                                           TranslateParams P, const uint32_t* __restrict__ out_len,
                                                           const uint64_t* __restrict__ out_off,
                                                           uint8_t* __restrict__ out, uint64_t* __restrict__ status,
                                                           const uint8_t* __restrict__ only) {
    constexpr int STEPB = G * 12;      // bases per step
    constexpr int STEPS = 4;           // steps per window
    constexpr int WIN = STEPB * STEPS; // bases per window
    constexpr int RAWCAP = WIN + WIN / 16 + 64;  // source lines are >= 16 wide (narrower records are linearised)
    constexpr int NG = 256 / G;
    __shared__ __attribute__((aligned(16))) uint8_t s_tab[8192];  // codon table ++ reverse-complement-indexed table
    __shared__ uint8_t s_iu[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[NG][RAWCAP];
    const uint8_t* s_fw = s_tab;
    const uint8_t* s_rc = s_tab + 4096;
    if (only) {  // after k_translate_wide: a block none of whose records is flagged leaves before it stages anything
        const uint64_t g0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
        if (!__syncthreads_or(g0 < t.n && only[g0] != 0)) return;
    }
    for (int i = threadIdx.x * 16; i < 8192; i += blockDim.x * 16)
        *reinterpret_cast<uint4*>(s_tab + i) = *reinterpret_cast<const uint4*>(P.baked + i);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_iu[i] = P.iupac[i];
    __syncthreads();
    // the ACGT fast path: residue of the codon with 2-bit codes (first base in the low bits; A 0, C 1, T 2, G 3)
    __shared__ __attribute__((aligned(64))) uint8_t s_fw64[64];
    __shared__ __attribute__((aligned(64))) uint8_t s_rc64[64];
    if (threadIdx.x < 64) {
        const uint32_t iu[4] = {1u, 2u, 8u, 4u};  // IUPAC bit of code 0..3
        const uint32_t i = threadIdx.x;
        const uint32_t full = (iu[i & 3u] << 8) | (iu[(i >> 2) & 3u] << 4) | iu[(i >> 4) & 3u];
        s_fw64[i] = s_fw[full];
        s_rc64[i] = s_rc[full];
    }
    __syncthreads();
    const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const uint32_t gl = threadIdx.x % G;
    if (g >= t.n) return;  // no block-level barrier below
    if (only && !only[g]) return;  // k_translate_wide ran first: only the records it flagged are left
    if (P.long_thresh && t.l_seq[g] >= P.long_thresh) return;  // launch_translate_long handles this record
    uint8_t* raw = s_raw[threadIdx.x / G];
    const Text T = text_of(buf, t, tt, g);
    const uint32_t L = T.L;
    const uint32_t W = T.W;
    const uint32_t raw_total = W ? L + (L ? (L - 1) / W : 0u) : L;
    const uint8_t* h = buf + t.start[g] + 1;
    const uint32_t lh = t.l_head[g];
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    const uint32_t lw = P.line_width > 0 ? (uint32_t)P.line_width : 0u;

    constexpr uint64_t NONE = ~0ull;
    uint64_t fb[3] = {NONE, NONE, NONE};   // body offsets from `out` (keeps the global address space)
    uint64_t rbq[3] = {NONE, NONE, NONE};
    uint32_t fk[3] = {0, 0, 0}, rkq[3] = {0, 0, 0};
#pragma unroll 1
    for (int k = 0; k < P.nframes; ++k) {
        const int frame = P.frames[k];
        const uint64_t e = g * (uint64_t)P.nframes + (uint64_t)k;
        const uint32_t n = out_len[e];
        uint8_t* o = out + out_off[e];
        const uint32_t H = header_len(t, g, h, hl, P, frame) + 1;
        const uint32_t body = n - H - 1;
        const uint32_t kept = body - (lw ? body / (lw + 1) : 0u);
        if (!P.append_frame) {
            for (uint32_t x = gl; x < H; x += G) o[x] = x == 0 ? (uint8_t)'>' : (x == H - 1 ? (uint8_t)'\n' : h[x - 1]);
        } else if (gl == 0) {
            uint32_t hdr = 0, ioff, doff;
            o[hdr++] = '>';
            const uint32_t il = id_span_rec(t, g, h, hl, P.id_mode, &ioff);
            const uint32_t dl = desc_of(h, hl, P.id_mode, il, &doff);
            for (uint32_t q = 0; q < il; ++q) o[hdr++] = h[ioff + q];
            const char* fs = "_frame=";
            for (int q = 0; q < 7; ++q) o[hdr++] = (uint8_t)fs[q];
            hdr += put_dec(o + hdr, frame);
            o[hdr++] = ' ';
            for (uint32_t q = 0; q < dl; ++q) o[hdr++] = h[doff + q];
            o[hdr++] = '\n';
        }
        if (gl == 0) o[n - 1] = '\n';
        if (lw) for (uint32_t x = lw + gl * (lw + 1); x < body; x += G * (lw + 1)) o[H + x] = '\n';
        uint64_t bodyp = out_off[e] + H;
        if constexpr (G == 64)  // one record per wave: the six body addresses are wave-uniform -> scalar base + 32-bit offset
            bodyp = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bodyp >> 32)) << 32) |
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bodyp);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (frame == c + 1) { fb[c] = bodyp; fk[c] = kept; }
            if (frame == -(c + 1)) { rbq[c] = bodyp; rkq[c] = kept; }
        }
    }
    const uint32_t Lm = (L % 3u);
    uint64_t rb[3];
    uint32_t rk[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t sl = (Lm + 3u - (uint32_t)c) % 3u;  // reverse slot of class c
        rb[c] = sl == 0 ? rbq[0] : (sl == 1 ? rbq[1] : rbq[2]);
        rk[c] = sl == 0 ? rkq[0] : (sl == 1 ? rkq[1] : rkq[2]);
    }
    // unknown codon == a 0 byte among the existing residues.  Every real residue ('*', 'A'..'Z') has bit 5 or bit 6
    // set, so bit 6 of (pk | pk << 1) is 1 for a residue and 0 for the unknown marker: AND-accumulate that bit.
    uint32_t andacc = 0xFFFFFFFFu;
    const bool wide = lw == 0u || lw >= 4u;  // put4 assumes at most one line break among four residues
    const uint32_t g4 = 4u * G;
    const uint32_t gd = lw ? g4 / lw : 0u, gm = lw ? g4 % lw : 0u;
    // forward cursor (shared by the forward frames): first residue of the lane's group of four
    uint32_t fj = 4u * gl, fo = 0, fc = fj;
    if (lw) { fo = fj / lw; fc = fj - fo * lw; }
    // reverse cursors: LOWEST residue of the lane's group (the one of its 4th codon); descending by 4G per step
    int32_t rj[3];  // (a record holds < 2^32 bases, so < 2^31 residues)
    uint32_t ro[3], rcc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rj[c] = (L >= 3u + (uint32_t)c ? (int32_t)((L - 3u - (uint32_t)c) / 3u) : -1) - 4 * (int32_t)gl - 3;
        ro[c] = 0; rcc[c] = 0;
        if (lw && rj[c] >= 0) { ro[c] = (uint32_t)rj[c] / lw; rcc[c] = (uint32_t)rj[c] - ro[c] * lw; }
    }
    // raw cursor of the lane's first base (wrapped source): newlines before it and its column
    uint32_t rnl = 0, rcol = 0;
    if (W) { rnl = (12u * gl) / W; rcol = 12u * gl - rnl * W; }
    const uint32_t sdq = W ? (uint32_t)STEPB / W : 0u, smq = W ? (uint32_t)STEPB % W : 0u;

    for (uint32_t q0 = 0; q0 < L; q0 += WIN) {
        // ---- stage the window's raw text (bases + newlines) in LDS: coalesced 16-byte copies
        const uint32_t rawbase = W ? q0 + q0 / W : q0;
        uint32_t span = raw_total - rawbase;
        if (span > (uint32_t)RAWCAP) span = RAWCAP;
        for (uint32_t off = gl * 16u; off < span; off += G * 16u) {
            if (off + 16u <= span) {
                uint4 v;
                __builtin_memcpy(&v, T.p + rawbase + off, 16);
                *reinterpret_cast<uint4*>(raw + off) = v;
            } else {
                for (uint32_t bq = off; bq < span; ++bq) raw[bq] = T.p[rawbase + bq];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
        for (uint32_t it = 0; it < (uint32_t)STEPS; ++it) {
            const uint32_t sb = q0 + it * (uint32_t)STEPB;
            if (sb >= L) break;
            const uint32_t q = sb + 12u * gl;  // first base of the lane
            // ---- 16 raw bytes from the lane's first base
            const uint32_t rp = W ? (q + rnl) - rawbase : q - q0;
            uint32_t w[4];
            {
                const uint32_t* a = reinterpret_cast<const uint32_t*>(raw + (rp & ~3u));
                const uint32_t sh = rp & 3u;
                const uint32_t d0 = a[0], d1 = a[1], d2 = a[2], d3 = a[3], d4 = a[4];
                w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
                w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
                w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
                w[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
            }
            if (W) {
                // the line's newline sits k bytes after the first base; bytes above it move down by one
                const uint32_t k = W - rcol;
                if (k < 16u) {
                    const uint32_t s0 = __builtin_amdgcn_alignbyte(w[1], w[0], 1), s1 = __builtin_amdgcn_alignbyte(w[2], w[1], 1),
                                   s2 = __builtin_amdgcn_alignbyte(w[3], w[2], 1), s3 = w[3] >> 8;
                    const uint32_t sv[4] = {s0, s1, s2, s3};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int32_t km = (int32_t)k - 4 * i;  // bytes of dword i that stay
                        const uint32_t m = km >= 4 ? 0xFFFFFFFFu : (km <= 0 ? 0u : ((1u << (8 * km)) - 1u));
                        w[i] = (w[i] & m) | (sv[i] & ~m);
                    }
                }
            }
            const bool interior = sb + (uint32_t)STEPB + 2u <= L;  // every codon of every lane is complete
            // ---- residues of the lane's four codon slots, per class c (codons starting at base c) and strand.
            // FAST PATH (wave-uniform): all 14 letters of every lane are A, C, G, T (any case) and no codon is cut by the
            // record end.  A letter's bits 1..2 are its 2-bit code (A 0, C 1, T 2, G 3); v_perm rebuilds the letter from
            // the code for the check, v_dot4 packs four codes into a byte, the 16 codes of the lane form one dword X and
            // a codon's 6-bit index is ONE v_bfe -- into two 64-byte tables in LDS (forward, reverse complement) whose
            // 16 dwords sit in 16 different banks: no bank conflicts, against ~6-way conflicts of the 4 KiB IUPAC tables
            // (the LDS pipe, not the VALU, bounded the previous version: 38 byte reads per step, 47 of its 75 ms).
            uint32_t pkf[3] = {0, 0, 0}, pkr[3] = {0, 0, 0};
            uint32_t nvc[3] = {4, 4, 4};
            const uint32_t c0 = (w[0] >> 1) & 0x03030303u, c1 = (w[1] >> 1) & 0x03030303u, c2 = (w[2] >> 1) & 0x03030303u,
                           c3 = (w[3] >> 1) & 0x03030303u;
            constexpr uint32_t LET = 0x67746361u;  // 'a' 'c' 't' 'g' at byte 0..3 == letter of code 0..3
            const uint32_t bad = ((w[0] | 0x20202020u) ^ __builtin_amdgcn_perm(LET, LET, c0)) |
                                 ((w[1] | 0x20202020u) ^ __builtin_amdgcn_perm(LET, LET, c1)) |
                                 ((w[2] | 0x20202020u) ^ __builtin_amdgcn_perm(LET, LET, c2)) |
                                 (((w[3] | 0x20202020u) ^ __builtin_amdgcn_perm(LET, LET, c3)) & 0x0000FFFFu);
            if (BSK_TR_FAST && __ballot(bad != 0u || !interior) == 0ull) {
                const uint32_t X = __builtin_amdgcn_udot4(c0, 0x40100401u, 0u, false) |
                                   (__builtin_amdgcn_udot4(c1, 0x40100401u, 0u, false) << 8) |
                                   (__builtin_amdgcn_udot4(c2, 0x40100401u, 0u, false) << 16) |
                                   (__builtin_amdgcn_udot4(c3, 0x40100401u, 0u, false) << 24);  // base j at bits 2j, 2j+1
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (fb[c] == NONE && rb[c] == NONE) continue;
                    const uint32_t i0 = (X >> (2 * c)) & 63u, i1 = (X >> (2 * c + 6)) & 63u, i2 = (X >> (2 * c + 12)) & 63u,
                                   i3 = (X >> (2 * c + 18)) & 63u;
                    if (fb[c] != NONE)
                        pkf[c] = (uint32_t)s_fw64[i0] | ((uint32_t)s_fw64[i1] << 8) | ((uint32_t)s_fw64[i2] << 16) | ((uint32_t)s_fw64[i3] << 24);
                    if (rb[c] != NONE)  // descending residues: codon slot 3 is the lowest residue = byte 0
                        pkr[c] = (uint32_t)s_rc64[i3] | ((uint32_t)s_rc64[i2] << 8) | ((uint32_t)s_rc64[i1] << 16) | ((uint32_t)s_rc64[i0] << 24);
                }
            } else {
                uint32_t cd[14];
#pragma unroll
                for (int i = 0; i < 14; ++i) cd[i] = s_iu[byte_of(w, i)];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (fb[c] == NONE && rb[c] == NONE) continue;
                    uint32_t idx[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) idx[k] = (cd[3 * k + c] << 8) | (cd[3 * k + c + 1] << 4) | cd[3 * k + c + 2];
                    if (!interior) {  // complete codons of the lane in this class
                        nvc[c] = 0;
                        if (q + (uint32_t)c + 2u < L) { nvc[c] = (L - q - (uint32_t)c - 3u) / 3u + 1u; if (nvc[c] > 4u) nvc[c] = 4u; }
                    }
                    if (fb[c] != NONE)
                        pkf[c] = (uint32_t)s_fw[idx[0]] | ((uint32_t)s_fw[idx[1]] << 8) | ((uint32_t)s_fw[idx[2]] << 16) |
                                 ((uint32_t)s_fw[idx[3]] << 24);
                    if (rb[c] != NONE)
                        pkr[c] = (uint32_t)s_rc[idx[3]] | ((uint32_t)s_rc[idx[2]] << 8) | ((uint32_t)s_rc[idx[1]] << 16) |
                                 ((uint32_t)s_rc[idx[0]] << 24);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t nv = nvc[c];
                if (fb[c] != NONE) {
                    const uint32_t pk = pkf[c];
                    if (interior) andacc &= pk | (pk << 1);
                    else andacc &= pk | (pk << 1) | (nv >= 4u ? 0u : ~((1u << (8 * nv)) - 1u));
                    uint8_t* body = out + fb[c];
                    if (wide && sb / 3u + g4 <= fk[c]) put4(body + (uint32_t)(fj + fo), pk, lw ? lw - fc : 4u);
                    else put4_edge(body, (int64_t)fj, pk, lw, fk[c], fo, fc);
                }
                if (rb[c] != NONE) {
                    const uint32_t pk = pkr[c];
                    if (interior) andacc &= pk | (pk << 1);
                    else andacc &= pk | (pk << 1) | (nv >= 4u ? 0u : (nv == 0u ? 0xFFFFFFFFu : ((1u << (8 * (4u - nv))) - 1u)));
                    uint8_t* body = out + rb[c];
                    // lane 0 holds the highest residues of the step, lane G-1 the lowest
                    const int32_t hi0 = rj[c] + 4 * (int32_t)gl + 3, lo0 = hi0 - 4 * (int32_t)G + 1;
                    if (wide && lo0 >= 0 && (uint32_t)hi0 < rk[c]) put4(body + (uint32_t)((uint32_t)rj[c] + ro[c]), pk, lw ? lw - rcc[c] : 4u);
                    else put4_edge(body, rj[c], pk, lw, rk[c], ro[c], rcc[c]);
                }
            }
            // ---- advance the cursors by one step
            fj += g4;
            if (lw) { fo += gd; fc += gm; if (fc >= lw) { fc -= lw; ++fo; } }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rj[c] -= g4;
                if (lw && rj[c] >= 0) {
                    ro[c] -= gd;
                    if (rcc[c] < gm) { rcc[c] += lw - gm; --ro[c]; } else rcc[c] -= gm;
                }
            }
            if (W) { rnl += sdq; rcol += smq; if (rcol >= W) { rcol -= W; ++rnl; } }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (!P.allow_unknown && (andacc & 0x40404040u) != 0x40404040u)
        atomicOr((unsigned long long*)&status[0], (unsigned long long)ERR_UNKNOWN_CODON);
    // ---- -M: residue 0 of a frame becomes 'M' when its codon is a start codon (after every other store)
    if (P.init_m && gl == 0) {
        __threadfence();
        for (int k = 0; k < P.nframes; ++k) {
            const int frame = P.frames[k];
            const uint32_t f = (uint32_t)(frame < 0 ? -frame : frame);
            if (L < f + 2u) continue;
            const uint32_t p0 = frame > 0 ? f - 1u : L - f - 2u;  // first base of the codon (forward coordinates)
            const uint32_t ix = ((uint32_t)s_iu[T.at(p0)] << 8) | ((uint32_t)s_iu[T.at(p0 + 1)] << 4) | (uint32_t)s_iu[T.at(p0 + 2)];
            const bool st = frame > 0 ? P.start[ix] != 0 : P.start_rc[ix] != 0;
            const uint64_t e = g * (uint64_t)P.nframes + (uint64_t)k;
            const uint32_t H = header_len(t, g, h, hl, P, frame) + 1;
            if (st && out_len[e] > H + 1u) out[out_off[e] + H] = 'M';
        }
    }
}


// ---------------------------------------------------------------------------
// k_translate_wide<G>: the kernel of the common case -- plain A/C/G/T letters (any case), source lines of >= 50 bases
// (or unwrapped), output lines of >= 16 residues (or unwrapped).  A record whose text does not fit (an N, a narrow
// line, ...) is flagged in `redo` and translated by k_translate_frames4 afterwards; nothing here guesses.
//
// Why another kernel: k_translate_frames4 is bound by its WRITES, not by its instructions (measured: neither the ACGT
// fast path nor 16 vs 64 lanes per record nor 100 M fewer store instructions moved its 77 ms; PMC WRITE_SIZE 159 GB
// for 100.8 GB of output).  Its lanes own 4 residues per frame and step, so a store instruction covers 256 contiguous
// bytes per frame and every 128-byte line of the output is written in pieces by several instructions.
// Here a lane owns SIXTEEN codon slots (48 bases + 2 of look-ahead) per step:
//   * its 64 raw bytes come straight from global memory (four 16-byte loads; neighbouring lanes overlap by 2-4 bytes,
//     absorbed by L1/L2) -- no LDS staging, no barrier;
//   * letters -> 2-bit codes ((b >> 1) & 3: A 0, C 1, T 2, G 3) with one shift+and per dword, checked against the
//     letters by v_perm + v_sad_u8 (sum of |letter - letter(code)| over all bytes: 0, or exactly 57 when the lane's one
//     line break '\n' is among them), packed by v_dot4_u32_u8 into a 128-bit string X of codes; the break's two bits are
//     cut out of X with four v_alignbit / v_bfi -- the raw bytes are never spliced;
//   * a codon's 6-bit index is one v_bfe / v_alignbit on X, its residue one read from a 64-byte LDS table (16 banks, no
//     conflicts), forward and reverse-complement strand from the same index;
//   * 16 residues leave as ONE 16-byte store per frame (the line break of the output spliced in by v_perm with selectors
//     from a 6-entry LDS table, the 17th byte by a byte store): a wave instruction covers 1 KiB (G = 64) of contiguous
//     output per frame, so almost every line is written whole.
// ---------------------------------------------------------------------------
typedef uint4 __attribute__((aligned(1))) uint4_unaligned;
typedef uint32_t trw_u32x4 __attribute__((ext_vector_type(4)));

// codes of base slot `bit / 2` .. from the 128-bit code string (bit is a compile-time constant)
template <int BIT>
__device__ __forceinline__ uint32_t code6(const uint32_t (&X)[5]) {
    constexpr int w = BIT >> 5, sh = BIT & 31;
    if constexpr (sh <= 26) return (X[w] >> sh) & 63u;
    else return __builtin_amdgcn_alignbit(X[w + 1], X[w], sh) & 63u;
}

// twelve bits of the code string (two codon slots) times two: the byte offset of the pair in the tables below
template <int BIT>
__device__ __forceinline__ uint32_t code12x2(const uint32_t (&X)[5]) {
    constexpr int w = BIT >> 5, sh = BIT & 31;
    if constexpr (sh == 0) return (X[w] << 1) & 0x1FFEu;
    else if constexpr (sh <= 20) return (X[w] >> (sh - 1)) & 0x1FFEu;
    else return __builtin_amdgcn_alignbit(X[w + 1], X[w], sh - 1) & 0x1FFEu;
}

template <int C, int K>
struct Slots {  // residues of class C, slots K..15, forward ascending / reverse descending, packed into 4 dwords each
    // TWO slots per lookup: s_pair holds, for every pair of 2-bit codons (slot K in the low six bits, K + 1 above), the two
    // forward residues as they stand in the output (K first) at [0, 8192) and the two reverse-complement residues (K + 1
    // first: the reverse frame runs the other way) at [8192, 16384).  One residue per lookup cost six vector
    // instructions per dword (byte masks, shifts, an or3): 64 of the 250 per class and step.
    static __device__ __forceinline__ void run(const uint32_t (&X)[5], const uint8_t* s_pair, uint32_t (&pf)[4], uint32_t (&pr)[4]) {
        if constexpr (K < 16) {
            const uint32_t at = code12x2<2 * C + 6 * K>(X);
            const uint32_t f = *reinterpret_cast<const uint16_t*>(s_pair + at);
            const uint32_t r = *reinterpret_cast<const uint16_t*>(s_pair + 8192 + at);
            pf[K >> 2] |= f << (16 * ((K >> 1) & 1));
            pr[(14 - K) >> 2] |= r << (16 * (((14 - K) >> 1) & 1));
            Slots<C, K + 2>::run(X, s_pair, pf, pr);
        }
    }
};

// 16 residues (4 dwords, ascending addresses) at `at`, `first` = residues before the line break (>= 16: none inside);
// nbytes = how many of the 16 (17 with a break inside) bytes exist, counted from `at` (whole: 16 / 17)
// after16: a line break follows the block's 16th residue directly (first == 16) and exists in the body (another residue
// comes after it) -- the block writes it, so that no separate pass has to touch the lines of the output beforehand
__device__ __forceinline__ void put16(uint8_t* at, const uint32_t (&p)[4], uint32_t first, const uint2* s_ins, uint32_t skip,
                                      uint32_t nres, bool after16) {
    uint32_t d[4];
    if (first >= 16u) {
        d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int fm = (int)first - 4 * i;  // position of the break inside dword i
            fm = fm < -1 ? -1 : (fm > 4 ? 4 : fm);
            const uint2 e = s_ins[fm + 1];  // selector over {p[i] (bytes 4..7), p[i-1] (bytes 0..3)}, and the '\n' to OR in
            d[i] = __builtin_amdgcn_perm(p[i], i ? p[i - 1] : 0u, e.x) | e.y;
        }
    }
    if (skip == 0u && nres == 16u) {
        uint4 v = make_uint4(d[0], d[1], d[2], d[3]);
        *reinterpret_cast<uint4_unaligned*>(at) = v;
        if (first < 16u) at[16] = (uint8_t)(p[3] >> 24);
        else if (after16) at[16] = (uint8_t)'\n';
    } else {
        // a lane at the start / end of the element: residues [skip, skip + nres) of the sixteen exist, i.e. bytes [b0, b1) of
        // the 16 / 17.  Whole dwords leave as dwords, the rest as single bytes -- straight-line, predicated (a byte loop
        // here kept one lane busy for ~250 instructions per frame while 63 waited: 1 500 of the 2 000 per record)
        const uint32_t b0 = skip + (first < skip ? 1u : 0u);
        const uint32_t b1 = skip + nres + (first < skip + nres ? 1u : 0u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (b0 <= 4u * i && 4u * i + 4u <= b1) {
                *reinterpret_cast<u32_unaligned*>(at + 4 * i) = d[i];
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (b0 <= 4u * i + b && 4u * i + b < b1) at[4 * i + b] = (uint8_t)(d[i] >> (8 * b));
            }
        }
        if (b1 > 16u) at[16] = (uint8_t)(p[3] >> 24);
        else if (after16 && skip + nres == 16u) at[16] = (uint8_t)'\n';
    }
}

// ---- blocks written through a buffer resource (a wave per record) ----------------------------------------------------
// The body of an element is described by a raw buffer resource (base = its first byte, num_records = its bytes): gfx950
// checks every dword of a buffer_store_dwordx4 against the end of the resource on its own and drops the dwords that do
// not fit entirely (measured: scripts/experiments/probe_buffer_clip.hip -- a dword at offset s is written iff
// s + 4 <= num_records, a byte iff s < num_records, a NEGATIVE offset drops the whole store).  So the blocks at the end of
// an element (fewer residues than sixteen: the end of the sequence, --trim) leave as the same 16-byte store as every
// other block, and the at most three bytes of the dword across the end as byte stores that the hardware clips as well.
// The straight-line predicated byte stores this replaces (put16, skip / nres) were more than half of the kernel's 2 400
// vector instructions per record: every wave ran them in its last step for all six frames.
// selectors of the splice of a line break into the four dwords of a block (first = residues in front of the break)
__device__ __forceinline__ void splice_sel(uint32_t first, const uint2* s_ins, uint2 (&e)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int fm = (int)first - 4 * i;  // position of the break inside dword i
        fm = fm < -1 ? -1 : (fm > 4 ? 4 : fm);
        e[i] = s_ins[fm + 1];
    }
}

// at_end (wave-uniform): the body may end inside this block (the step that holds the end of the frame)
__device__ __forceinline__ void put16_buf(__amdgpu_buffer_rsrc_t rs, uint32_t nr, uint32_t o, const uint32_t (&p)[4], uint32_t first,
                                          const uint2 (&e)[4], bool at_end) {
    uint32_t d[4];
    if (first >= 16u) {
        d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_perm(p[i], i ? p[i - 1] : 0u, e[i].x) | e[i].y;
    }
    trw_u32x4 v;
    v.x = d[0]; v.y = d[1]; v.z = d[2]; v.w = d[3];
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)o, 0, 0);
    // the 17th byte: the block's last residue behind a break inside it, or the break that follows the block directly
    // (dropped by the hardware when the body ends with the block)
    if (first <= 16u) __builtin_amdgcn_raw_buffer_store_b8(first < 16u ? (uint8_t)(p[3] >> 24) : (uint8_t)'\n', rs, (int)(o + 16u), 0, 0);
    if (at_end) {
        // the dword across the end of the body
        const uint32_t rem = nr - o;  // bytes of the body from the block's first one (wraps when the block lies behind the body)
        if (rem < 16u && (rem & 3u)) {
            const uint32_t k = rem >> 2;
            const uint32_t w = k == 0u ? d[0] : (k == 1u ? d[1] : (k == 2u ? d[2] : d[3]));
            const uint32_t at = o + 4u * k;
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)w, rs, (int)at, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(w >> 8), rs, (int)(at + 1u), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(w >> 16), rs, (int)(at + 2u), 0, 0);
        }
    }
}

// the 52 raw bytes of a lane's window (50 bases and at most one line break; 13 dwords)
__device__ __forceinline__ void load_window(const uint8_t* a, const uint8_t* buf_end, uint32_t (&r)[13]) {
    if (a + 52 <= buf_end) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            uint4 v;
            __builtin_memcpy(&v, a + 16 * i, 16);
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
        __builtin_memcpy(&r[12], a + 48, 4);
    } else {  // the last bytes of the shard
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            uint32_t wv = 0;
            for (int b = 0; b < 4; ++b)
                if (a + 4 * i + b < buf_end) wv |= (uint32_t)a[4 * i + b] << (8 * b);
            r[i] = wv;
        }
    }
}

// the raw byte at which a lane's window expects its line break (offset kb; none expected: any byte of the window).  The
// window check (v_sad_u8 == 57) says that the window holds ONE byte that is no letter; it does not say where.  Lines of
// 60 / 50 / 60 / 60 bases have the line count and the length of 60 / 60 / 60 / 50 and a break in every window that expects
// one -- ten bytes early.  So the byte at the expected offset is fetched beside the window (the same cache
// line, one byte load per lane and step) and must be '\n'.
__device__ __forceinline__ uint32_t load_break(const uint8_t* a, uint32_t kb, const uint8_t* buf_end) {
    const uint8_t* p = a + (kb < 51u ? kb : 51u);
    return p < buf_end ? (uint32_t)*p : 0u;
}

#ifndef BSK_TRW_WAVES
#define BSK_TRW_WAVES 0
#endif
#if BSK_TRW_WAVES
#define BSK_TRW_ATTR __attribute__((amdgpu_waves_per_eu(BSK_TRW_WAVES, 8)))
#else
#define BSK_TRW_ATTR
#endif
// what the wide kernel needs to know of ONE record: where it is, how its text is laid out, and where its elements go
struct WideRec {
    uint64_t g;        // record number (table modes: row of the record table)
    Text T;            // sequence text: first base, bases, line width
    uint64_t rstart;   // the record's '>'
    uint32_t lh;       // header line length, marker included
    uint32_t ne[6];    // bytes of element k (frame k of the call)
    uint64_t oe[6];    // ... and its offset in the output
};

// The body of k_translate_wide for one record and one group of G lanes.  UNI: the record follows from its number
// (UniformLayout) and its frame is verified here; STREAM (round 5, k_translate_stream): the record comes from the pass that
// found it -- no record table exists (no redo list, no -M / -F: the host does not choose that pass for them).
template <int G, bool UNI, bool STREAM>
__device__ __forceinline__ void translate_wide_record(const uint8_t* __restrict__ buf, uint64_t buf_n, const RecordTable& t,
                                                      const TranslateParams& P, const uint32_t* __restrict__ out_len,
                                                      const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
                                                      uint8_t* __restrict__ redo, uint64_t* __restrict__ redo_count,
                                                      const uint8_t* s_pair, const uint8_t* s_iu, const uint2* s_ins, const WideRec& R) {
    constexpr uint32_t LB = 48;            // bases per lane and step (16 codon slots)
    constexpr uint32_t STEPB = G * LB;     // bases per group and step
    const uint64_t g = R.g;
    const uint32_t gl = threadIdx.x % G;
    const Text T = R.T;
    const uint64_t rstart = R.rstart;
    const uint32_t lh = R.lh;
    const uint32_t L = T.L;
    const uint32_t W = T.W;
    const uint32_t lw = P.line_width > 0 ? (uint32_t)P.line_width : 0u;
    if ((W && W < 50u) || (lw && lw < 16u)) {  // (wave-uniform per group) not this kernel's layout
        if (gl == 0) { if constexpr (!UNI && !STREAM) redo[g] = 1; atomicAdd((unsigned long long*)redo_count, 1ull); }
        return;
    }
    const uint8_t* h = buf + rstart + 1;
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    bool give_up = false;
    if constexpr (UNI) {
        // the frame of the record: '>' at its place behind a line break, a header line of exactly lh bytes, the last line
        // break at S - 1 (or the shard ends there: a file need not end with '\n').  The sequence lines are verified window
        // by window below (letters, and every break at its place).
        const uint8_t* rec = buf + rstart;
        const uint64_t endp = rstart + P.uni.S - 1u;  // the record's final line break
        bool bad0 = false;
        if (gl == 0) {
            bad0 = rec[0] != (uint8_t)'>' || (rstart > 0 && rec[-1] != (uint8_t)'\n') || rec[lh] != (uint8_t)'\n' ||
                   (endp < buf_n ? buf[endp] != (uint8_t)'\n' : endp != buf_n);
        }
        for (uint32_t x = 1u + gl; x < lh; x += G) bad0 = bad0 || rec[x] == (uint8_t)'\n';
        const uint64_t bb = __ballot(bad0);
        const uint32_t shift = ((threadIdx.x & 63u) / G) * G;
        const uint64_t gmask = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << shift);
        if (bb & gmask) {
            if (gl == 0) atomicAdd((unsigned long long*)redo_count, 1ull);
            return;
        }
    }

    // raw cursor of the lane's first base (wrapped source): line breaks before it and its column
    uint32_t rnl = 0, rcol = LB * gl;
    if (W) { rnl = rcol / W; rcol -= rnl * W; }
    const uint32_t sdq = W ? STEPB / W : 0u, smq = W ? STEPB % W : 0u;
    const uint8_t* const buf_end = buf + buf_n;
    // the window of the first step is requested before anything else is done with the record, the window of step s + 1
    // while step s is translated: the text's round trip to HBM hides behind the headers / the previous step
    uint32_t rn[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t rbk = 0;  // the byte at the window's expected line break (load_break)
    if (LB * gl < L) {
        load_window(T.p + LB * gl + rnl, buf_end, rn);
        rbk = load_break(T.p + LB * gl + rnl, W ? W - rcol : 0u, buf_end);
    }

    constexpr uint64_t NONE = ~0ull;
    uint64_t fb[3] = {NONE, NONE, NONE};   // body offsets from `out`
    uint64_t rbq[3] = {NONE, NONE, NONE};
    uint32_t fk[3] = {0, 0, 0}, rkq[3] = {0, 0, 0};
    uint32_t fnr[3] = {0, 0, 0}, rnq[3] = {0, 0, 0};  // bytes of the bodies (G == 64: the ends of the buffer resources)
    // everything the six elements need from memory is requested at once (a loop that fetched out_len / out_off / the
    // header bytes frame by frame put six dependent round trips in front of every record: the waves of this kernel live
    // for ~30 us, and most of that was waiting)
    uint32_t ne[6];
    uint64_t oe[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { ne[k] = R.ne[k]; oe[k] = R.oe[k]; }
    // header line and final line break of element k (n bytes at o, header of H bytes)
    const uint8_t hbyte = (gl >= 1u && gl - 1u < hl) ? h[gl - 1u] : (uint8_t)0;  // header byte of position gl (frames share it)
    auto put_header = [&](int frame, uint8_t* o, uint32_t n, uint32_t H) {
        if (!P.append_frame) {
            if (gl < H) o[gl] = gl == 0 ? (uint8_t)'>' : (gl == H - 1 ? (uint8_t)'\n' : hbyte);
            for (uint32_t x = gl + G; x < H; x += G) o[x] = x == H - 1 ? (uint8_t)'\n' : h[x - 1];
        } else if (gl == 0) {
            uint32_t hdr = 0, ioff, doff;
            o[hdr++] = '>';
            const uint32_t il = id_span_rec(t, g, h, hl, P.id_mode, &ioff);
            const uint32_t dl = desc_of(h, hl, P.id_mode, il, &doff);
            for (uint32_t q = 0; q < il; ++q) o[hdr++] = h[ioff + q];
            const char* fs = "_frame=";
            for (int q = 0; q < 7; ++q) o[hdr++] = (uint8_t)fs[q];
            hdr += put_dec(o + hdr, frame);
            o[hdr++] = ' ';
            for (uint32_t q = 0; q < dl; ++q) o[hdr++] = h[doff + q];
            o[hdr++] = '\n';
        }
        if (gl == 0) o[n - 1] = '\n';  // (the line breaks of the body are written with the residues around them)
    };
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (k >= P.nframes) break;
        const int frame = P.frames[k];
        const uint32_t n = ne[k];
        const uint32_t H = ((UNI || STREAM) ? 1u + hl : header_len(t, g, h, hl, P, frame)) + 1;
        const uint32_t body = n - H - 1;
        const uint32_t kept = body - (lw ? body / (lw + 1) : 0u);
        put_header(frame, out + oe[k], n, H);
        uint64_t bodyp = oe[k] + H;
        if constexpr (G == 64)  // one record per wave: the body addresses are wave-uniform -> scalar base + 32-bit offset
            bodyp = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bodyp >> 32)) << 32) |
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bodyp);
        uint32_t bodyn = body;
        if constexpr (G == 64) bodyn = (uint32_t)__builtin_amdgcn_readfirstlane((int)body);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (frame == c + 1) { fb[c] = bodyp; fk[c] = kept; fnr[c] = bodyn; }
            if (frame == -(c + 1)) { rbq[c] = bodyp; rkq[c] = kept; rnq[c] = bodyn; }
        }
    }
    // codons of class c (starting at base c + 3 s) belong to reverse frame -(sl + 1), sl = (L % 3 + 3 - c) % 3, as its
    // residue (L - 3 - c) / 3 - s
    const uint32_t Lm = L % 3u;
    uint64_t rb[3];
    uint32_t rk[3], rnr[3];
    int32_t r0[3];  // residue index of slot 0
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t sl = (Lm + 3u - (uint32_t)c) % 3u;
        rb[c] = sl == 0 ? rbq[0] : (sl == 1 ? rbq[1] : rbq[2]);
        rk[c] = sl == 0 ? rkq[0] : (sl == 1 ? rkq[1] : rkq[2]);
        rnr[c] = sl == 0 ? rnq[0] : (sl == 1 ? rnq[1] : rnq[2]);
        r0[c] = L >= 3u + (uint32_t)c ? (int32_t)((L - 3u - (uint32_t)c) / 3u) : -1;
    }
    const uint32_t g16 = 16u * G;  // residues per group and step
    const uint32_t gd = lw ? g16 / lw : 0u, gm = lw ? g16 % lw : 0u;
    // forward cursor: residue of the lane's slot 0, line breaks before it, its column
    uint32_t fj = 16u * gl, fo = 0, fc = fj;
    if (lw) { fo = fj / lw; fc = fj - fo * lw; }
    // reverse cursors: LOWEST residue of the lane's sixteen (slot 15); descending by 16 G per step
    int32_t rj[3];
    uint32_t ro[3], rcc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rj[c] = r0[c] - (int32_t)(16u * gl) - 15;
        ro[c] = 0; rcc[c] = 0;
        if (lw && rj[c] >= 0) { ro[c] = (uint32_t)rj[c] / lw; rcc[c] = (uint32_t)rj[c] - ro[c] * lw; }
    }
    constexpr uint32_t LET = 0x67746361u;  // 'a' 'c' 't' 'g' at byte 0..3 == letter of code 0..3

#pragma unroll 1
    for (uint32_t sb = 0; sb < L; sb += STEPB) {
        const uint32_t q = sb + LB * gl;  // first base of the lane
        const bool live = q < L;
        uint32_t X[5] = {0, 0, 0, 0, 0};
        bool bad = false;
        // cursor of the next step, and its window on the way
        uint32_t rnl2 = rnl, rcol2 = rcol;
        if (W) { rnl2 += sdq; rcol2 += smq; if (rcol2 >= W) { rcol2 -= W; ++rnl2; } }
        const uint32_t kbrk = W ? W - rcol : 0xFFFFu;            // raw bytes before this window's line break
        if (live) {
            const uint32_t nb = L - q < 50u ? L - q : 50u;     // bases of the lane's window that exist
            const bool brk = kbrk < nb;                          // (a break after the last base needed is not read)
            const uint32_t nraw = nb + (brk ? 1u : 0u);
            uint32_t r[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) r[i] = rn[i];
            // bytes past the window (the next letters, or whatever follows the record) must not take part in the check
            if (nraw >= 48u) {
                const uint32_t keep = nraw - 48u;  // 0..3 bytes of dword 12
                const uint32_t m = (1u << (8u * keep)) - 1u;
                r[12] = (r[12] & m) | (0x41414141u & ~m);
            } else {
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    const int keep = (int)nraw - 4 * i;
                    const uint32_t m = keep >= 4 ? 0xFFFFFFFFu : (keep <= 0 ? 0u : ((1u << (8 * keep)) - 1u));
                    r[i] = (r[i] & m) | (0x41414141u & ~m);
                }
            }
            uint32_t sad = 0;
            uint32_t d8[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                const uint32_t cd = (r[i] >> 1) & 0x03030303u;
                sad = __builtin_amdgcn_sad_u8(r[i] | 0x20202020u, __builtin_amdgcn_perm(LET, LET, cd), sad);
                d8[i] = __builtin_amdgcn_udot4(cd, 0x40100401u, 0u, false);
            }
            // |('\n' | 0x20) - 'c'| = 57: one byte, and nothing else, may differ -- and it must be a '\n' where the layout puts it
            bad = sad != (brk ? 57u : 0u) || (brk && rbk != 0x0Au);
            X[0] = d8[0] | (d8[1] << 8) | (d8[2] << 16) | (d8[3] << 24);
            X[1] = d8[4] | (d8[5] << 8) | (d8[6] << 16) | (d8[7] << 24);
            X[2] = d8[8] | (d8[9] << 8) | (d8[10] << 16) | (d8[11] << 24);
            X[3] = d8[12];
            if (q + STEPB < L) {  // (r is consumed: its registers are free)
                load_window(T.p + q + STEPB + rnl2, buf_end, rn);
                rbk = load_break(T.p + q + STEPB + rnl2, W ? W - rcol2 : 0u, buf_end);
            }
            if (brk) {  // cut the two bits of the break (raw byte kbrk) out of the string
                const uint32_t Y[5] = {X[0], X[1], X[2], X[3], 0u};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int b = 2 * (int)kbrk - 32 * i;
                    const uint32_t up = __builtin_amdgcn_alignbit(Y[i + 1], Y[i], 2);
                    const uint32_t m = b >= 32 ? 0xFFFFFFFFu : (b <= 0 ? 0u : ((1u << b) - 1u));
                    X[i] = (Y[i] & m) | (up & ~m);
                }
            }
        }
        // a group that met a letter this kernel does not know leaves the record to k_translate_frames4
        {
            const uint64_t bb = __ballot(bad);
            const uint32_t shift = ((threadIdx.x & 63u) / G) * G;
            const uint64_t gmask = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << shift);
            if (bb & gmask) { give_up = true; break; }
        }
        if (live) {
            // the forward frames share their cursor: one set of splice selectors for the three of them
            const uint32_t ffirst = lw ? lw - fc : 17u;
            const bool more_steps = sb + STEPB < L;  // (wave-uniform)
            uint2 fsel[4];
            if constexpr (G == 64) splice_sel(ffirst, s_ins, fsel);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bool wf = fb[c] != NONE, wr = rb[c] != NONE;
                if (!wf && !wr) continue;
                uint32_t pf[4] = {0, 0, 0, 0}, pr[4] = {0, 0, 0, 0};
                if (c == 0) Slots<0, 0>::run(X, s_pair, pf, pr);
                else if (c == 1) Slots<1, 0>::run(X, s_pair, pf, pr);
                else Slots<2, 0>::run(X, s_pair, pf, pr);
                // complete codons among the lane's sixteen slots of this class
                auto valid_slots = [&]() {
                    uint32_t nv = 0;
                    if (q + (uint32_t)c + 2u < L) { nv = (L - q - (uint32_t)c - 3u) / 3u + 1u; if (nv > 16u) nv = 16u; }
                    return nv;
                };
                if constexpr (G == 64) {
                    // a wave per record: the blocks leave through buffer resources that end with the bodies (put16_buf)
                    // (the resource words as scalars by decree: they come through selects on L % 3, a vector register as far as
                    // the compiler knows, and a resource in vector registers costs a waterfall loop around every store)
                    auto body_rsrc = [&](uint64_t off, uint32_t nr) {
                        const uint64_t a = (uint64_t)(uintptr_t)out + off;
                        const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
                        return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)u, 0, __builtin_amdgcn_readfirstlane((int)nr), 0x00020000);
                    };
                    if (wf) put16_buf(body_rsrc(fb[c], fnr[c]), fnr[c], fj + fo, pf, ffirst, fsel, !more_steps);
                    if (wr && rj[c] >= 0) {
                        const uint32_t rfirst = lw ? lw - rcc[c] : 17u;
                        uint2 rsel[4];
                        splice_sel(rfirst, s_ins, rsel);
                        put16_buf(body_rsrc(rb[c], rnr[c]), rnr[c], (uint32_t)rj[c] + ro[c], pr, rfirst, rsel, sb == 0u);
                    } else if (wr && rj[c] > -16) {
                        // the block across residue 0 (one lane per frame and record; a negative offset would drop the whole
                        // store): its residues 0 .. hi - 1 (< 16 <= lw) are on the first line
                        int32_t hi = rj[c] + 16;
                        if (hi > (int32_t)rk[c]) hi = (int32_t)rk[c];
                        int32_t lo = rj[c] + 16 - (int32_t)valid_slots();
                        if (lo < 0) lo = 0;
                        uint8_t* base = out + rb[c];
                        for (int32_t j = lo; j < hi; ++j) {
                            const uint32_t kk = (uint32_t)(j - rj[c]);
                            base[j] = (uint8_t)(pr[kk >> 2] >> (8 * (kk & 3u)));
                        }
                    }
                    continue;
                }
                const uint32_t nv = valid_slots();
                if (wf) {
                    // residues fj .. fj + nv - 1, cut by `kept` (--trim)
                    uint32_t nres = nv;
                    if (fj >= fk[c]) nres = 0; else if (fj + nres > fk[c]) nres = fk[c] - fj;
                    if (nres) {
                        const uint32_t first = lw ? lw - fc : 17u;
                        put16(out + fb[c] + (uint32_t)(fj + fo), pf, first, s_ins, 0u, nres, first == 16u && fj + 16u < fk[c]);
                    }
                }
                if (wr) {
                    // slot k is residue rj + 15 - k: the valid slots 0 .. nv-1 are the HIGH residues of the block
                    // [rj, rj + 15]; below 0 nothing exists, at `kept` and above nothing is printed (--trim)
                    int32_t lo = rj[c] + 16 - (int32_t)nv, hi = rj[c] + 16;  // residues [lo, hi)
                    if (lo < 0) lo = 0;
                    if (hi > (int32_t)rk[c]) hi = (int32_t)rk[c];
                    if (hi > lo) {
                        const uint32_t skip = (uint32_t)(lo - rj[c]);
                        // block position: residue rj may be negative (then skip > 0): positions are relative to residue rj
                        if (rj[c] >= 0) {
                            const uint32_t first = lw ? lw - rcc[c] : 17u;
                            put16(out + rb[c] + (uint32_t)((uint32_t)rj[c] + ro[c]), pr, first, s_ins, skip, (uint32_t)(hi - lo),
                                  first == 16u && (uint32_t)rj[c] + 16u < rk[c]);
                        } else {
                            // the block starts before residue 0: its residues 0 .. hi-1 (< 16 <= lw) are on the first line
                            uint8_t* base = out + rb[c];
                            for (int32_t j = lo; j < hi; ++j) {
                                const uint32_t kk = (uint32_t)(j - rj[c]);
                                base[j] = (uint8_t)(pr[kk >> 2] >> (8 * (kk & 3u)));
                            }
                            if (lw == (uint32_t)hi && (uint32_t)hi < rk[c]) base[hi] = (uint8_t)'\n';  // (hi == 16 == lw)
                        }
                    }
                }
            }
        }
        // ---- advance the cursors by one step
        fj += g16;
        if (lw) { fo += gd; fc += gm; if (fc >= lw) { fc -= lw; ++fo; } }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rj[c] -= (int32_t)g16;
            if (lw && rj[c] >= 0) {
                ro[c] -= gd;
                if (rcc[c] < gm) { rcc[c] += lw - gm; --ro[c]; } else rcc[c] -= gm;
            }
        }
        rnl = rnl2; rcol = rcol2;
    }
    if (give_up) {
        if (gl == 0) { if constexpr (!UNI && !STREAM) redo[g] = 1; atomicAdd((unsigned long long*)redo_count, 1ull); }
        return;
    }
    // ---- -M: residue 0 of a frame becomes 'M' when its codon is a start codon (after every other store)
    if constexpr (!UNI && !STREAM) if (P.init_m && gl == 0) {
        __threadfence();
        for (int k = 0; k < P.nframes; ++k) {
            const int frame = P.frames[k];
            const uint32_t f = (uint32_t)(frame < 0 ? -frame : frame);
            if (L < f + 2u) continue;
            const uint32_t p0 = frame > 0 ? f - 1u : L - f - 2u;  // first base of the codon (forward coordinates)
            const uint32_t ix = ((uint32_t)s_iu[T.at(p0)] << 8) | ((uint32_t)s_iu[T.at(p0 + 1)] << 4) | (uint32_t)s_iu[T.at(p0 + 2)];
            const bool st = frame > 0 ? P.start[ix] != 0 : P.start_rc[ix] != 0;
            const uint64_t e = g * (uint64_t)P.nframes + (uint64_t)k;
            const uint32_t H = header_len(t, g, h, hl, P, frame) + 1;
            if (st && out_len[e] > H + 1u) out[out_off[e] + H] = 'M';
        }
    }
    (void)out_len; (void)out_off;
}

template <int G, bool UNI>
__global__ __launch_bounds__(256) BSK_TRW_ATTR void k_translate_wide(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t, TextTable tt,
                                                        TranslateParams P, const uint32_t* __restrict__ out_len,
                                                        const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
                                                        uint8_t* __restrict__ redo, uint64_t* __restrict__ redo_count,
                                                        uint64_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t s_pair[16384];  // residues of two 2-bit codons at once (Slots), from P.pair
    __shared__ uint8_t s_iu[256];
    __shared__ uint2 s_ins[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // 16 KiB by 256 threads
        const uint4 v = *reinterpret_cast<const uint4*>(P.pair + 16 * (threadIdx.x + 256 * i));
        *reinterpret_cast<uint4*>(s_pair + 16 * (threadIdx.x + 256 * i)) = v;
    }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_iu[i] = P.iupac[i];
    if (threadIdx.x == 0) {
        // break inside dword i at byte fm (index fm + 1): bytes below fm from p[i], the break, bytes above from one lower
        s_ins[0] = make_uint2(0x06050403u, 0u);            // fm <= -1: the break came earlier, everything one byte up
        s_ins[1] = make_uint2(0x0605040Cu, 0x0000000Au);   // fm == 0
        s_ins[2] = make_uint2(0x06050C04u, 0x00000A00u);
        s_ins[3] = make_uint2(0x060C0504u, 0x000A0000u);
        s_ins[4] = make_uint2(0x0C060504u, 0x0A000000u);
        s_ins[5] = make_uint2(0x07060504u, 0u);            // fm >= 4: the break comes later
    }
    __syncthreads();
    const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (g >= (UNI ? P.uni.n : t.n)) return;  // no block-level barrier below
    WideRec R;
    R.g = g;
    if constexpr (UNI) {  // everything follows from the record number (UniformLayout) -- and is verified in the body
        R.rstart = g * P.uni.S;
        R.lh = P.uni.H;
        R.T.p = buf + R.rstart + R.lh + 1;
        R.T.L = P.uni.L;
        R.T.W = P.uni.W;
    } else {
        if (P.long_thresh && t.l_seq[g] >= P.long_thresh) return;  // launch_translate_long handles this record
        R.T = text_of(buf, t, tt, g);
        R.rstart = t.start[g];
        R.lh = t.l_head[g];
    }
    // everything the six elements need from memory is requested at once (a loop that fetched out_len / out_off / the
    // header bytes frame by frame put six dependent round trips in front of every record: the waves of this kernel live
    // for ~30 us, and most of that was waiting)
    const uint64_t e0 = g * (uint64_t)P.nframes;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if constexpr (UNI) {
            R.ne[k] = k < P.nframes ? P.uni.len[k] : 0u;
            R.oe[k] = k < P.nframes ? g * P.uni.out_S + P.uni.off[k] : 0ull;
        } else {
            R.ne[k] = k < P.nframes ? out_len[e0 + k] : 0u;
            R.oe[k] = k < P.nframes ? out_off[e0 + k] : 0ull;
        }
    }
    translate_wide_record<G, UNI, false>(buf, buf_n, t, P, out_len, out_off, out, redo, redo_count, s_pair, s_iu, s_ins, R);
    (void)status;
}


// ---- round 5: `translate` on FASTA records that do NOT all look alike, in ONE pass over the input --------------------------
// The table path reads the shard twice: k_fasta_starts looks for the '>' bytes (8.5 ms per 50 GB), then the wide kernel reads
// the text again -- with k_fasta_heads and the size / scan launches in between, 57 ms at C4 against 43 ms for records that
// all look alike.  Here persistent blocks take small line-start ranges of the input (64 KiB: what a block reads in phase A
// is still in the caches when phase D reads it again) from a queue:
//   A  the '>' at line starts of the range                                   (16 bytes per thread and step, SWAR compare)
//   B  per record: header end and first line from its first bytes, the next record's start (the range's last record:
//      a forward scan of the block), bases as "every line but the last is as long as the first" -- the rule of the
//      light table, verified window by window in phase D -- and the bytes of its elements
//   C  where the block's output begins: the blocks before it publish their totals in a chain of (flag, value) words as
//      soon as phase B is done; a wave looks back 64 predecessors at a time (as k_rmdup_place)
//   D  a wave per record: translate_wide_record<64, false, STREAM>
// Anything that does not fit -- more records in a range than the list holds, a first line under 16 bases that is not the
// only one, a record of a MiB, a window that fails its check, an output beyond the reserved capacity -- counts into
// redo_count, and the host runs the call again on the table paths (and the context stays with them).
constexpr uint32_t TS_MAXR = 512;   // records per range the lists hold (512 KiB ranges of records from 1.5 kB: the host's gate)
// 16-bit mask of the bytes of v that equal the replicated byte `rep` (bit b = byte b)
__device__ __forceinline__ uint32_t ts_eq_mask16(const uint4& v, uint32_t rep) {
    auto z = [](uint32_t x) { const uint32_t t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); return ((t >> 7) | (t >> 14) | (t >> 21) | (t >> 28)) & 0xFu; };
    return z(v.x ^ rep) | (z(v.y ^ rep) << 4) | (z(v.z ^ rep) << 8) | (z(v.w ^ rep) << 12);
}
constexpr uint32_t TS_PER = TS_MAXR / 256u;
constexpr uint64_t TS_FLAG_AGG = 1ull << 62, TS_FLAG_PREFIX = 2ull << 62, TS_VALUE = (1ull << 62) - 1ull;
__device__ __forceinline__ uint64_t ts_load(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ts_store(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ts_wave_sum(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
// bytes of the elements of a record with header line lh (marker included) and L bases
__device__ __forceinline__ uint32_t ts_element_bytes(const TranslateParams& P, uint32_t lh, uint32_t L, int k) {
    const uint32_t naa = num_aa(L, P.frames[k]);
    return lh + 1u + naa + ((P.line_width > 0 && naa > 0) ? (naa - 1u) / (uint32_t)P.line_width : 0u) + 1u;
}

#ifndef BSK_TRS_INFLIGHT
#define BSK_TRS_INFLIGHT 4  // 16-byte loads per thread in flight while a range is searched for its '>' (scripts/history/r05_trs_inflight.sh, one visit: 4 -> 49.8-50.1 ms, 8 -> 50.3-50.4, 16 -> 50.7: the search shares the memory system with the blocks that translate)
#endif
#ifndef BSK_TRS_WAVES
#define BSK_TRS_WAVES 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BSK_TRS_WAVES, 8))) void k_translate_stream(const uint8_t* __restrict__ buf, uint64_t buf_n,
                                                          const uint64_t* __restrict__ anchors, uint32_t nranges,
                                                          uint32_t* __restrict__ queue, TranslateParams P, uint8_t* __restrict__ out,
                                                          uint64_t out_cap, uint64_t* __restrict__ chain, uint64_t* __restrict__ fin,
                                                          uint64_t* __restrict__ redo_count, uint64_t* __restrict__ status, uint32_t span_hint) {
    __shared__ __attribute__((aligned(16))) uint8_t s_pair[16384];
    __shared__ uint32_t s_probe_bad;
    __shared__ uint64_t s_probe_first, s_probe_span;
    __shared__ uint8_t s_iu[256];
    __shared__ uint2 s_ins[6];
    __shared__ uint64_t s_pos[TS_MAXR], s_start[TS_MAXR + 1], s_off[TS_MAXR];
    __shared__ uint32_t s_lh[TS_MAXR], s_L[TS_MAXR], s_W[TS_MAXR], s_sz[TS_MAXR];
    __shared__ uint32_t s_n, s_r;
    __shared__ unsigned long long s_next;
    __shared__ uint64_t s_w[4], s_excl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 v = *reinterpret_cast<const uint4*>(P.pair + 16 * (threadIdx.x + 256 * i));
        *reinterpret_cast<uint4*>(s_pair + 16 * (threadIdx.x + 256 * i)) = v;
    }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_iu[i] = P.iupac[i];
    if (threadIdx.x == 0) {
        s_ins[0] = make_uint2(0x06050403u, 0u);
        s_ins[1] = make_uint2(0x0605040Cu, 0x0000000Au);
        s_ins[2] = make_uint2(0x06050C04u, 0x00000A00u);
        s_ins[3] = make_uint2(0x060C0504u, 0x000A0000u);
        s_ins[4] = make_uint2(0x0C060504u, 0x0A000000u);
        s_ins[5] = make_uint2(0x07060504u, 0u);
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t n_eff = anchors[nranges];
    const uint8_t* lim = buf + n_eff;
    RecordTable t_none;  // (no table on this pass; the record body does not touch it: no -F, no -M)
    memset(&t_none, 0, sizeof t_none);
    for (;;) {
        __syncthreads();  // (the lists of the range before are done with)
        if (threadIdx.x == 0) { s_r = atomicAdd(queue, 1u); s_n = 0; s_next = ~0ull; }
        __syncthreads();
        const uint32_t r = s_r;
        if (r >= nranges) break;
        uint64_t rs = anchors[r], re = anchors[r + 1];
        rs = rs < n_eff ? rs : n_eff;
        re = re < n_eff ? re : n_eff;
        if (r == 0 && threadIdx.x == 0 && n_eff && buf[0] != (uint8_t)'>') atomicOr((unsigned long long*)&status[0], (unsigned long long)ERR_BAD_HEADER);
        // ---- A: record starts of the range (four loads in flight per thread: one at a time, the 128 steps of a 512 KiB range
        // cost a memory round trip each -- 15 % of the pass; four in flight: see BSK_TRS_INFLIGHT)
        {
            auto take = [&](uint64_t at, const uint4& v) {
                // (sequence text is letters: a piece without a byte below 0x3F holds no '>' -- 3 in 4 pieces stop here)
                const uint32_t h = (((v.x - 0x3F3F3F3Fu) & ~v.x) | ((v.y - 0x3F3F3F3Fu) & ~v.y) | ((v.z - 0x3F3F3F3Fu) & ~v.z) | ((v.w - 0x3F3F3F3Fu) & ~v.w)) & 0x80808080u;
                if (h == 0u) return;
                uint32_t m = ts_eq_mask16(v, 0x3E3E3E3Eu);
                if (re - at < 16) m &= (1u << (re - at)) - 1u;
                while (m) {
                    const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
                    m &= m - 1u;
                    const uint64_t p = at + b;
                    if (p == 0 || buf[p - 1] == (uint8_t)'\n') {
                        const uint32_t k = atomicAdd(&s_n, 1u);
                        if (k < TS_MAXR) s_pos[k] = p;
                    }
                }
            };
            auto load = [&](uint64_t at) -> uint4 {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (at + 16 <= buf_n) __builtin_memcpy(&v, buf + at, 16);
                else {
                    uint32_t wv[4] = {0, 0, 0, 0};
                    for (uint32_t b = 0; at + b < buf_n; ++b) wv[b >> 2] |= (uint32_t)buf[at + b] << ((b & 3u) * 8u);
                    v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                }
                return v;
            };
            constexpr uint64_t STRIDE = 16ull * 256ull;
            // the search of [from, re): every 16-byte piece of it, BSK_TRS_INFLIGHT loads per thread in flight
            auto scan = [&](uint64_t from, uint64_t to) {
                uint64_t at = from + 16ull * threadIdx.x;
                const uint64_t re_saved = re;
                re = to;   // (`take` clips at re)
                for (; at + (BSK_TRS_INFLIGHT - 1) * STRIDE < to; at += BSK_TRS_INFLIGHT * STRIDE) {
                    uint4 v[BSK_TRS_INFLIGHT];
#pragma unroll
                    for (int k = 0; k < BSK_TRS_INFLIGHT; ++k) v[k] = load(at + (uint64_t)k * STRIDE);
#pragma unroll
                    for (int k = 0; k < BSK_TRS_INFLIGHT; ++k) take(at + (uint64_t)k * STRIDE, v[k]);
                }
                for (; at < to; at += STRIDE) take(at, load(at));
                re = re_saved;
            };
            // Round 6: PREDICT the record starts, then verify them (searching the whole range and then
            // translating it read the input 2.07 times -- the megabyte a block searched has left every cache when its records are
            // translated a millisecond later).  Records of one file tend to be of one size (reads, amplicons, CDS sets): the
            // head of the range is searched as before, the starts found there give a stride, and every further start of the
            // range is looked for in a 128-byte window around its predicted place only -- one '>' at a line start in every
            // window, or the range is searched in full after all (as is every range of a file whose records vary: the attempt
            // costs ~7 % of the range's bytes).  A start that is MISSED this way -- a short record between two windows -- makes
            // its neighbour's "sequence" hold a header line, which the window check of the translation refuses (redo_count: the
            // call takes the table path).  translate_probe = off keeps the full search.
            bool probed = false;
            if (span_hint) {
                const uint64_t head = rs + (((uint64_t)span_hint * 3u + 4095u) & ~4095ull);
                if (head + 2ull * span_hint < re) {
                    scan(rs, head);
                    if (threadIdx.x == 0) s_probe_bad = 0;
                    __syncthreads();
                    const uint32_t n0 = s_n;
                    if (threadIdx.x == 0) {
                        uint64_t lo = ~0ull, hi = 0;
                        for (uint32_t i = 0; i < n0 && i < TS_MAXR; ++i) { lo = s_pos[i] < lo ? s_pos[i] : lo; hi = s_pos[i] > hi ? s_pos[i] : hi; }
                        s_probe_first = hi;                                                 // the last start of the head
                        s_probe_span = n0 >= 2u && n0 <= 64u ? (hi - lo) / (n0 - 1u) : 0ull;  // mean distance of the head's starts
                    }
                    __syncthreads();
                    const uint64_t span = s_probe_span, last = s_probe_first;
                    if (span >= 256u) {
                        const uint64_t K = last + span < re ? (re - 1 - last) / span : 0;  // predicted starts last + k span < re, k = 1 .. K
                        if (n0 + K + 2 <= TS_MAXR) {
                            for (uint64_t k = 1 + threadIdx.x; k <= K + 1; k += 256u) {
                                // (k = K + 1: a start just below re that the drift moved there would else be lost)
                                const uint64_t pk = last + k * span;
                                const uint64_t w0 = (pk - 56) & ~15ull;   // [w0, w0 + 128) holds pk - 56 .. pk + 56
                                uint32_t found = 0;
#pragma unroll
                                for (int half = 0; half < 2; ++half) {  // (four loads in flight, twice: the registers of the search)
                                    uint4 v[4];
#pragma unroll
                                    for (int q = 0; q < 4; ++q) v[q] = load(w0 + 64ull * half + 16ull * q);
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        const uint64_t at = w0 + 64ull * half + 16ull * q;
                                        uint32_t m = ts_eq_mask16(v[q], 0x3E3E3E3Eu);
                                        while (m) {
                                            const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
                                            m &= m - 1u;
                                            const uint64_t pp = at + b;
                                            if (pp < n_eff && buf[pp - 1] == (uint8_t)'\n') {
                                                ++found;
                                                if (pp < re) {
                                                    const uint32_t slot = atomicAdd(&s_n, 1u);
                                                    if (slot < TS_MAXR) s_pos[slot] = pp;
                                                }
                                            }
                                        }
                                    }
                                }
                                // exactly one start per window; only the window behind the range's last start may lie behind the file
                                if (found != 1u && !(found == 0u && k == K + 1 && w0 + 128 >= n_eff)) atomicAdd(&s_probe_bad, 1u);
                            }
                            __syncthreads();
                            probed = s_probe_bad == 0u && s_n <= TS_MAXR;
                        }
                    }
                    if (!probed) {  // the starts of the head stay; the rest of the range in full
                        __syncthreads();
                        if (threadIdx.x == 0) s_n = n0;
                        __syncthreads();
                        scan(head, re);
                        probed = true;
                    }
                }
            }
            if (!probed) scan(rs, re);
        }
        __syncthreads();
        uint32_t nrec = s_n;
        bool unfit = nrec > TS_MAXR;
        if (unfit) nrec = 0;
        // in file order (a rank sort: the list is a few dozen entries)
        for (uint32_t i = threadIdx.x; i < nrec; i += 256u) {
            const uint64_t p = s_pos[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nrec; ++j) rank += s_pos[j] < p;
            s_start[rank] = p;
        }
        __syncthreads();
        // ---- B: the start of the record behind the range's last one (it lies in a later range): the first '>' at a line
        // start at or behind re
        if (nrec) {
            for (uint64_t base = re; base < n_eff; base += 16ull * 256ull) {
                const uint64_t at = base + 16ull * threadIdx.x;
                if (at < n_eff) {
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (at + 16 <= buf_n) __builtin_memcpy(&v, buf + at, 16);
                    else {
                        uint32_t wv[4] = {0, 0, 0, 0};
                        for (uint32_t b = 0; at + b < buf_n; ++b) wv[b >> 2] |= (uint32_t)buf[at + b] << ((b & 3u) * 8u);
                        v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                    }
                    uint32_t m = ts_eq_mask16(v, 0x3E3E3E3Eu);
                    if (n_eff - at < 16) m &= (1u << (n_eff - at)) - 1u;
                    while (m) {
                        const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
                        m &= m - 1u;
                        const uint64_t p = at + b;
                        if (buf[p - 1] == (uint8_t)'\n') { atomicMin(&s_next, (unsigned long long)p); break; }
                    }
                }
                __syncthreads();
                const bool found = s_next != ~0ull;
                __syncthreads();
                if (found) break;
            }
            if (threadIdx.x == 0) s_start[nrec] = s_next != ~0ull ? (uint64_t)s_next : n_eff;
        }
        __syncthreads();
        // header end, first line, bases (the rule of k_fasta_heads) and the bytes of the elements of every record
        uint64_t mine = 0;
        // (a thread takes TS_PER NEIGHBOURING records: the block scan below then is a scan over the records in file order)
        for (uint32_t i = threadIdx.x * TS_PER; i < nrec && i < (threadIdx.x + 1u) * TS_PER; ++i) {
            const uint64_t s0 = s_start[i], e0 = s_start[i + 1];
            const uint64_t span = e0 - s0;
            uint32_t lh = span > 0xFFFFFFFFull ? 0xFFFFFFFFu : find_byte_in(buf + s0, (uint32_t)span, '\n', lim);  // header line without its newline (== span: none)
            const uint64_t region = span > (uint64_t)lh + 1 ? span - lh - 1 : 0;
            uint32_t w = 0, lseq = 0, tw = 0;
            bool bad = span > 0xFFFFFFFFull;
            if (region && !bad) {
                w = find_byte_in(buf + s0 + lh + 1, (uint32_t)region, '\n', lim);
                const uint64_t R = (e0 == n_eff && buf[n_eff - 1] != (uint8_t)'\n') ? region + 1 : region;
                if (R <= (uint64_t)w + 1) { lseq = w; tw = 0; }
                else if (w < 16u) bad = true;
                else {
                    const uint64_t qn = R / ((uint64_t)w + 1), rem = R % ((uint64_t)w + 1);
                    lseq = (uint32_t)(qn * w + (rem ? rem - 1 : 0));
                    tw = w;
                }
            }
            if (P.long_thresh && lseq >= P.long_thresh) bad = true;  // a chromosome: whole blocks translate it (table path)
            s_lh[i] = lh;
            s_L[i] = lseq;
            s_W[i] = tw;
            uint32_t bytes = 0;
            if (bad) atomicAdd((unsigned long long*)redo_count, 1ull);
            else
                for (int k = 0; k < P.nframes; ++k) bytes += ts_element_bytes(P, lh, lseq, k);
            s_sz[i] = bytes;
            mine += bytes;
        }
        if (unfit && threadIdx.x == 0) atomicAdd((unsigned long long*)redo_count, 1ull);
        // ---- C: output offsets -- block scan, then the chain over the ranges
        uint64_t x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
            if ((int)lane >= d) x += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        if (wave == 0) {
            const uint64_t total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            if (lane == 0) ts_store(chain + r, (r == 0 ? TS_FLAG_PREFIX : TS_FLAG_AGG) | total);
            uint64_t excl = 0;
            if (r > 0) {
                int64_t j = (int64_t)r - 1 - (int64_t)lane;
                for (;;) {
                    uint64_t v = j >= 0 ? ts_load(chain + j) : TS_FLAG_PREFIX;
                    while (__ballot((v >> 62) == 0ull) != 0ull) {
                        __builtin_amdgcn_s_sleep(1);
                        if ((v >> 62) == 0ull) v = ts_load(chain + j);
                    }
                    const uint64_t pm = __ballot((v >> 62) == 2ull);
                    if (pm) {
                        const uint32_t fl = (uint32_t)__ffsll((long long)pm) - 1u;
                        excl += ts_wave_sum(lane <= fl ? (v & TS_VALUE) : 0ull);
                        break;
                    }
                    excl += ts_wave_sum(v & TS_VALUE);
                    j -= 64;
                }
                if (lane == 0) ts_store(chain + r, TS_FLAG_PREFIX | (excl + total));
            }
            if (lane == 0) {
                s_excl = excl;
                if (nrec) atomicAdd((unsigned long long*)&fin[1], (unsigned long long)nrec);
                if (r + 1 == nranges) fin[0] = excl + total;  // the whole output
            }
        }
        __syncthreads();
        {
            uint64_t off = s_excl + x - mine;
            for (uint32_t w2 = 0; w2 < wave; ++w2) off += s_w[w2];
            for (uint32_t i = threadIdx.x * TS_PER; i < nrec && i < (threadIdx.x + 1u) * TS_PER; ++i) {
                s_off[i] = off;
                off += s_sz[i];
                if (off > out_cap) { s_L[i] = 0xFFFFFFFFu; atomicAdd((unsigned long long*)redo_count, 1ull); }  // (never written)
            }
        }
        __syncthreads();
        // ---- D: a wave per record
        for (uint32_t i = wave; i < nrec; i += 4u) {
            const uint32_t L = s_L[i];
            if (L == 0xFFFFFFFFu) continue;
            WideRec R;
            R.g = 0;
            R.rstart = s_start[i];
            R.lh = s_lh[i];
            R.T.p = buf + R.rstart + R.lh + 1;
            R.T.L = L;
            R.T.W = s_W[i];
            uint64_t o = s_off[i];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const uint32_t nk = k < P.nframes ? ts_element_bytes(P, R.lh, L, k) : 0u;
                R.ne[k] = nk;
                R.oe[k] = o;
                o += nk;
            }
            translate_wide_record<64, false, true>(buf, buf_n, t_none, P, nullptr, nullptr, out, nullptr, redo_count, s_pair, s_iu, s_ins, R);
        }
    }
}

}  // namespace

hipError_t launch_translate_stream(int blocks, const uint8_t* buf, uint64_t buf_n, const uint64_t* anchors, uint32_t nranges, uint32_t* queue,
                                   const TranslateParams& P, uint8_t* out, uint64_t out_cap, uint64_t* chain, uint64_t* fin,
                                   uint64_t* redo_count, uint64_t* status, hipStream_t st, uint32_t span_hint) {
    hipLaunchKernelGGL(k_translate_stream, dim3(blocks), dim3(256), 0, st, buf, buf_n, anchors, nranges, queue, P, out, out_cap, chain, fin,
                       redo_count, status, span_hint);
    return hipGetLastError();
}
int translate_stream_max_blocks_per_cu() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_translate_stream, 256, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

hipError_t launch_translate_size(const uint8_t* buf, const RecordTable& t, const TextTableH& tt,
                                 const TranslateParams& P, uint32_t* out_len, uint64_t* status, hipStream_t st) {
    const uint64_t ne = t.n * (uint64_t)P.nframes;
    if (ne == 0) return hipSuccess;
    TextTable d{tt.text_w, tt.lin_off, tt.lin};
    hipLaunchKernelGGL(k_translate_size, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, buf, t, d, P, out_len, status);
    return hipGetLastError();
}

hipError_t launch_translate_frames(int lanes_per_record, const uint8_t* buf, const RecordTable& t, const TextTableH& tt,
                                   const TranslateParams& P, const uint32_t* out_len, const uint64_t* out_off,
                                   uint8_t* out, uint64_t* status, hipStream_t st, uint64_t buf_n, uint8_t* redo, int wide_lanes,
                                   uint64_t* redo_count, int variant, uint64_t* redo_left_stop) {
    if (t.n == 0) return hipSuccess;
    TextTable d{tt.text_w, tt.lin_off, tt.lin};
    const bool v3 = variant == 1, nowide = variant == 2;  // (the context's switch "translate": v3 / frames4)
    if (!v3) {
        // the wide kernel first (plain A/C/G/T text in ordinary layouts), then frames4 for the records it flagged
        const uint8_t* only = nullptr;
        if (redo && !nowide && buf_n >= TRANSLATE_WIDE_MIN_BYTES) {  // (k_translate_wide asks for 52 bytes at `buf` on behalf of idle lanes)
            if (wide_lanes == 64)
                hipLaunchKernelGGL((k_translate_wide<64, false>), dim3((unsigned)((t.n * 64 + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, d,
                                   P, out_len, out_off, out, redo, redo_count, status);
            else if (wide_lanes == 4)
                hipLaunchKernelGGL((k_translate_wide<4, false>), dim3((unsigned)((t.n * 4 + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, d,
                                   P, out_len, out_off, out, redo, redo_count, status);
            else
                hipLaunchKernelGGL((k_translate_wide<16, false>), dim3((unsigned)((t.n * 16 + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, d,
                                   P, out_len, out_off, out, redo, redo_count, status);
            only = redo;
            // how many records are left for k_translate_frames4: usually none, and 2.4 M blocks that look at their flags and
            // leave still cost 2 ms at C4 -- one small read-back instead
            uint64_t left = 0;
            hipError_t e = hipMemcpyAsync(&left, redo_count, sizeof left, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
            if (redo_left_stop) { *redo_left_stop = left; if (left) return hipGetLastError(); }  // (the caller starts over)
            if (left == 0) return hipGetLastError();
        }
        if (lanes_per_record == 64)
            hipLaunchKernelGGL(k_translate_frames4<64>, dim3((unsigned)((t.n * 64 + 255) / 256)), dim3(256), 0, st, buf, t,
                               d, P, out_len, out_off, out, status, only);
        else
            hipLaunchKernelGGL(k_translate_frames4<16>, dim3((unsigned)((t.n * 16 + 255) / 256)), dim3(256), 0, st, buf, t,
                               d, P, out_len, out_off, out, status, only);
        return hipGetLastError();
    }
    if (lanes_per_record == 64) {
        hipLaunchKernelGGL(k_translate_frames<64>, dim3((unsigned)((t.n * 64 + 255) / 256)), dim3(256), 0, st, buf, t, d,
                           P, out_len, out_off, out, status);
    } else {
        hipLaunchKernelGGL(k_translate_frames<16>, dim3((unsigned)((t.n * 16 + 255) / 256)), dim3(256), 0, st, buf, t, d,
                           P, out_len, out_off, out, status);
    }
    return hipGetLastError();
}

hipError_t launch_translate_uniform(int wide_lanes, const uint8_t* buf, uint64_t buf_n, const TranslateParams& P, uint8_t* out,
                                    uint64_t* redo_count, uint64_t* status, hipStream_t st) {
    if (!P.uni.on || P.uni.n == 0) return hipSuccess;
    const RecordTable t{};
    const TextTable d{nullptr, nullptr, nullptr};
    const uint64_t n = P.uni.n;
    if (wide_lanes == 64)
        hipLaunchKernelGGL((k_translate_wide<64, true>), dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, d, P,
                           (const uint32_t*)nullptr, (const uint64_t*)nullptr, out, (uint8_t*)nullptr, redo_count, status);
    else if (wide_lanes == 4)
        hipLaunchKernelGGL((k_translate_wide<4, true>), dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, d, P,
                           (const uint32_t*)nullptr, (const uint64_t*)nullptr, out, (uint8_t*)nullptr, redo_count, status);
    else
        hipLaunchKernelGGL((k_translate_wide<16, true>), dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, d, P,
                           (const uint32_t*)nullptr, (const uint64_t*)nullptr, out, (uint8_t*)nullptr, redo_count, status);
    return hipGetLastError();
}

hipError_t launch_translate_long(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const TranslateParams& P,
                                 const uint32_t* out_len, const uint64_t* out_off, uint8_t* out, uint64_t* status,
                                 uint64_t max_len, hipStream_t st) {
    if (!P.long_count) return hipSuccess;
    TextTable d{tt.text_w, tt.lin_off, tt.lin};
    uint64_t body = max_len / 3 + 2;
    if (P.line_width > 0) body += body / (uint64_t)P.line_width + 1;
    const dim3 grid((unsigned)((body + LONG_BODY - 1) / LONG_BODY), (unsigned)(P.long_count * (uint64_t)P.nframes));
    hipLaunchKernelGGL(k_translate_long, grid, dim3(256), 0, st, buf, t, d, P, out_len, out_off, out, status);
    return hipGetLastError();
}

hipError_t launch_translate_emit(const uint8_t* buf, const RecordTable& t, const TextTableH& tt,
                                  const TranslateParams& P, const uint32_t* out_len, const uint64_t* out_off,
                                  uint8_t* out, uint64_t* status, hipStream_t st) {
    const uint64_t ne = t.n * (uint64_t)P.nframes;
    if (ne == 0) return hipSuccess;
    TextTable d{tt.text_w, tt.lin_off, tt.lin};
    hipLaunchKernelGGL(k_translate_emit, dim3((unsigned)((ne * GROUP + 255) / 256)), dim3(256), 0, st, buf, t, d, P,
                       out_len, out_off, out, status);
    return hipGetLastError();
}

}  // namespace bsk
