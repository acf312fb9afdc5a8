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

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "ops_translate.hpp"
#include "text.cuh"

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
__device__ uint32_t header_len(const uint8_t* h, uint32_t hl, const TranslateParams& P, int frame) {
    if (!P.append_frame) return 1 + hl;
    uint32_t ioff, doff;
    const uint32_t il = id_span_of(h, hl, P.id_mode, &ioff);
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
    uint32_t n = header_len(buf + t.start[i] + 1, lh > 0 ? lh - 1 : 0, P, frame) + 1;
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
            const uint32_t il = id_span_of(h, hl, P.id_mode, &ioff);
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
    const uint32_t H = header_len(h, hl, P, frame) + 1;
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
    const uint32_t H = header_len(h, hl, P, frame) + 1;
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
            const uint32_t il = id_span_of(h, hl, P.id_mode, &ioff);
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
        const uint32_t H = header_len(h, hl, P, frame) + 1;
        const uint32_t body = n - H - 1;
        const uint32_t kept = body - (lw ? body / (lw + 1) : 0u);
        // header and the element's final newline
        if (!P.append_frame) {
            for (uint32_t x = gl; x < H; x += G) o[x] = x == 0 ? (uint8_t)'>' : (x == H - 1 ? (uint8_t)'\n' : h[x - 1]);
        } else if (gl == 0) {
            uint32_t hdr = 0, ioff, doff;
            o[hdr++] = '>';
            const uint32_t il = id_span_of(h, hl, P.id_mode, &ioff);
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

// four residues at `at` = position of the first one; `first` = residues before the line break (>= 4: none inside)
__device__ __forceinline__ void put4(uint8_t* at, uint32_t packed, uint32_t first) {
    if (first >= 4u) {
        *reinterpret_cast<u32_unaligned*>(at) = packed;
    } else {
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) at[(uint32_t)tq + ((uint32_t)tq >= first ? 1u : 0u)] = (uint8_t)(packed >> (8 * tq));
    }
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
#if BSK_TR_WAVES
#define BSK_TR_ATTR __attribute__((amdgpu_waves_per_eu(BSK_TR_WAVES, 8)))
#else
#define BSK_TR_ATTR
#endif

template <int G>
__global__ __launch_bounds__(256) BSK_TR_ATTR void k_translate_frames4(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt,
                                                           TranslateParams P, const uint32_t* __restrict__ out_len,
                                                           const uint64_t* __restrict__ out_off,
                                                           uint8_t* __restrict__ out, uint64_t* __restrict__ status) {
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
    for (int i = threadIdx.x * 16; i < 8192; i += blockDim.x * 16)
        *reinterpret_cast<uint4*>(s_tab + i) = *reinterpret_cast<const uint4*>(P.baked + i);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_iu[i] = P.iupac[i];
    __syncthreads();
    const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const uint32_t gl = threadIdx.x % G;
    if (g >= t.n) return;  // no block-level barrier below
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
        const uint32_t H = header_len(h, hl, P, frame) + 1;
        const uint32_t body = n - H - 1;
        const uint32_t kept = body - (lw ? body / (lw + 1) : 0u);
        if (!P.append_frame) {
            for (uint32_t x = gl; x < H; x += G) o[x] = x == 0 ? (uint8_t)'>' : (x == H - 1 ? (uint8_t)'\n' : h[x - 1]);
        } else if (gl == 0) {
            uint32_t hdr = 0, ioff, doff;
            o[hdr++] = '>';
            const uint32_t il = id_span_of(h, hl, P.id_mode, &ioff);
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
        const uint64_t bodyp = out_off[e] + H;
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
    int64_t rj[3];
    uint32_t ro[3], rcc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rj[c] = (L >= 3u + (uint32_t)c ? (int64_t)((L - 3u - (uint32_t)c) / 3u) : -1) - 4 * (int64_t)gl - 3;
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
            uint32_t cd[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) cd[i] = s_iu[byte_of(w, i)];
            const bool interior = sb + (uint32_t)STEPB + 2u <= L;  // every codon of every lane is complete
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (fb[c] == NONE && rb[c] == NONE) continue;
                uint32_t idx[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) idx[k] = (cd[3 * k + c] << 8) | (cd[3 * k + c + 1] << 4) | cd[3 * k + c + 2];
                uint32_t nv = 4;  // complete codons of the lane in this class
                if (!interior) {
                    nv = 0;
                    if (q + (uint32_t)c + 2u < L) { nv = (L - q - (uint32_t)c - 3u) / 3u + 1u; if (nv > 4u) nv = 4u; }
                }
                if (fb[c] != NONE) {
                    const uint32_t pk = (uint32_t)s_fw[idx[0]] | ((uint32_t)s_fw[idx[1]] << 8) | ((uint32_t)s_fw[idx[2]] << 16) |
                                        ((uint32_t)s_fw[idx[3]] << 24);
                    if (interior) andacc &= pk | (pk << 1);
                    else andacc &= pk | (pk << 1) | (nv >= 4u ? 0u : ~((1u << (8 * nv)) - 1u));
                    uint8_t* body = out + fb[c];
                    if (wide && (uint64_t)sb / 3u + g4 <= fk[c]) put4(body + fj + fo, pk, lw ? lw - fc : 4u);
                    else put4_edge(body, (int64_t)fj, pk, lw, fk[c], fo, fc);
                }
                if (rb[c] != NONE) {
                    // descending residues: codon slot 3 is the lowest residue = byte 0
                    const uint32_t pk = (uint32_t)s_rc[idx[3]] | ((uint32_t)s_rc[idx[2]] << 8) | ((uint32_t)s_rc[idx[1]] << 16) |
                                        ((uint32_t)s_rc[idx[0]] << 24);
                    if (interior) andacc &= pk | (pk << 1);
                    else andacc &= pk | (pk << 1) | (nv >= 4u ? 0u : (nv == 0u ? 0xFFFFFFFFu : ((1u << (8 * (4u - nv))) - 1u)));
                    uint8_t* body = out + rb[c];
                    // lane 0 holds the highest residues of the step, lane G-1 the lowest
                    const int64_t hi0 = rj[c] + 4 * (int64_t)gl + 3, lo0 = hi0 - 4 * (int64_t)G + 1;
                    if (wide && lo0 >= 0 && (uint64_t)hi0 < rk[c]) put4(body + rj[c] + ro[c], pk, lw ? lw - rcc[c] : 4u);
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
            const uint32_t H = header_len(h, hl, P, frame) + 1;
            if (st && out_len[e] > H + 1u) out[out_off[e] + H] = 'M';
        }
    }
}

}  // namespace

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
                                   uint8_t* out, uint64_t* status, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    TextTable d{tt.text_w, tt.lin_off, tt.lin};
    static const bool v3 = [] { const char* e = getenv("BSK_TRANSLATE"); return e && !strcmp(e, "v3"); }();
    if (!v3) {
        if (lanes_per_record == 64)
            hipLaunchKernelGGL(k_translate_frames4<64>, dim3((unsigned)((t.n * 64 + 255) / 256)), dim3(256), 0, st, buf, t,
                               d, P, out_len, out_off, out, status);
        else
            hipLaunchKernelGGL(k_translate_frames4<16>, dim3((unsigned)((t.n * 16 + 255) / 256)), dim3(256), 0, st, buf, t,
                               d, P, out_len, out_off, out, status);
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
