// see ops_concat.hpp
#include <hip/hip_runtime.h>

#include "ops_concat.hpp"
#include "ops_records.hpp"  // ERR_RECORD_TOO_LARGE
#include "text_dev.hpp"

namespace bsk {
namespace {

__device__ __forceinline__ uint64_t lower_bound64(const uint64_t* __restrict__ a, uint64_t lo, uint64_t hi, uint64_t v) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr uint32_t SEG_DIRECT = 0x80000000u;  // seg[3 i + 2]: seg[3 i] is the single mate, not a segment start

__global__ __launch_bounds__(256) void k_concat_segments(const uint64_t* __restrict__ sorted, uint64_t n, uint32_t first2,
                                                         uint32_t* __restrict__ seg) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint64_t e = sorted[p], g = e >> 32;
    const uint32_t i = (uint32_t)e;
    uint64_t s = p, end = p + 1;
    int steps = 0;
    while (s > 0 && (sorted[s - 1] >> 32) == g && steps < 8) { --s; ++steps; }
    if (s > 0 && (sorted[s - 1] >> 32) == g) s = lower_bound64(sorted, 0, s, g << 32);
    steps = 0;
    while (end < n && (sorted[end] >> 32) == g && steps < 8) { ++end; ++steps; }
    if (end < n && (sorted[end] >> 32) == g) end = lower_bound64(sorted, end, n, (g + 1) << 32);
    const uint64_t m1 = lower_bound64(sorted, s, end, (g << 32) | first2) - s;
    const uint64_t m2 = end - s - m1;
    // a record of file 1 with exactly one mate (the usual case) gets the mate itself instead of the segment start:
    // one dependent load less before the emit can begin
    const bool direct = i < first2 && m2 == 1;
    seg[3 * (uint64_t)i] = direct ? (uint32_t)sorted[s + m1] : (uint32_t)s;
    seg[3 * (uint64_t)i + 1] = (uint32_t)m1;
    seg[3 * (uint64_t)i + 2] = direct ? (1u | SEG_DIRECT) : (uint32_t)m2;
}

// the k-th mate (file 2) of a record of file 1
__device__ __forceinline__ uint32_t mate_of(const uint64_t* __restrict__ sorted, uint32_t s, uint32_t m1, uint32_t m2raw, uint32_t k) {
    return (m2raw & SEG_DIRECT) ? s : (uint32_t)sorted[(uint64_t)s + m1 + k];
}

__device__ __forceinline__ uint32_t wrapped(uint32_t L, uint32_t lw) { return L + ((lw && L) ? (L - 1) / lw : 0u); }

// bytes of one element: '>'/'@' + header + '\n' + wrapped sequence + '\n' [+ "+\n" + quality + '\n']
__device__ __forceinline__ uint64_t element_bytes(uint32_t header, uint64_t L, const ConcatParams& P) {
    uint64_t n = 1ull + header + 1ull + (L + ((P.line_width && L) ? (L - 1) / (uint64_t)P.line_width : 0ull)) + 1ull;
    if (P.fastq) n += 2ull + L + 1ull;
    return n;
}

__global__ __launch_bounds__(256) void k_concat_size(const uint8_t* __restrict__ buf, RecordTable t, ConcatParams P,
                                                     const uint64_t* __restrict__ sorted, const uint32_t* __restrict__ seg,
                                                     uint32_t* __restrict__ out_len, uint32_t* __restrict__ count,
                                                     uint64_t* __restrict__ status) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint32_t s = seg[3 * i], m1 = seg[3 * i + 1], m2raw = seg[3 * i + 2], m2 = m2raw & ~SEG_DIRECT;
    const uint8_t* h = buf + t.start[i] + 1;
    const uint32_t lh = t.l_head[i];
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    uint64_t bytes = 0;
    uint32_t cnt = 0;
    if (i < P.first2) {
        if (m2 == 0) {
            if (P.full) { bytes = element_bytes(hl, t.l_seq[i], P); cnt = 1; }   // kept as it is (concat.go:112-127)
        } else {
            uint32_t off;
            const uint32_t il = id_span_rec(t, i, h, hl, P.id_mode, &off, P.buf_end);    // Name: recordA.ID (:133)
            for (uint32_t k = 0; k < m2; ++k) {
                const uint32_t b = mate_of(sorted, s, m1, m2raw, k);
                bytes += element_bytes(il, (uint64_t)t.l_seq[i] + t.l_seq[b], P);
            }
            cnt = m2;
        }
    } else if (m1 == 0 && P.full) {
        bytes = element_bytes(hl, t.l_seq[i], P);
        cnt = 1;
    }
    if (bytes > 0xFFFFFFFFull) {
        atomicOr((unsigned long long*)&status[0], (unsigned long long)ERR_RECORD_TOO_LARGE);
        bytes = 0;
        cnt = 0;
    }
    out_len[i] = (uint32_t)bytes;
    count[i] = cnt;
}


// FASTQ elements as segments of the segmented copy (ops_segcopy.hip): an element "@ID\nSEQA SEQB\n+\nQUALA QUALB\n" is
// five verbatim slices of the shard -- "@ID" of A | "\n" + sequence of A (the newline that ends A's head line) | sequence
// of B + "\n+\n" (B's bare '+' line) | quality of A | quality of B + "\n" -- and a record kept as it is (Full) is one.
// Five slots per element; a record whose elements do not fit that shape (a mate whose '+' line repeats the name, a last
// record without final newline) keeps slot 0 of every element with source 0 and stays with k_concat_emit.
__global__ __launch_bounds__(256) void k_concat_segs(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t, ConcatParams P,
                                                     const uint64_t* __restrict__ sorted, const uint32_t* __restrict__ seg,
                                                     const uint32_t* __restrict__ out_len, const uint64_t* __restrict__ out_off,
                                                     const uint64_t* __restrict__ cnt_off, uint64_t* __restrict__ seg_src,
                                                     uint64_t* __restrict__ seg_off, uint8_t* __restrict__ seg_done,
                                                     unsigned long long* __restrict__ n_other) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    seg_done[i] = 0;
    if (out_len[i] == 0) return;
    const uint32_t s = seg[3 * i], m1 = seg[3 * i + 1], m2raw = seg[3 * i + 2], m2 = m2raw & ~SEG_DIRECT;
    const uint64_t sa = t.start[i];
    const uint32_t lh = t.l_head[i], hl = lh > 0 ? lh - 1 : 0, LA = t.l_seq[i];
    const uint64_t e0 = cnt_off[i];
    uint64_t o = out_off[i];
    const uint64_t A = (uint64_t)(uintptr_t)buf;
    auto put = [&](uint64_t slot, uint64_t src, uint64_t off) { seg_src[slot] = src; seg_off[slot] = off; };
    const bool joined = i < P.first2 && m2 > 0;
    if (!joined) {  // kept as it is: the whole record when Format() reproduces it
        const uint64_t n = element_bytes(hl, LA, P);
        const bool ok = t.aux[i] == 1u && sa + n <= buf_n;
        put(5 * e0, ok ? A + sa : 0ull, o);
        for (int k = 1; k < 5; ++k) put(5 * e0 + k, 0ull, o + n);
        if (ok) seg_done[i] = 1; else atomicAdd(n_other, 1ull);
        return;
    }
    uint32_t off;
    const uint32_t il = id_span_rec(t, i, buf + sa + 1, hl, P.id_mode, &off, P.buf_end);
    bool ok = off == 0;  // "@" + ID is one slice only when the ID begins the head (always, but for --id-ncbi / custom spans)
    for (uint32_t k = 0; k < m2 && ok; ++k) {
        const uint64_t b = mate_of(sorted, s, m1, m2raw, k);
        const uint64_t qb = t.start[b] + t.l_head[b] + 1 + t.l_seq[b] + 1 + t.aux[b] + 1;
        ok = t.aux[b] == 1u && qb + t.l_seq[b] + 1 <= buf_n;
    }
    if (!ok) atomicAdd(n_other, 1ull);
    const uint64_t qa = sa + lh + 1 + LA + 1 + t.aux[i] + 1;
    for (uint32_t k = 0; k < m2; ++k) {
        const uint64_t b = mate_of(sorted, s, m1, m2raw, k);
        const uint32_t LB = t.l_seq[b];
        const uint64_t n = element_bytes(il, (uint64_t)LA + LB, P);
        const uint64_t slot = 5 * (e0 + k);
        if (!ok) {
            put(slot, 0ull, o);
            for (int q = 1; q < 5; ++q) put(slot + q, 0ull, o + n);
        } else {
            const uint64_t sb = t.start[b] + t.l_head[b] + 1;        // sequence of B
            const uint64_t qb = sb + LB + 1 + 1 + 1;                 // its quality (bare '+' line)
            put(slot, A + sa, o);                                    // "@ID"
            put(slot + 1, A + sa + lh, o + 1 + il);                  // "\n" + SEQA
            put(slot + 2, A + sb, o + 1 + il + 1 + LA);              // SEQB + "\n+\n"
            put(slot + 3, A + qa, o + 1 + il + 1 + LA + LB + 3);     // QUALA
            put(slot + 4, A + qb, o + 1 + il + 1 + LA + LB + 3 + LA);  // QUALB + "\n"
        }
        o += n;
    }
    if (ok) seg_done[i] = 1;
}

// FASTQ quality of record i (strict 4-line layout)
__device__ __forceinline__ const uint8_t* qual_of(const uint8_t* __restrict__ buf, const RecordTable& t, uint64_t i) {
    return buf + t.start[i] + t.l_head[i] + 1 + t.l_seq[i] + 1 + t.aux[i] + 1;
}

// (not inlined: with this function inlined into k_concat_emit, clang 22 of ROCm 7.2 crashes in the gfx950 backend)
// one element written by G lanes: header bytes from `hd`, sequence A ++ B (B may be empty), quality likewise
template <int G>
__device__ __noinline__ uint64_t put_element(uint8_t* __restrict__ o, uint32_t gl, const uint8_t* hd, uint32_t hlen,
                                                const Text& TA, const Text& TB, const uint8_t* qa, const uint8_t* qb,
                                                const ConcatParams& P) {
    const uint32_t LA = TA.L, L = TA.L + TB.L;
    const uint32_t lw = (uint32_t)P.line_width;
    const uint32_t W = wrapped(L, lw);
    uint64_t x0 = 0;
    for (uint32_t x = gl; x < hlen + 2u; x += G)
        o[x] = x == 0 ? (uint8_t)(P.fastq ? '@' : '>') : (x == hlen + 1u ? (uint8_t)'\n' : hd[x - 1]);
    x0 = hlen + 2u;
    // contiguous sources and no newline to insert (every FASTQ, single-line FASTA): the element is a chain of plain
    // spans, copied 16 bytes per lane
    if (W == L && TA.W == 0 && TB.W == 0) {
        auto span = [&](uint64_t off, const uint8_t* src, uint32_t nb) {
            uint8_t* dst = o + off;
            for (uint32_t x = gl * 16u; x < nb; x += G * 16u) {
                if (x + 16u <= nb) {
                    uint4 v;
                    __builtin_memcpy(&v, src + x, 16);
                    __builtin_memcpy(dst + x, &v, 16);
                } else {
                    for (uint32_t k = x; k < nb; ++k) dst[k] = src[k];
                }
            }
        };
        span(x0, TA.p, LA);
        span(x0 + LA, TB.p, TB.L);
        x0 += L;
        if (gl == 0) o[x0] = '\n';
        x0 += 1;
        if (P.fastq) {
            if (gl == 0) { o[x0] = '+'; o[x0 + 1] = '\n'; }
            x0 += 2;
            span(x0, qa, LA);
            span(x0 + LA, qb, TB.L);
            x0 += L;
            if (gl == 0) o[x0] = '\n';
            x0 += 1;
        }
        return x0;
    }
    for (uint32_t x = gl; x < W; x += G) {
        uint8_t c;
        if (lw && (x % (lw + 1u)) == lw) c = '\n';
        else {
            const uint32_t j = lw ? x - x / (lw + 1u) : x;
            c = j < LA ? TA.at(j) : TB.at(j - LA);
        }
        o[x0 + x] = c;
    }
    x0 += W;
    if (gl == 0) o[x0] = '\n';
    x0 += 1;
    if (P.fastq) {
        if (gl == 0) { o[x0] = '+'; o[x0 + 1] = '\n'; }
        x0 += 2;
        for (uint32_t x = gl; x < L; x += G) o[x0 + x] = x < LA ? qa[x] : qb[x - LA];
        x0 += L;
        if (gl == 0) o[x0] = '\n';
        x0 += 1;
    }
    return x0;
}

template <int G>
__global__ __launch_bounds__(256) void k_concat_emit(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt, ConcatParams P,
                                                     const uint64_t* __restrict__ sorted, const uint32_t* __restrict__ seg,
                                                     const uint32_t* __restrict__ out_len, const uint64_t* __restrict__ out_off,
                                                     uint8_t* __restrict__ out, const uint8_t* __restrict__ seg_done) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const uint32_t gl = threadIdx.x % G;
    if (i >= t.n || out_len[i] == 0) return;
    if (seg_done && seg_done[i]) return;  // written by the segmented copy (k_concat_segs)
    const uint32_t s = seg[3 * i], m1 = seg[3 * i + 1], m2raw = seg[3 * i + 2], m2 = m2raw & ~SEG_DIRECT;
    const uint8_t* h = buf + t.start[i] + 1;
    const uint32_t lh = t.l_head[i];
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    uint8_t* o = out + out_off[i];
    const Text TA = text_of(buf, t, tt, i);
    const uint8_t* qa = P.fastq ? qual_of(buf, t, i) : nullptr;
    if (i < P.first2 && m2 > 0) {
        uint32_t off;
        const uint32_t il = id_span_rec(t, i, h, hl, P.id_mode, &off, P.buf_end);
        for (uint32_t k = 0; k < m2; ++k) {
            const uint64_t b = mate_of(sorted, s, m1, m2raw, k);
            const Text TB = text_of(buf, t, tt, b);
            o += put_element<G>(o, gl, h + off, il, TA, TB, qa, P.fastq ? qual_of(buf, t, b) : nullptr, P);
        }
    } else {
        Text none;
        none.p = TA.p; none.L = 0; none.W = 0;
        put_element<G>(o, gl, h, hl, TA, none, qa, nullptr, P);
    }
}

}  // namespace

hipError_t launch_concat_segments(const uint64_t* sorted, uint64_t n, uint32_t first2, uint32_t* seg, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_concat_segments, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sorted, n, first2, seg);
    return hipGetLastError();
}

hipError_t launch_concat_size(const uint8_t* buf, const RecordTable& t, const ConcatParams& P, const uint64_t* sorted,
                              const uint32_t* seg, uint32_t* out_len, uint32_t* count, uint64_t* status, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_concat_size, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, t, P, sorted, seg, out_len, count, status);
    return hipGetLastError();
}

hipError_t launch_concat_emit(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const ConcatParams& P,
                              const uint64_t* sorted, const uint32_t* seg, const uint32_t* out_len, const uint64_t* out_off,
                              uint8_t* out, uint64_t avg_bytes, hipStream_t st, const uint8_t* seg_done) {
    if (t.n == 0) return hipSuccess;
    TextTable d{tt.text_w, tt.lin_off, tt.lin};
    // short reads: 4 lanes per record (a 150-byte span is 10 steps of 16 bytes), otherwise 16
    if (avg_bytes < 1024)
        hipLaunchKernelGGL((k_concat_emit<4>), dim3((unsigned)((t.n * 4 + 255) / 256)), dim3(256), 0, st, buf, t, d, P, sorted, seg,
                           out_len, out_off, out, seg_done);
    else
        hipLaunchKernelGGL((k_concat_emit<16>), dim3((unsigned)((t.n * 16 + 255) / 256)), dim3(256), 0, st, buf, t, d, P, sorted, seg,
                           out_len, out_off, out, seg_done);
    return hipGetLastError();
}

hipError_t launch_concat_segs(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const ConcatParams& P, const uint64_t* sorted,
                              const uint32_t* seg, const uint32_t* out_len, const uint64_t* out_off, const uint64_t* cnt_off,
                              uint64_t* seg_src, uint64_t* seg_off, uint8_t* seg_done, uint64_t* n_other, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_concat_segs, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, P, sorted, seg, out_len, out_off,
                       cnt_off, seg_src, seg_off, seg_done, (unsigned long long*)n_other);
    return hipGetLastError();
}

}  // namespace bsk
