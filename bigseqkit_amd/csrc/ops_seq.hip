// ============================================================================
// ops_seq.hip -- SeqTransform.Call (/root/reference/bigseqkit-lib/seq.go:81-269)
// on the record table.
//   k_seq_size : one thread per record -> bytes the record contributes (0 = filtered)
//   (exclusive scan of the sizes, stream_index.hip)
//   k_seq_emit : 16 lanes per record; every output byte is a pure function of
//                (record, output offset): header / wrapped sequence / "+" / quality,
//                reverse as an index map, complement + dna<->rna + case as ONE
//                256-byte lookup.  Gap removal and multi-line FASTA sources take a
//                sequential per-record path.
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ops_seq.hpp"
#include "pattern_match_dev.hpp"  // fnv1a64, TEXT_IRREGULAR (text_dev.hpp)

namespace bsk {

namespace {

__device__ __forceinline__ bool in_set(const uint32_t* set, uint8_t c) { return (set[c >> 5] >> (c & 31)) & 1u; }

struct RecView {
    const uint8_t* head;  // after the marker
    uint32_t head_len;
    const uint8_t* seq;   // FASTQ: contiguous; FASTA: region that may contain '\n'
    uint32_t seq_len;     // bases
    uint32_t region;      // FASTA: bytes of the region
    const uint8_t* qual;
    bool contiguous;      // sequence bytes are contiguous in memory
};

__device__ __forceinline__ RecView view(const uint8_t* buf, const RecordTable& t, uint64_t i, int fastq) {
    RecView r;
    const uint64_t s = t.start[i];
    const uint32_t lh = t.l_head[i];
    r.head = buf + s + 1;
    r.head_len = lh > 0 ? lh - 1 : 0;
    r.seq = buf + s + lh + 1;
    r.seq_len = t.l_seq[i];
    if (fastq) {
        r.region = r.seq_len;
        r.qual = r.seq + r.seq_len + 1 + t.aux[i] + 1;
        r.contiguous = true;
    } else {
        r.region = t.aux[i];
        r.qual = nullptr;
        // contiguous iff the region holds no newline except an optional final one
        const uint32_t tail_nl = (r.region > 0 && r.seq[r.region - 1] == '\n') ? 1u : 0u;
        r.contiguous = r.region == r.seq_len + tail_nl;
    }
    return r;
}

__device__ __forceinline__ uint32_t wrapped_len(uint32_t L, int w) {
    if (w < 1 || L == 0) return L;
    return L + (L - 1) / (uint32_t)w;
}

// feature of a record (subseq --gtf/--bed): index into the f_* arrays or -1
__device__ int feature_of(const SeqParams& P, const uint8_t* id, uint32_t id_len) {
    const bool fold = !P.feat_query || P.feat_fold;  // subseq lower-cases the names; faidx only with -i
    const uint64_t key = fnv1a64(id, id_len, fold);
    for (uint64_t slot = key & P.fset_mask;; slot = (slot + 1) & P.fset_mask) {
        const uint64_t sk = P.fset_keys[slot];
        if (sk == 0) return -1;
        if (sk != key) continue;
        const uint32_t f = P.fset_idx[slot];
        const uint32_t o = P.fname_off[f];
        if (P.fname_off[f + 1] - o != id_len) continue;
        bool ok = true;
        for (uint32_t q = 0; q < id_len; ++q) {
            uint8_t c = id[q];
            if (fold && c >= 'A' && c <= 'Z') c += 32;
            if (c != P.fname[o + q]) { ok = false; break; }
        }
        if (ok) return (int)f;
    }
}

// region of the feature on a record of length L: subseq.go:339-376 then Seq.SubSeq; [b, e) 0-based
__device__ __forceinline__ void feature_region(const SeqParams& P, int f, uint32_t L, uint32_t* b, uint32_t* e) {
    int64_t s = P.f_s[f], t = P.f_e[f];
    if (P.feat_query) { sub_location(L, (int)s, (int)t, b, e); return; }  // seq.SubLocation (faidx.go:391-402)
    if (s < 1) s = 1;
    if (t > (int64_t)L) t = L;
    *b = *e = 0;
    if (t < 1 || s > (int64_t)L || s > t) return;  // SubSeq of positive bounds; t < 1: empty (PARITY.md SUB0)
    *b = (uint32_t)(s - 1);
    *e = (uint32_t)t;
}

// Where the bytes of an output header come from: the record's own head (whole, or the ID only), the ID followed by a
// feature suffix (subseq --gtf / --bed), or the ID followed by "_<ord> " and the description (rename).
struct HeadSrc {
    const uint8_t* head;     // after the marker
    const uint8_t* suffix;   // feature mode
    uint32_t hoff, id_len;   // ID span inside head
    uint32_t ord, ndig;      // rename: ordinal (0: header unchanged) and its decimal digits
    uint32_t desc_off;       // rename: first byte of the description inside head
    uint32_t len;            // header bytes
    __device__ __forceinline__ uint8_t at(uint32_t k) const {
        if (k < id_len) return head[hoff + k];
        if (suffix) return suffix[k - id_len];
        if (ord) {
            const uint32_t q = k - id_len;
            if (q == 0) return '_';
            if (q <= ndig) {
                uint32_t v = ord;
                for (uint32_t z = ndig - q; z; --z) v /= 10u;
                return (uint8_t)('0' + v % 10u);
            }
            if (q == ndig + 1u) return ' ';
            return head[desc_off + (q - ndig - 2u)];
        }
        return head[hoff + k];
    }
};

// rename: header of record g (ord > 0); returns false when the header stays as it is
__device__ __forceinline__ bool rename_head(const RecordTable& t, const SeqParams& P, uint64_t g, const uint8_t* head, uint32_t head_len,
                                            HeadSrc* H) {
    if (!P.ren_ord) return false;
    const uint32_t ord = P.ren_ord[g];
    if (!ord) return false;
    uint32_t hoff, doff;
    const uint32_t il = id_span_rec(t, g, head, head_len, P.id_mode, &hoff, P.buf_end);
    // Desc of parseHeadIDAndDesc (helper.go:329-369): only the default regexp yields one
    const uint32_t dl = hoff == 0 ? desc_of(head, head_len, P.id_mode, il, &doff) : 0u;
    uint32_t nd = 1;
    for (uint32_t v = ord; v >= 10u; v /= 10u) ++nd;
    H->hoff = hoff; H->id_len = il; H->ord = ord; H->ndig = nd; H->desc_off = doff;
    H->len = il + 1u + nd + 1u + dl;  // ID '_' digits ' ' Desc   (fmt.Sprintf("%s %s", newID, record.Desc))
    return true;
}

__global__ __launch_bounds__(256) void k_seq_size(const uint8_t* __restrict__ buf, RecordTable t, SeqParams P,
                                                  uint32_t* __restrict__ out_len, uint64_t* __restrict__ status) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const RecView r = view(buf, t, i, P.fastq);
    uint32_t L = r.seq_len;
    uint32_t err = 0;
    if (P.validate) {
        // (16 bytes per load, the letters tested in registers: a byte load per base on one lane was 45 ms per 5 GB of reads)
        const uint32_t limit = P.validate_len <= 0 ? 0xFFFFFFFFu : (uint32_t)P.validate_len;
        uint32_t seen = 0, k = 0;
        bool bad = false;
        for (; !bad && seen < limit && k + 16u <= r.region && r.seq + k + 16 <= P.buf_end; k += 16u) {
            uint4 v;
            __builtin_memcpy(&v, r.seq + k, 16);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const uint8_t c = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
                const bool counts = !(c == '\n' && !P.fastq) && seen < limit;
                if (counts) { ++seen; bad |= !in_set(P.valid_set, c); }
            }
        }
        for (; !bad && k < r.region && seen < limit; ++k) {
            const uint8_t c = r.seq[k];
            if (c == '\n' && !P.fastq) continue;
            ++seen;
            bad = !in_set(P.valid_set, c);
        }
        if (bad) err |= ERR_INVALID_LETTER;
    }
    uint32_t kept = L;
    if (P.remove_gaps) {
        // 16 bytes per load; a chunk without any byte below 64 holds neither a gap letter ('-', '.', ' ', '*' ...) nor a line
        // end and counts as a whole (one byte load per base on one lane made this loop 40 ms per 5 GB of reads)
        kept = 0;
        uint32_t k = 0;
        for (; k + 16u <= r.region && r.seq + k + 16 <= P.buf_end; k += 16u) {
            uint4 v;
            __builtin_memcpy(&v, r.seq + k, 16);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            bool low = !P.gap_lt64;
#pragma unroll
            for (int d = 0; d < 4; ++d) { const uint32_t nx = ~w[d]; low |= (nx & (nx << 1) & 0x80808080u) != 0u; }
            if (!low) { kept += 16u; continue; }
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const uint8_t c = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
                if (c == '\n' && !P.fastq) continue;
                if (!in_set(P.gap_set, c)) ++kept;
            }
        }
        for (; k < r.region; ++k) {
            const uint8_t c = r.seq[k];
            if (c == '\n' && !P.fastq) continue;
            if (!in_set(P.gap_set, c)) ++kept;
        }
    }
    if (P.region_on) {
        uint32_t b, e;
        sub_location(L, P.region_start, P.region_end, &b, &e);
        kept = e - b;
    }
    if (P.feat_on) {
        uint32_t off;
        const uint32_t il = id_span_rec(t, i, r.head, r.head_len, P.id_mode, &off, P.buf_end);
        const int f = feature_of(P, r.head + off, il);
        if (f < 0) { out_len[i] = 0; return; }
        uint32_t b, e;
        feature_region(P, f, L, &b, &e);
        kept = e - b;
        if (P.feat_query && kept == 0) { out_len[i] = 0; return; }  // SubLocation !ok: the record is skipped
        const uint32_t hl = il + (P.fsuffix_off[f + 1] - P.fsuffix_off[f]);
        out_len[i] = 1u + hl + 1u + wrapped_len(kept, P.line_width) + 1u + (P.print_qual ? 2u + kept + 1u : 0u);
        return;
    }
    bool keep = true;
    if (P.min_len > 0 && (int64_t)kept < P.min_len) keep = false;
    if (P.max_len > 0 && (int64_t)kept > P.max_len) keep = false;
    if (keep && (P.min_qual > 0 || P.max_qual > 0)) {
        // Seq.AvgQual: mean error probability, summed sequentially in float64
        double aq = 0;
        if (P.fastq && kept > 0) {
            double sum = 0;
            uint32_t k = 0;
            // the sum stays sequential (float64, in record order: Seq.AvgQual); only the bytes come 16 at a time
            for (; k + 16u <= L && r.qual + k + 16 <= P.buf_end && r.seq + k + 16 <= P.buf_end; k += 16u) {
                uint4 q4, s4;
                __builtin_memcpy(&q4, r.qual + k, 16);
                const uint32_t qw[4] = {q4.x, q4.y, q4.z, q4.w};
                uint32_t sw[4] = {0, 0, 0, 0};
                if (P.remove_gaps) { __builtin_memcpy(&s4, r.seq + k, 16); sw[0] = s4.x; sw[1] = s4.y; sw[2] = s4.z; sw[3] = s4.w; }
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    if (P.remove_gaps && in_set(P.gap_set, (uint8_t)(sw[b >> 2] >> (8 * (b & 3))))) continue;
                    sum += P.qual_err[(uint8_t)(qw[b >> 2] >> (8 * (b & 3)))];
                }
            }
            for (; k < L; ++k) {
                if (P.remove_gaps && in_set(P.gap_set, r.seq[k])) continue;
                sum += P.qual_err[r.qual[k]];
            }
            aq = -10.0 * log10(sum / (double)kept);
        }
        if (P.min_qual > 0 && aq < P.min_qual) keep = false;
        if (P.max_qual > 0 && aq >= P.max_qual) keep = false;
    }
    uint32_t n = 0;
    if (keep) {
        if (P.print_name) {
            uint32_t hl = r.head_len, off;
            if (P.only_id) hl = id_span_rec(t, i, r.head, r.head_len, P.id_mode, &off, P.buf_end);
            HeadSrc H;
            if (rename_head(t, P, i, r.head, r.head_len, &H)) hl = H.len;
            n += (P.print_seq ? 1u : 0u) + hl + 1u;
        }
        if (P.print_seq) n += wrapped_len(kept, P.line_width) + 1u;
        if (P.print_qual) n += (P.qual_only ? 0u : 2u) + kept + 1u;
    }
    out_len[i] = n;
    if (err) atomicOr((unsigned long long*)&status[0], (unsigned long long)err);
}

// GROUP lanes per record: 16 for long records, 4 when the average output record is small (a 16-lane group would sit
// idle on a 120-byte `subseq` record or a 317-byte read: 4x the waves for the same bytes)
// LONG: records whose output is at least P.long_thresh bytes (chromosomes) are skipped by the per-record kernel and
// written by whole blocks instead, one block per LONG_CH output bytes of one record (grid = chunks x long records).
constexpr uint32_t LONG_CH = 64u * 1024u;

template <int GROUP, bool LONG>
__global__ __launch_bounds__(256) void k_seq_emit(const uint8_t* __restrict__ buf, RecordTable t, SeqParams P,
                                                  const uint32_t* __restrict__ out_len,
                                                  const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out) {
    // byte map (complement / case / dna<->rna) in LDS for the 16-bytes-per-step transform path
    __shared__ uint8_t s_lut[256];
    if (P.use_lut || P.feat_on) {
        const uint8_t* src = P.use_lut ? P.lut : P.comp;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) s_lut[i] = src ? src[i] : (uint8_t)i;
        __syncthreads();
    }
    constexpr uint32_t LANES = LONG ? 256u : (uint32_t)GROUP;  // lanes that share one record (or one chunk of it)
    const uint64_t g = LONG ? (uint64_t)P.long_list[blockIdx.y] : ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
    const uint32_t gl = LONG ? threadIdx.x : threadIdx.x % GROUP;
    if (g >= t.n) return;
    const uint32_t n = out_len[g];
    if (n == 0) return;
    if (P.seg_src && P.seg_src[g]) return;                      // written by k_seg_copy
    if (!LONG && P.long_thresh && n >= P.long_thresh) return;  // written by the LONG launch
    const uint32_t clo = LONG ? blockIdx.x * LONG_CH : 0u;       // this block's slice [clo, chi) of the record's output
    if (clo >= n) return;
    const uint32_t chi = LONG ? (n - clo < LONG_CH ? n : clo + LONG_CH) : n;
    // 16-byte steps of a segment that sits at output offset `off`, restricted to the slice (a step that straddles a
    // slice boundary is written by both neighbours with the same bytes)
    auto first_step = [&](uint32_t off) -> uint32_t { return clo > off ? ((clo - off) & ~15u) : 0u; };
    auto last_byte = [&](uint32_t off, uint32_t nb) -> uint32_t { return chi < off ? 0u : (chi - off < nb ? chi - off : nb); };
    auto mine = [&](uint32_t x) -> bool { return x >= clo && x < chi; };  // single bytes
    uint8_t* o = out + out_off[g];
    const RecView r = view(buf, t, g, P.fastq);
    uint32_t hl = r.head_len, hoff = 0;
    if (P.print_name && P.only_id) hl = id_span_rec(t, g, r.head, r.head_len, P.id_mode, &hoff, P.buf_end);
    uint32_t sub_b = 0, sub_e = r.seq_len;
    if (P.region_on) sub_location(r.seq_len, P.region_start, P.region_end, &sub_b, &sub_e);
    bool reverse = P.reverse != 0, use_lut = P.use_lut != 0;
    const uint8_t* lut = P.lut;
    const uint8_t* suffix = nullptr;  // feature mode: header = ID + suffix
    uint32_t id_len = hl;
    if (P.feat_on) {
        id_len = id_span_rec(t, g, r.head, r.head_len, P.id_mode, &hoff, P.buf_end);
        const int f = feature_of(P, r.head + hoff, id_len);
        feature_region(P, f, r.seq_len, &sub_b, &sub_e);
        suffix = P.fsuffix + P.fsuffix_off[f];
        hl = id_len + (P.fsuffix_off[f + 1] - P.fsuffix_off[f]);
        if (P.f_minus[f]) { reverse = true; use_lut = true; lut = P.comp; }
    }
    HeadSrc HS;
    HS.head = r.head; HS.suffix = suffix; HS.hoff = hoff; HS.id_len = id_len; HS.ord = 0; HS.ndig = 0; HS.desc_off = 0;
    if (rename_head(t, P, g, r.head, r.head_len, &HS)) hl = HS.len;
    HS.len = hl;
    const uint32_t a = P.print_name ? (P.print_seq ? 1u : 0u) + hl + 1u : 0u;
    // random access to the bases: contiguous text, or a wrapped FASTA record through the text view
    const uint8_t* sp = r.seq;
    uint32_t TW = 0;
    bool random_access = r.contiguous;
    if (!random_access && P.text_w) {
        const uint32_t w = P.text_w[g];
        if (w == TEXT_IRREGULAR) sp = P.lin + P.lin_off[g];
        else TW = w;
        random_access = true;
    }
    // gap removal: a record WITHOUT a gap letter (nearly all) is written by the parallel paths -- its output size says so
    bool rm_gaps = P.remove_gaps != 0;
    if (rm_gaps) {
        const uint32_t k0 = sub_e - sub_b;
        const uint32_t expect = (P.print_seq ? wrapped_len(k0, P.line_width) + 1u : 0u) + (P.print_qual ? (P.qual_only ? 0u : 2u) + k0 + 1u : 0u);
        if (n - a == expect) rm_gaps = false;
    }
    const bool fast = random_access && !rm_gaps;
    // Whole FASTQ record printed unchanged (grep / rmdup / plain seq): Format() reproduces the
    // record text byte for byte when the '+' line is bare, so copy it 16 bytes per lane.
    if (fast && P.fastq && P.print_name && P.print_seq && P.print_qual && !P.qual_only && !P.only_id && !reverse &&
        !use_lut && !P.region_on && !P.feat_on && t.aux[g] == 1 && HS.ord == 0) {
        const uint8_t* src = buf + t.start[g];
        const uint32_t body = n - 1;  // everything but the final newline, which the shard may lack
        const uint32_t hi = last_byte(0, body);
        for (uint32_t x0 = first_step(0) + gl * 16u; x0 < hi; x0 += LANES * 16u) {
            const uint32_t x = (!LONG && x0 + 16u > body && body >= 16u) ? body - 16u : x0;  // (see xform_copy)
            if (x + 16u <= body) {
                uint4 v;
                __builtin_memcpy(&v, src + x, 16);
                __builtin_memcpy(o + x, &v, 16);
            } else {
                for (uint32_t k = x; k < body; ++k) o[k] = src[k];
            }
        }
        if (gl == 0 && mine(body)) o[body] = '\n';
        return;
    }
    if (fast) {
        const uint32_t L = sub_e - sub_b;
        const uint8_t* rqual = r.qual ? r.qual + sub_b : nullptr;
        const uint32_t W = wrapped_len(L, P.line_width);
        const uint32_t b = P.print_seq ? W + 1u : 0u;
        const uint32_t w1 = (uint32_t)(P.line_width > 0 ? P.line_width + 1 : 0);
        // The source bytes already have the output layout when nothing is transformed and either no newline has to
        // be inserted (contiguous source) or the source is wrapped at the output width and the region starts at
        // base 0: header byte-wise, sequence and quality as 16-byte copies.
        const bool verbatim = !reverse && !use_lut &&
                              ((TW == 0 && W == L) || (TW != 0 && (int)TW == P.line_width && sub_b == 0));
        auto put_head = [&]() {
            if (!LONG && a && !HS.suffix && !HS.ord) {
                // the head (or the ID) is one slice of the record: marker and newline by lane 0, the bytes 16 at a time with
                // the last step taken from the slice's end, or -- below 16 bytes -- 8 / 4 / 2 / 1 by one lane (a byte per
                // lane and turn cost the group four turns for a 12-byte name)
                const uint32_t m = P.print_seq ? 1u : 0u, nb = a - 1u - m;
                const uint8_t* hs = HS.head + HS.hoff;
                uint8_t* dst = o + m;
                if (gl == 0) {
                    if (m) o[0] = (P.fastq && !P.fasta_out) ? '@' : '>';
                    o[a - 1] = '\n';
                }
                if (nb >= 16u) {
                    for (uint32_t x0 = gl * 16u; x0 < nb; x0 += LANES * 16u) {
                        const uint32_t x = x0 + 16u > nb ? nb - 16u : x0;
                        uint4 v;
                        __builtin_memcpy(&v, hs + x, 16);
                        __builtin_memcpy(dst + x, &v, 16);
                    }
                } else if (gl == (LANES > 1u ? 1u : 0u)) {
                    uint32_t k = 0;
                    if (nb & 8u) { uint2 v; __builtin_memcpy(&v, hs, 8); __builtin_memcpy(dst, &v, 8); k = 8u; }
                    if (nb & 4u) { uint32_t v; __builtin_memcpy(&v, hs + k, 4); __builtin_memcpy(dst + k, &v, 4); k += 4u; }
                    if (nb & 2u) { uint16_t v; __builtin_memcpy(&v, hs + k, 2); __builtin_memcpy(dst + k, &v, 2); k += 2u; }
                    if (nb & 1u) dst[k] = hs[k];
                }
            } else
            for (uint32_t x = clo + gl; x < (a < chi ? a : chi); x += LANES) {
                const uint32_t m = P.print_seq ? 1u : 0u;
                uint8_t c;
                if (x < m) c = (P.fastq && !P.fasta_out) ? '@' : '>';
                else if (x == a - 1) c = '\n';
                else c = HS.at(x - m);
                o[x] = c;
            }
        };
        if (verbatim) {
            put_head();
            auto group_copy = [&](uint32_t off, const uint8_t* src, uint32_t nb) {
                uint8_t* dst = o + off;
                const uint32_t hi = last_byte(off, nb);
                for (uint32_t x0 = first_step(off) + gl * 16u; x0 < hi; x0 += LANES * 16u) {
                    const uint32_t x = (!LONG && x0 + 16u > nb && nb >= 16u) ? nb - 16u : x0;  // (see xform_copy)
                    if (x + 16u <= nb) {
                        uint4 v;
                        __builtin_memcpy(&v, src + x, 16);
                        __builtin_memcpy(dst + x, &v, 16);
                    } else {
                        for (uint32_t k = x; k < nb; ++k) dst[k] = src[k];
                    }
                }
            };
            if (P.print_seq) {
                group_copy(a, sp + (TW ? 0u : sub_b), W);
                if (gl == 0 && mine(a + W)) o[a + W] = '\n';
            }
            if (P.print_qual) {
                uint32_t q0 = a + b;
                if (!P.qual_only) {
                    if (gl == 0 && mine(q0)) o[q0] = '+';
                    if (gl == 0 && mine(q0 + 1)) o[q0 + 1] = '\n';
                    q0 += 2;
                }
                group_copy(q0, rqual, L);
                if (gl == 0 && mine(q0 + L)) o[q0 + L] = '\n';
            }
            return;
        }
        // Transformed but not re-wrapped (reverse and / or the byte map on a contiguous source, no newline to insert):
        // 16 output bytes per lane and step -- one 16-byte load, a byte reversal with v_perm, the map through LDS.
        // (The LDS map holds P.lut, or the complement of a '-' strand feature; both are never active together.)
        const bool lut_in_lds = use_lut && (lut == P.lut ? P.use_lut != 0 : true);
        if (TW == 0 && W == L && (!use_lut || lut_in_lds)) {
            put_head();
            auto xform_copy = [&](uint32_t off, const uint8_t* src, uint32_t nb, bool map) {
                uint8_t* dst = o + off;
                const uint32_t hi = last_byte(off, nb);
                for (uint32_t x0 = first_step(off) + gl * 16u; x0 < hi; x0 += LANES * 16u) {
                    // the last, partial step of a span of 16 bytes or more is taken 16 bytes wide from the span's end: it
                    // overlaps the step before with the same bytes (a byte loop here kept one lane busy for up to 15 turns
                    // per span while its group waited)
                    const uint32_t x = (!LONG && x0 + 16u > nb && nb >= 16u) ? nb - 16u : x0;
                    if (x + 16u <= nb) {
                        uint4 v;
                        __builtin_memcpy(&v, src + (reverse ? nb - 16u - x : x), 16);
                        uint32_t w[4] = {v.x, v.y, v.z, v.w};
                        if (reverse) {
                            const uint32_t r0 = __builtin_bswap32(w[3]), r1 = __builtin_bswap32(w[2]), r2 = __builtin_bswap32(w[1]),
                                           r3 = __builtin_bswap32(w[0]);
                            w[0] = r0; w[1] = r1; w[2] = r2; w[3] = r3;
                        }
                        if (map) {
#pragma unroll
                            for (int d = 0; d < 4; ++d)
                                w[d] = (uint32_t)s_lut[w[d] & 0xFFu] | ((uint32_t)s_lut[(w[d] >> 8) & 0xFFu] << 8) |
                                       ((uint32_t)s_lut[(w[d] >> 16) & 0xFFu] << 16) | ((uint32_t)s_lut[w[d] >> 24] << 24);
                        }
                        const uint4 ov = make_uint4(w[0], w[1], w[2], w[3]);
                        __builtin_memcpy(dst + x, &ov, 16);
                    } else {
                        for (uint32_t k = x; k < nb; ++k) {
                            uint8_t c = src[reverse ? nb - 1u - k : k];
                            if (map) c = s_lut[c];
                            dst[k] = c;
                        }
                    }
                }
            };
            if (P.print_seq) {
                xform_copy(a, sp + sub_b, L, use_lut);
                if (gl == 0 && mine(a + L)) o[a + L] = '\n';
            }
            if (P.print_qual) {
                uint32_t q0 = a + b;
                if (!P.qual_only) {
                    if (gl == 0 && mine(q0)) o[q0] = '+';
                    if (gl == 0 && mine(q0 + 1)) o[q0 + 1] = '\n';
                    q0 += 2;
                }
                xform_copy(q0, rqual, L, false);
                if (gl == 0 && mine(q0 + L)) o[q0 + L] = '\n';
            }
            return;
        }
        // FASTA whose line layout changes (unwrapped, re-wrapped at another width, cut to a region) or whose bases are mapped
        // (case, dna <-> rna, complement, reversed), source lines and output lines of at least 16 bases: OUTPUT-driven, a
        // lane writes 16 consecutive output bytes per step.  Those hold at most one output newline and come from at most 17
        // consecutive source bytes with at most one source newline in them, both at positions known from two divisions:
        // 32 source bytes in registers, the source newline squeezed out, (reverse: the 16 bytes mirrored,) the map through
        // LDS, the output newline opened up -- all dword-wise with v_alignbyte and masks.  Stores of a group are contiguous.
        // (The source-driven version below cut a source line at the output line ends and copied the pieces: pieces under
        // 16 bytes went byte by byte, 24 ms for 10 GB re-wrapped from 60 to 70.)
        {
            const uint32_t lw = (uint32_t)(P.line_width > 0 ? P.line_width : 0);
            if (!P.fastq && P.print_seq && !P.print_qual && (TW == 0 || TW >= 16u) && (lw == 0 || lw >= 16u) && (!use_lut || lut_in_lds) &&
                W >= 16u) {
                put_head();
                uint8_t* d0 = o + a;
                const uint32_t text_len = r.seq_len + (TW && r.seq_len ? (r.seq_len - 1u) / TW : 0u);  // source bytes of the record's text
                const bool wide_ok = sp >= buf && sp + text_len <= P.buf_end;
                const uint32_t hi = last_byte(a, W);
                for (uint32_t x0 = first_step(a) + gl * 16u; x0 < hi; x0 += LANES * 16u) {
                    const uint32_t x = x0 + 16u > W ? W - 16u : x0;  // the last step is taken from the region's end
                    uint32_t q0 = x, k_out = 16u;
                    if (lw) {
                        const uint32_t ol = x / w1, col = x - ol * w1;  // col <= lw; == lw: the step starts on the newline
                        q0 = ol * lw + (col < lw ? col : lw);
                        k_out = lw - col;
                    }
                    const uint32_t nbases = k_out < 16u ? 15u : 16u;
                    const uint32_t lo_b = reverse ? sub_e - q0 - nbases : sub_b + q0;  // first (lowest) source base of the step
                    uint32_t s0 = lo_b, k_in = 64u;
                    if (TW) { const uint32_t sl = lo_b / TW; s0 = lo_b + sl; k_in = TW - (lo_b - sl * TW); }
                    uint32_t w[6];
                    if (wide_ok && sp + s0 + 32 <= P.buf_end) {
                        uint4 v0, v1;
                        __builtin_memcpy(&v0, sp + s0, 16);
                        __builtin_memcpy(&v1, sp + s0 + 16, 16);
                        w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y;
                    } else {
#pragma unroll
                        for (int d = 0; d < 6; ++d) {
                            uint32_t z = 0;
                            for (int e = 0; e < 4; ++e)
                                if (s0 + (uint32_t)(4 * d + e) < text_len) z |= (uint32_t)sp[s0 + (uint32_t)(4 * d + e)] << (8 * e);
                            w[d] = z;
                        }
                    }
                    uint32_t c[5];  // the bases lo_b .. lo_b + 19, the source newline (window byte k_in) removed
#pragma unroll
                    for (int d = 0; d < 5; ++d) {
                        const int tt = (int)k_in - 4 * d;
                        const uint32_t keep = tt >= 4 ? 0xFFFFFFFFu : (tt <= 0 ? 0u : (1u << (8 * tt)) - 1u);
                        c[d] = (w[d] & keep) | (__builtin_amdgcn_alignbyte(w[d + 1], w[d], 1) & ~keep);
                    }
                    uint32_t B[4];
                    if (reverse) {
                        const uint32_t rv[5] = {__builtin_bswap32(c[3]), __builtin_bswap32(c[2]), __builtin_bswap32(c[1]), __builtin_bswap32(c[0]), 0u};
#pragma unroll
                        for (int d = 0; d < 4; ++d) B[d] = nbases == 16u ? rv[d] : __builtin_amdgcn_alignbyte(rv[d + 1], rv[d], 1);
                    } else {
#pragma unroll
                        for (int d = 0; d < 4; ++d) B[d] = c[d];
                    }
                    if (use_lut) {
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            B[d] = (uint32_t)s_lut[B[d] & 0xFFu] | ((uint32_t)s_lut[(B[d] >> 8) & 0xFFu] << 8) |
                                   ((uint32_t)s_lut[(B[d] >> 16) & 0xFFu] << 16) | ((uint32_t)s_lut[B[d] >> 24] << 24);
                    }
                    uint32_t O[4];
                    if (k_out < 16u) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const int tt = (int)k_out - 4 * d;
                            const uint32_t keep = tt >= 4 ? 0xFFFFFFFFu : (tt <= 0 ? 0u : (1u << (8 * tt)) - 1u);
                            const uint32_t up = __builtin_amdgcn_alignbyte(B[d], d ? B[d - 1] : 0u, 3);
                            uint32_t v = (B[d] & keep) | (up & ~keep);
                            if ((int)(k_out >> 2) == d) { const uint32_t sh = 8u * (k_out & 3u); v = (v & ~(0xFFu << sh)) | (0x0Au << sh); }
                            O[d] = v;
                        }
                    } else {
#pragma unroll
                        for (int d = 0; d < 4; ++d) O[d] = B[d];
                    }
                    const uint4 ov = make_uint4(O[0], O[1], O[2], O[3]);
                    __builtin_memcpy(d0 + x, &ov, 16);
                }
                if (gl == 0 && mine(a + W)) d0[W] = (uint8_t)'\n';
                return;
            }
        }
        // A wrapped FASTA source that is unwrapped, re-wrapped at another width, cut to a region or mapped (case, dna <-> rna,
        // complement, reversed): the output is made of pieces that are contiguous in the source AND in the output --
        // between two line ends of either.  A lane takes a source line, cuts it at the output line ends and copies every
        // piece 16 bytes at a time.  (One byte per lane and turn with a division each: 68 ms for 10 GB of 1 kb records
        // against 8.6 ms for the verbatim copy.)
        if (!LONG && TW != 0 && !P.fastq && (!use_lut || lut_in_lds)) {
            put_head();
            if (P.print_seq) {
                uint8_t* d0 = o + a;
                const uint32_t lw = (uint32_t)P.line_width;
                auto piece = [&](uint8_t* dst, const uint8_t* src, uint32_t nb) {
                    if (nb >= 16u) {
                        for (uint32_t x0 = 0; x0 < nb; x0 += 16u) {
                            const uint32_t x = x0 + 16u > nb ? nb - 16u : x0;
                            uint4 v;
                            __builtin_memcpy(&v, src + (reverse ? nb - 16u - x : x), 16);
                            if (reverse) v = make_uint4(__builtin_bswap32(v.w), __builtin_bswap32(v.z), __builtin_bswap32(v.y), __builtin_bswap32(v.x));
                            if (use_lut) {
                                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                                for (int d = 0; d < 4; ++d)
                                    w[d] = (uint32_t)s_lut[w[d] & 0xFFu] | ((uint32_t)s_lut[(w[d] >> 8) & 0xFFu] << 8) |
                                           ((uint32_t)s_lut[(w[d] >> 16) & 0xFFu] << 16) | ((uint32_t)s_lut[w[d] >> 24] << 24);
                                v = make_uint4(w[0], w[1], w[2], w[3]);
                            }
                            __builtin_memcpy(dst + x, &v, 16);
                        }
                    } else {
                        for (uint32_t k = 0; k < nb; ++k) { const uint8_t c = src[reverse ? nb - 1u - k : k]; dst[k] = use_lut ? s_lut[c] : c; }
                    }
                };
                if (L) {
                    const uint32_t s_first = sub_b / TW, s_last = (sub_e - 1u) / TW;
                    for (uint32_t sl = s_first + gl; sl <= s_last; sl += LANES) {
                        const uint32_t g0 = sl * TW > sub_b ? sl * TW : sub_b;                    // bases of this source line
                        const uint32_t g1 = (sl + 1u) * TW < sub_e ? (sl + 1u) * TW : sub_e;
                        const uint8_t* src = sp + g0 + sl;                                        // (g0 / TW == sl)
                        // output base indices of the line: ascending, or (reverse) the mirror image inside the region
                        uint32_t q = reverse ? sub_e - g1 : g0 - sub_b;
                        const uint32_t qe = reverse ? sub_e - g0 : g1 - sub_b;
                        while (q < qe) {
                            uint32_t pe = qe;
                            if (lw) { const uint32_t lend = (q / lw + 1u) * lw; if (lend < pe) pe = lend; }
                            // source bytes of output bases [q, pe): ascending from the line start, or (reverse) the bases
                            // sub_e - pe .. sub_e - q - 1 read backwards
                            const uint8_t* ps = reverse ? src + ((sub_e - pe) - g0) : src + (q - (g0 - sub_b));
                            piece(d0 + q + (lw ? q / lw : 0u), ps, pe - q);
                            if (lw && pe % lw == 0u && pe < L) d0[pe + pe / lw - 1u] = (uint8_t)'\n';  // the output line is full
                            q = pe;
                        }
                    }
                }
                if (gl == 0) d0[W] = (uint8_t)'\n';
            }
            return;
        }
        for (uint32_t x = clo + gl; x < chi; x += LANES) {
            uint8_t c;
            if (x < a) {
                const uint32_t m = P.print_seq ? 1u : 0u;
                if (x < m) c = (P.fastq && !P.fasta_out) ? '@' : '>';
                else if (x == a - 1) c = '\n';
                else c = HS.at(x - m);
            } else if (x < a + b) {
                uint32_t q = x - a;
                if (q == W) c = '\n';
                else {
                    bool nl = false;
                    if (w1) {
                        const uint32_t line = q / w1, col = q - line * w1;
                        if (col == w1 - 1) nl = true;
                        q = line * (w1 - 1) + col;
                    }
                    if (nl) c = '\n';
                    else {
                        const uint32_t bi = sub_b + (reverse ? L - 1 - q : q);
                        c = TW ? sp[bi + bi / TW] : sp[bi];
                        if (use_lut) c = lut[c];
                    }
                }
            } else {
                uint32_t q = x - a - b;
                if (!P.qual_only) {
                    if (q == 0) { o[x] = '+'; continue; }
                    if (q == 1) { o[x] = '\n'; continue; }
                    q -= 2;
                }
                c = q == L ? (uint8_t)'\n' : rqual[reverse ? L - 1 - q : q];
            }
            o[x] = c;
        }
        return;
    }
    // sequential path: gap removal and / or a multi-line FASTA source without a text view
    if (gl != 0 || clo != 0) return;  // (LONG: the first chunk's lane walks the whole record)
    uint32_t x = 0;
    if (P.print_name) {
        if (P.print_seq) o[x++] = (P.fastq && !P.fasta_out) ? '@' : '>';
        for (uint32_t k = 0; k < hl; ++k) o[x++] = HS.at(k);
        o[x++] = '\n';
    }
    const uint32_t R = r.region;
    if (P.print_seq) {
        uint32_t col = 0, base_i = 0;
        bool first = true;
        for (uint32_t k = 0; k < R; ++k) {
            const uint8_t c0 = r.seq[reverse ? R - 1 - k : k];
            if (c0 == '\n' && !P.fastq) continue;
            const uint32_t bi = reverse ? r.seq_len - 1 - base_i : base_i;  // index in the forward sequence
            ++base_i;
            if (bi < sub_b || bi >= sub_e) continue;
            if (P.remove_gaps && in_set(P.gap_set, c0)) continue;
            if (P.line_width > 0 && !first && col == (uint32_t)P.line_width) { o[x++] = '\n'; col = 0; }
            o[x++] = use_lut ? lut[c0] : c0;
            ++col;
            first = false;
        }
        o[x++] = '\n';
    }
    if (P.print_qual) {
        if (!P.qual_only) { o[x++] = '+'; o[x++] = '\n'; }
        const uint32_t L = r.seq_len;
        for (uint32_t k = 0; k < L; ++k) {
            const uint32_t j = reverse ? L - 1 - k : k;
            if (j < sub_b || j >= sub_e) continue;
            if (P.remove_gaps && in_set(P.gap_set, r.seq[j])) continue;
            o[x++] = r.qual[j];
        }
        o[x++] = '\n';
    }
}

__global__ __launch_bounds__(256) void k_count_nonzero(const uint32_t* __restrict__ v, uint64_t n, uint64_t* counter) {
    uint64_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        c += v[i] != 0;
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)c, d, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(c >> 32), d, 64);
        c += ((uint64_t)hi << 32) | lo;
    }
    // one atomic per block (an atomic on one address costs ~12 ns whoever issues it: 12 000 waves were 0.15 ms)
    __shared__ uint64_t s_c[4];
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t t = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        if (t) atomicAdd((unsigned long long*)counter, (unsigned long long)t);
    }
}

}  // namespace

hipError_t launch_seq_size(const uint8_t* buf, const RecordTable& t, const SeqParams& P, uint32_t* out_len,
                           uint64_t* status, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    const uint64_t blocks = (t.n + 255) / 256;
    hipLaunchKernelGGL(k_seq_size, dim3((unsigned)blocks), dim3(256), 0, st, buf, t, P, out_len, status);
    return hipGetLastError();
}

hipError_t launch_seq_emit(const uint8_t* buf, const RecordTable& t, const SeqParams& Pin, const uint32_t* out_len,
                           const uint64_t* out_off, uint8_t* out, hipStream_t st, uint64_t total_bytes, uint64_t records) {
    if (t.n == 0) return hipSuccess;
    SeqParams P = Pin;
    if (!(P.long_list && P.long_count)) P.long_thresh = 0u;
    // tiny records (names only) are written by the per-byte path: one pass of 16 lanes beats three of 4
    // (4 lanes up to 640 bytes per record: 1 kb records measured 12.1 ms with 4 lanes against 8.7 ms with 16 at 10 GB)
    const bool small = records > 0 && total_bytes / records < 640 && total_bytes / records >= 48;
    if (small) {
        const uint64_t blocks = (t.n * 4 + 255) / 256;
        hipLaunchKernelGGL((k_seq_emit<4, false>), dim3((unsigned)blocks), dim3(256), 0, st, buf, t, P, out_len, out_off, out);
    } else {
        const uint64_t blocks = (t.n * 16 + 255) / 256;
        hipLaunchKernelGGL((k_seq_emit<16, false>), dim3((unsigned)blocks), dim3(256), 0, st, buf, t, P, out_len, out_off, out);
    }
    if (P.long_thresh) {
        const unsigned chunks = (unsigned)((P.long_max + LONG_CH - 1) / LONG_CH);
        hipLaunchKernelGGL((k_seq_emit<16, true>), dim3(chunks, (unsigned)P.long_count), dim3(256), 0, st, buf, t, P, out_len,
                           out_off, out);
    }
    return hipGetLastError();
}

// indices of the records whose output is at least `thresh` bytes (any order), their number and the largest size
__global__ __launch_bounds__(256) void k_find_long(const uint32_t* __restrict__ out_len, uint64_t n, uint32_t thresh,
                                                   uint32_t* __restrict__ list, unsigned long long* __restrict__ count_max) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = out_len[i];
        if (v >= thresh) {  // rare by construction (a record of a MiB or more)
            list[atomicAdd(&count_max[0], 1ull)] = (uint32_t)i;
            atomicMax(&count_max[1], (unsigned long long)v);
        }
    }
}

hipError_t launch_find_long(const uint32_t* out_len, uint64_t n, uint32_t thresh, uint32_t* list, uint64_t* count_max,
                            hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_find_long, dim3((unsigned)blocks), dim3(256), 0, st, out_len, n, thresh, list,
                       (unsigned long long*)count_max);
    return hipGetLastError();
}

hipError_t launch_count_nonzero(const uint32_t* v, uint64_t n, uint64_t* counter, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_count_nonzero, dim3((unsigned)blocks), dim3(256), 0, st, v, n, counter);
    return hipGetLastError();
}

}  // namespace bsk
