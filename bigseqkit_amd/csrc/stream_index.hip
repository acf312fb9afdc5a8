// ============================================================================
// stream_index.hip -- the record table: IndexSink of the streaming skeleton.
//
// Replaces SeqParser.Read (/root/reference/bigseqkit-lib/helper.go:219-325) as a
// producer of per-record (head, seq, qual) slices: one streaming pass counts the
// records of every range, a tiny scan turns counts into bases, a second pass
// writes the SoA table at exact global positions (record order == file order).
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

#include "index.hpp"
#include "anchor.hpp"
#include "stream_core_dev.hpp"

namespace bsk {

namespace {

using namespace stream;

// FASTQ: the deferred window of the whole-record sink (stream_core_dev.hpp): its size, and whether the sink runs at the end of
// a tile (1) or when the window is full (0)
#ifndef BSK_INDEX_WINDOW
#define BSK_INDEX_WINDOW REC_WINDOW
#endif
#ifndef BSK_INDEX_TE
#define BSK_INDEX_TE 1
#endif

struct IndexSink {
    static constexpr bool TILE_HOOK = false;
    static constexpr bool RECORDS4 = true;  // FASTQ: whole records, 64 at a time (records() below)
    static constexpr bool REC_TILE_END = BSK_INDEX_TE != 0;
    IndexDev D;
    uint64_t base = 0;   // global index of the first record of the range (wave-uniform)
    uint64_t limit = 0;  // first index this range must not write (table capacity or end of its sparse slice)
    uint32_t nrec = 0;   // FASTA: records closed so far in this range (wave-uniform)
    uint32_t err = 0;
    // FASTA: the record that is open at the start of a batch
    uint64_t open_start = 0;  // absolute offset of its '>'
    uint32_t open_lhead = 0, open_key = 0;
    // FASTA line layout of the open record (text_dev.hpp): length of its first sequence line, any line seen so far that
    // breaks "all lines but the last are equally long, the last is 1..W"
    uint32_t open_w = 0;
    bool open_irr = false;
    // FASTA with line-start ranges (D.parts != null): a range may begin inside a record
    uint32_t range_id = 0, nhdr = 0;      // headers seen so far in this range == records it owns
    bool mid_start = false;               // the range begins inside a record
    bool open_is_header = false;          // open_* describe a header of THIS range
    uint32_t open_hdr_rank = 0;           // line index of that header
    uint32_t part_first = 0;              // length of the first line of a mid-record range
    bool any_event = false, last_closing = true;
    uint32_t last_key = 0, last_rank = 0, last_len = 0;
    uint64_t range_end = 0;

    __device__ __forceinline__ void begin_range(uint64_t b, uint64_t lim, uint64_t rs, uint64_t re = 0, uint32_t r = 0,
                                                bool mid = false) {
        base = b;
        limit = lim;
        nrec = 0;
        open_start = rs;
        open_lhead = 0;
        open_key = 0;
        open_w = 0;
        open_irr = false;
        range_id = r; nhdr = 0; mid_start = mid; open_is_header = false; open_hdr_rank = 0; part_first = 0;
        any_event = false; last_closing = true; last_key = 0; last_rank = 0; last_len = 0;
        range_end = re;
    }

    // FASTA: the parts this range leaves to k_index_stitch (lane 0)
    __device__ __forceinline__ void end_range() {
        if ((threadIdx.x & 63) != 0 || !D.parts || !D.write) return;
        RangePart& Q = D.parts[range_id];
        uint32_t f = IP_VISITED | (open_is_header ? IP_HAS_HEADER : 0u);
        if (any_event && !last_closing) {
            if (open_is_header) {
                // the last record of the range is still open: its table entry holds what is known so far
                const uint64_t g = base + nhdr - 1u;
                const uint32_t bases = last_key - open_key;
                if (g < limit) {
                    D.t.start[g] = open_start;
                    D.t.l_head[g] = open_lhead;
                    D.t.l_seq[g] = bases;
                    D.t.aux[g] = 0;
                    D.t.text_w[g] = 0;
                } else {
                    err |= ERR_CAPACITY;
                }
                Q.tail_bases = bases;
                Q.tail_nlines = last_rank - open_hdr_rank;
                Q.tail_first = open_w;
                Q.tail_last = last_len;
                f |= IP_TAIL_OPEN | (open_irr ? IP_TAIL_IRR : 0u);
            } else {
                // no header and no closing line: the whole range is inside one record
                Q.head_bases = last_key;
                Q.head_end_abs = range_end;
                Q.head_nlines = last_rank + 1u;
                Q.head_first = part_first;
                Q.head_last = last_len;
                f |= open_irr ? IP_HEAD_IRR : 0u;
            }
        }
        atomicOr(&Q.flags, f);
    }

    // FASTQ, whole records: lane j = record j of the window (stream_core_dev.hpp sink_records4) -- the rules of batch() for
    // its four events, then the table row; 64 consecutive rows per store instruction (batch() wrote the 13 of a tile)
    template <class LDS>
    __device__ __forceinline__ void records(LDS& L, uint32_t R, uint32_t wb, uint64_t tile_idx, uint32_t tile_rel, uint64_t rs,
                                            uint64_t re, const uint8_t* __restrict__ buf) {
        if (!D.write) return;  // count pass of the exact fallback: the number of lines is all it wants
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t end_rel = (uint32_t)(re - rs);
        auto next_of = [&](uint32_t v16, uint32_t p) -> uint32_t {
            if (p + 1u >= end_rel) return 0u;
            if (v16 & 0x100u) return v16 & 0xFFu;
            return buf[rs + p + 1u];
        };
        // the rows of this call begin at a wave-uniform place: scalar bases + a small lane index (no 64-bit row number per lane)
        const uint64_t g0 = base + (wb >> 2);
        const uint64_t room64 = limit > g0 ? limit - g0 : 0ull;
        const uint32_t room = room64 > 0xFFFFull ? 0xFFFFu : (uint32_t)room64;
        uint64_t* const t_start = D.t.start + g0;
        uint32_t* const t_lhead = D.t.l_head + g0;
        uint32_t* const t_lseq = D.t.l_seq + g0;
        uint32_t* const t_aux = D.t.aux + g0;
        for (uint32_t r0 = 0; r0 < R; r0 += WAVE) {
            const uint32_t j = (r0 + lane) & 0xFFFFu;  // (R <= window / 4)
            if (j >= R) continue;
            const uint32_t s = HISTORY + 4u * j;
            const uint32_t p0 = L.pos[s - 1u], ph = L.pos[s], pb = L.pos[s + 1u];
            if (next_of(L.nc[s], ph) == '+') err |= ERR_BAD_PLUS;
            if (next_of(L.nc[s + 1u], pb) != '+') err |= ERR_BAD_PLUS;
            const uint32_t pp = L.pos[s + 2u], pq = L.pos[s + 3u];
            const uint32_t ls = pb - ph - 1u;
            if (pq - pp - 1u != ls) err |= ERR_LEN_MISMATCH;
            if (pq + 1u < end_rel && next_of(L.nc[s + 3u], pq) != '@') err |= ERR_BAD_HEADER;
            if (j < room) {
                t_start[j] = rs + (uint64_t)(p0 + 1u);   // (p0 + 1 wraps to 0 at the range start)
                t_lhead[j] = ph - p0 - 1u;
                t_lseq[j] = ls;
                t_aux[j] = pp - pb - 1u;
            } else {
                err |= ERR_CAPACITY;
            }
        }
        (void)tile_idx; (void)tile_rel;
    }

    template <bool FASTQ, bool ALL, class LDS>
    __device__ __forceinline__ void batch(LDS& L, uint32_t E, uint32_t wb, uint64_t tile_idx,
                                          uint32_t tile_rel, uint64_t re, const uint8_t* __restrict__ buf) {
        const int lane = threadIdx.x & 63;
        for (uint32_t e0 = 0; e0 < E; e0 += WAVE) {
            const uint32_t e = e0 + lane;
            const bool on = e < E;
            const uint32_t s = HISTORY + (on ? e : 0);
            const uint32_t rank = wb + e;
            const uint32_t p = L.pos[s];
            const uint64_t abs_next = tile_idx + (uint64_t)(uint32_t)(p - tile_rel) + 1;  // byte after the newline
            if constexpr (FASTQ) {
                const uint32_t role = rank & 3u;
                if (on && D.write) {
                    // same structural validation as the stats kernel (strict 4-line FASTQ)
                    if (role == 1u) {
                        if (next_char(L, s, abs_next, re, buf) != '+') err |= ERR_BAD_PLUS;
                    } else if (role == 0u) {
                        if (next_char(L, s, abs_next, re, buf) == '+') err |= ERR_BAD_PLUS;
                    } else if (role == 3u) {
                        const uint32_t p1 = L.pos[s - 1], p2 = L.pos[s - 2], p3 = L.pos[s - 3], p4 = L.pos[s - 4];
                        const uint32_t lq = p - p1 - 1u, lp = p1 - p2 - 1u, ls = p2 - p3 - 1u, lh = p3 - p4 - 1u;
                        if (lq != ls) err |= ERR_LEN_MISMATCH;
                        if (abs_next < re && next_char(L, s, abs_next, re, buf) != '@') err |= ERR_BAD_HEADER;
                        const uint64_t g = base + (rank >> 2);
                        if (g < limit) {
                            D.t.start[g] = abs_of(p4, tile_idx, tile_rel) + 1;
                            D.t.l_head[g] = lh;
                            D.t.l_seq[g] = ls;
                            D.t.aux[g] = lp;
                        } else {
                            err |= ERR_CAPACITY;
                        }
                    }
                }
            } else {
                const bool closing = on && L.flag[s] != 0;
                // what a header-end event knows about its record (every lane computes it: the closing lane of a record
                // fetches it with a shuffle instead of walking back through the events)
                const uint32_t key = p - rank;
                const uint32_t my_lhead = p - L.pos[s - 1] - 1u;
                const uint64_t my_start = abs_of(L.pos[s - 1], tile_idx, tile_rel) + 1;
                // ---- line layout (replaces a separate pass over every line end): an event is a header end (H), the
                // first sequence line of its record (F), or a later sequence line, which must be as long as the line
                // before it (the last one: 1..that length)
                const bool H = on && L.flag[s - 1] != 0;
                const bool F = on && !H && L.flag[s - 2] != 0;
                const uint32_t len = p - L.pos[s - 1] - 1u;
                const bool cont_first = mid_start && rank == 0u;  // first line of a range that begins inside a record:
                bool viol = false;                                // its predecessor is in another range (k_index_stitch)
                if (on && !H && !cont_first) {
                    if (F) viol = !closing && len == 0u;
                    else {
                        const uint32_t plen = L.pos[s - 1] - L.pos[s - 2] - 1u;
                        viol = closing ? (len > plen || len == 0u) : (len != plen);
                    }
                }
                const uint64_t hb = __ballot(H), fb = __ballot(F), vb = __ballot(viol);
                if (mid_start && wb == 0u && e0 == 0u && E > 0u) part_first = (uint32_t)__builtin_amdgcn_readlane((int)len, 0);
                const uint64_t upto = (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
                // records owned by this range are numbered by their headers; a closing line before the first header
                // of the range ends a record that began in an earlier range
                const uint32_t hdrs = nhdr + (uint32_t)__popcll(hb & upto);
                const bool whole = hdrs > 0u;
                uint32_t tw = 0;  // text_w of the record this lane closes
                bool irr_here = false;
                {
                    // every lane takes part in the shuffle (a ds_bpermute reads nothing from inactive lanes)
                    const uint64_t hh = hb & upto, ff = fb & upto;
                    bool irr;
                    int src = lane;
                    if (hh) {
                        const int hpos = 63 - __clzll((long long)hh);
                        const uint64_t seg = upto & ~((hpos == 63) ? ~0ull : ((2ull << hpos) - 1ull));
                        irr = (vb & seg) != 0;
                        src = (hpos + 1) & 63;  // the event after a header end is the first sequence line
                    } else {
                        irr = open_irr || (vb & upto) != 0;
                        if (ff) src = (int)__ffsll((long long)ff) - 1;
                    }
                    const uint32_t wsh = (uint32_t)__shfl((int)len, src, 64);
                    const uint32_t W = (hh || ff) ? wsh : open_w;
                    irr_here = irr;
                    if (closing && !H && !F) tw = (irr || W < 16u) ? 0xFFFFFFFFu : W;
                }
                // header-end event of the record a lane closes: the last H at or below it in these 64 events
                const int hsrc = (hb & upto) ? 63 - __clzll((long long)(hb & upto)) : lane;
                const uint32_t key_h = (uint32_t)__shfl((int)key, hsrc, 64);
                const uint32_t lhead_h = (uint32_t)__shfl((int)my_lhead, hsrc, 64);
                const uint32_t start_lo = (uint32_t)__shfl((int)(uint32_t)my_start, hsrc, 64);
                const uint32_t start_hi = (uint32_t)__shfl((int)(uint32_t)(my_start >> 32), hsrc, 64);
                if (closing && D.write) {
                    uint32_t key_i, lhead;
                    uint64_t start;
                    if (hb & upto) {
                        key_i = key_h;
                        lhead = lhead_h;
                        start = ((uint64_t)start_hi << 32) | start_lo;
                    } else {
                        key_i = open_key;
                        lhead = open_lhead;
                        start = open_start;
                    }
                    const uint32_t seqlen = (p - rank) - key_i;
                    const uint64_t rec_end = abs_next > re ? re : abs_next;  // the virtual event sits at re
                    if (whole) {
                        const uint64_t region = rec_end - (start + lhead + 1 < rec_end ? start + lhead + 1 : rec_end);
                        if (region > 0xFFFFFFFFull) err |= ERR_LINE_TOO_LONG;
                        const uint64_t g = base + (hdrs - 1u);
                        if (g < limit) {
                            D.t.start[g] = start;
                            D.t.l_head[g] = lhead;
                            D.t.l_seq[g] = seqlen;
                            D.t.aux[g] = (uint32_t)region;
                            D.t.text_w[g] = tw;
                        } else {
                            err |= ERR_CAPACITY;
                        }
                    } else if (D.parts) {
                        // end of the record that was open when the range began
                        RangePart& Q = D.parts[range_id];
                        Q.head_bases = seqlen;  // open_key == 0: bases since the range start
                        Q.head_end_abs = rec_end;
                        Q.head_nlines = rank + 1u;
                        Q.head_first = rank == 0u ? len : part_first;
                        Q.head_last = len;
                        atomicOr(&Q.flags, IP_HEAD_CLOSED | (irr_here ? IP_HEAD_IRR : 0u));
                    }
                }
                nrec = nhdr + (uint32_t)__popcll(hb);  // records owned so far (k_index reads it after the range)
                nhdr = nrec;
                // state of the record that stays open after these 64 events (wave-uniform)
                if (hb) {
                    const int hl = 63 - __clzll((long long)hb);
                    const uint64_t above = hl == 63 ? 0ull : ~((2ull << hl) - 1ull);
                    open_irr = (vb & above) != 0;
                    open_w = (fb & above) ? (uint32_t)__builtin_amdgcn_readlane((int)len, hl + 1) : 0u;
                    open_key = (uint32_t)__builtin_amdgcn_readlane((int)key, hl);
                    open_lhead = (uint32_t)__builtin_amdgcn_readlane((int)my_lhead, hl);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_start, hl);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_start >> 32), hl);
                    open_start = ((uint64_t)hi << 32) | lo;
                    open_is_header = true;
                    open_hdr_rank = wb + e0 + (uint32_t)hl;
                } else {
                    open_irr = open_irr || vb != 0;
                    if (fb) open_w = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)__ffsll((long long)fb) - 1);
                }
            }
        }
        if constexpr (!FASTQ) {
            if (E > 0) {  // the last event of the batch (uniform LDS reads)
                const uint32_t sl = HISTORY + (E - 1u);
                last_rank = wb + (E - 1u);
                last_key = L.pos[sl] - last_rank;
                last_len = L.pos[sl] - L.pos[sl - 1] - 1u;
                last_closing = L.flag[sl] != 0;
                any_event = true;
            }
        }
    }
};

// 7 waves per SIMD (72 VGPRs instead of the compiler's 74-75): k_index 3.20 -> 3.06 ms on 12.5 GB FASTQ, 12.7 -> 12.3 ms
// on 50 GB FASTA (scripts/variant_index_waves.sh)
#ifndef BSK_INDEX_WAVES
#define BSK_INDEX_WAVES 7
#endif
#if BSK_INDEX_WAVES
#define BSK_INDEX_ATTR __attribute__((amdgpu_waves_per_eu(BSK_INDEX_WAVES, 8)))
#else
#define BSK_INDEX_ATTR
#endif

template <bool FASTQ, bool DPP>
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) BSK_INDEX_ATTR void k_index(const uint8_t* __restrict__ buf, uint64_t n,
                                                                   const uint64_t* __restrict__ anchors,
                                                                   uint32_t nranges, uint32_t* __restrict__ queue,
                                                                   IndexDev D, uint64_t chunk) {
    constexpr int WINDOW = FASTQ ? BSK_INDEX_WINDOW : CAP;  // FASTQ: 64 whole records per sink call (IndexSink::records)
    __shared__ Lds<FASTQ, false, WINDOW> s_l[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Lds<FASTQ, false, WINDOW>& L = s_l[wave];
    IndexSink sink;
    sink.D = D;
    PredConsts P;  // unused (ALL == false)
    P.k20 = P.k30 = 0;
    P.ngap = 0;
    const uint64_t n_eff = anchors[nranges];
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(queue, 1u);
        r = wave_first(r);
        if (r >= nranges) break;
        uint64_t rs = anchors[r], re = anchors[r + 1];
        rs = rs < n_eff ? rs : n_eff;
        re = re < n_eff ? re : n_eff;
        if (rs >= re) {
            if (D.write != 1 && lane == 0) D.range_count[r] = 0;
            continue;
        }
        uint64_t b = 0, lim = D.t.cap;
        if (D.write == 1) b = D.range_base[r];
        else if (D.write == 2) { b = (uint64_t)r * D.sparse_cap; lim = b + D.sparse_cap; if (lim > D.t.cap) lim = D.t.cap; }
        sink.begin_range(b, lim, rs, re, r, !FASTQ && buf[rs] != '>');
        // FASTA: the newline-free middle of a line longer than the nominal chunk is not read (stream_core_dev.hpp)
        const uint64_t skip_from = (!FASTQ && chunk) ? (uint64_t)(r + 1u) * chunk : ~0ull;
        const uint32_t lines = stream_range<FASTQ, false, DPP>(L, buf, n, rs, re, re == n_eff, P, sink, skip_from);
        if constexpr (!FASTQ) sink.end_range();
        if (D.write != 1 && lane == 0) D.range_count[r] = FASTQ ? (uint64_t)(lines >> 2) : (uint64_t)sink.nrec;
    }
    const uint32_t err = wave_or_u32(sink.err);
    if (lane == 0 && err) atomicOr((unsigned long long*)&D.status[0], (unsigned long long)err);
}

// exclusive scan of up to a few 10^6 u64 values with one block: 8 consecutive values per thread and round (the block
// sums of a 3 x 10^8 record scan are 154 000 values: 75 rounds instead of 600)
__global__ __launch_bounds__(256) void k_scan_small(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n,
                                                    uint64_t* __restrict__ total_at) {
    constexpr uint32_t PER = 8;
    __shared__ uint64_t s_w[4];
    __shared__ uint64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 256 * PER) {
        const uint32_t i = i0 + threadIdx.x * PER;
        uint64_t v[PER];
        uint64_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            v[k] = i + k < n ? in[i + k] : 0;
            mine += v[k];
        }
        // inclusive scan of the thread sums inside the wave
        uint64_t x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64);
            const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
            if (lane >= d) x += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint64_t off = s_carry;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        uint64_t run = off + x - mine;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            if (i + k < n) out[i + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[n] = s_carry;
        if (total_at) *total_at = s_carry;  // (a second place for the total: the end of a two-level scan, the control block)
    }
}

// TWO such scans in one launch (the bytes and the record counts of the ranges of a streaming pass: k_names, k_subseq_stream),
// a block of 1 024 threads each: 2 x 0.12 ms -> 0.03 ms for the 2 x 10^5 ranges of a 100 GB shard
__global__ __launch_bounds__(1024) void k_scan_small2(const uint64_t* __restrict__ in0, uint64_t* __restrict__ out0, uint64_t* __restrict__ total0,
                                                      const uint64_t* __restrict__ in1, uint64_t* __restrict__ out1, uint64_t* __restrict__ total1,
                                                      uint32_t n) {
    constexpr uint32_t PER = 8, NT = 1024, NW = NT / 64;
    const uint64_t* in = blockIdx.x ? in1 : in0;
    uint64_t* out = blockIdx.x ? out1 : out0;
    uint64_t* total_at = blockIdx.x ? total1 : total0;
    __shared__ uint64_t s_w[NW];
    __shared__ uint64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += NT * PER) {
        const uint32_t i = i0 + threadIdx.x * PER;
        uint64_t v[PER];
        uint64_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            v[k] = i + k < n ? in[i + k] : 0;
            mine += v[k];
        }
        uint64_t x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64);
            const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
            if (lane >= d) x += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint64_t off = s_carry;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        uint64_t run = off + x - mine;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            if (i + k < n) out[i + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == NT - 1) s_carry = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[n] = s_carry;
        if (total_at) *total_at = s_carry;
    }
}

// ---- large exclusive scan u32 -> u64: reduce / scan of block sums / downsweep ----------
constexpr int SCAN_ITEMS = 8;                  // per thread
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;   // 2048 per block

__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ sums) {
    const uint64_t b0 = (uint64_t)blockIdx.x * SCAN_BLOCK;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint64_t i = b0 + (uint64_t)k * 256 + threadIdx.x;
        if (i < n) acc += in[i];
    }
    acc = wave_sum_u64(acc);
    __shared__ uint64_t s_w[4];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void k_scan_down(const uint32_t* __restrict__ in, uint64_t n,
                                                   const uint64_t* __restrict__ block_off, uint64_t* __restrict__ out) {
    // thread t owns items [b0 + t*8, +8): scan its 8 items, then scan thread totals
    __shared__ uint64_t s_w[4];
    const uint64_t b0 = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint64_t tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = b0 + k < n ? in[b0 + k] : 0u;
        tot += v[k];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t x = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64);
        const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
        if (lane >= d) x += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint64_t off = block_off[blockIdx.x] + x - tot;
    for (int w = 0; w < wave; ++w) off += s_w[w];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (b0 + k < n) out[b0 + k] = off;
        off += v[k];
    }
}

__global__ void k_set_total(const uint64_t* __restrict__ block_off, uint64_t nblocks, uint64_t* __restrict__ out, uint64_t n) {
    out[n] = block_off[nblocks];
}

__global__ void k_reset_queue(uint32_t* q) { *q = 0; }

// ---- the same scan with the summary of finish_sizes folded in (round 4): three launches instead of seven (reduce, block
// scan, down sweep, total, memset, count_nonzero, find_long), and every scalar the host wants lands in `fin` (the control
// block of the context: ONE copy brings total, kept, the long records' count / largest size and the status word)
__global__ __launch_bounds__(256) void k_scan_reduce_fin(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ sums,
                                                         uint64_t* __restrict__ cnts) {
    const uint64_t b0 = (uint64_t)blockIdx.x * SCAN_BLOCK;
    uint64_t acc = 0, nz = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint64_t i = b0 + (uint64_t)k * 256 + threadIdx.x;
        if (i < n) { const uint32_t v = in[i]; acc += v; nz += v != 0u; }
    }
    acc = wave_sum_u64(acc);
    nz = wave_sum_u64(nz);
    __shared__ uint64_t s_w[4], s_z[4];
    if ((threadIdx.x & 63) == 0) { s_w[threadIdx.x >> 6] = acc; s_z[threadIdx.x >> 6] = nz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        cnts[blockIdx.x] = s_z[0] + s_z[1] + s_z[2] + s_z[3];
    }
}

// one block: exclusive scan of the block sums (k_scan_small's loop), the sum of the non-zero counts, out[n] and the
// summary words; zeroes the long-record words that the down sweep behind it fills
__global__ __launch_bounds__(256) void k_scan_small_fin(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t nb,
                                                        const uint64_t* __restrict__ cnts, uint64_t* __restrict__ total_at,
                                                        uint64_t* __restrict__ fin) {
    constexpr uint32_t PER = 8;
    __shared__ uint64_t s_w[4];
    __shared__ uint64_t s_carry;
    __shared__ uint64_t s_z[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    uint64_t nz = 0;
    for (uint32_t i0 = 0; i0 < nb; i0 += 256 * PER) {
        const uint32_t i = i0 + threadIdx.x * PER;
        uint64_t v[PER];
        uint64_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            v[k] = i + k < nb ? in[i + k] : 0;
            mine += v[k];
            if (i + k < nb) nz += cnts[i + k];
        }
        uint64_t x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64);
            const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
            if (lane >= d) x += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint64_t off = s_carry;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        uint64_t run = off + x - mine;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            if (i + k < nb) out[i + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = off + x;
        __syncthreads();
    }
    nz = wave_sum_u64(nz);
    if (lane == 0) s_z[wave] = nz;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[nb] = s_carry;
        *total_at = s_carry;
        fin[0] = s_carry;                             // FIN_TOTAL
        fin[1] = s_z[0] + s_z[1] + s_z[2] + s_z[3];   // FIN_KEPT
        fin[2] = 0;                                   // FIN_LONG_COUNT   (k_scan_down_fin)
        fin[3] = 0;                                   // FIN_LONG_MAX
    }
}

__global__ __launch_bounds__(256) void k_scan_down_fin(const uint32_t* __restrict__ in, uint64_t n,
                                                       const uint64_t* __restrict__ block_off, uint64_t* __restrict__ out,
                                                       uint32_t thresh, uint32_t* __restrict__ long_list,
                                                       unsigned long long* __restrict__ fin) {
    __shared__ uint64_t s_w[4];
    const uint64_t b0 = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint64_t tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = b0 + k < n ? in[b0 + k] : 0u;
        tot += v[k];
        if (v[k] >= thresh && long_list) {  // rare by construction (a record of a MiB or more)
            long_list[atomicAdd(&fin[2], 1ull)] = (uint32_t)(b0 + k);
            atomicMax(&fin[3], (unsigned long long)v[k]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t x = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64);
        const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
        if (lane >= d) x += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint64_t off = block_off[blockIdx.x] + x - tot;
    for (int w = 0; w < wave; ++w) off += s_w[w];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (b0 + k < n) out[b0 + k] = off;
        off += v[k];
    }
}

__global__ void k_fin_zero(uint64_t* __restrict__ out0, uint64_t* __restrict__ fin) {
    out0[0] = 0;
    fin[0] = 0; fin[1] = 0; fin[2] = 0; fin[3] = 0;
}

// FASTA, line-start ranges: records that span ranges.  The thread of range r finishes the record that is open at the
// end of r: bases, region and line layout are completed from the head parts of the following ranges, up to the range
// that closes it.  Chains are disjoint (see k_stats_stitch).
__global__ __launch_bounds__(256) void k_index_stitch(RecordTable t, const RangePart* __restrict__ parts,
                                                      const uint64_t* __restrict__ range_count,
                                                      const uint64_t* __restrict__ range_base, uint32_t nranges,
                                                      uint64_t* __restrict__ status) {
    const uint32_t r0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (r0 >= nranges) return;
    const uint32_t f0 = parts[r0].flags;
    if ((f0 & (IP_VISITED | IP_TAIL_OPEN)) != (IP_VISITED | IP_TAIL_OPEN)) return;
    const uint64_t g = range_base[r0] + range_count[r0] - 1;
    uint64_t bases = parts[r0].tail_bases;
    uint32_t nl = parts[r0].tail_nlines, W = parts[r0].tail_first, last = parts[r0].tail_last;
    bool irr = (f0 & IP_TAIL_IRR) != 0;
    for (uint32_t r = r0 + 1; r < nranges; ++r) {
        const RangePart& Q = parts[r];
        const uint32_t f = Q.flags;
        if (!(f & IP_VISITED)) continue;
        if (!(f & IP_HAS_HEADER) || (f & IP_HEAD_CLOSED)) {  // this range holds a part of the open record
            const bool closed = (f & IP_HEAD_CLOSED) != 0;
            if (Q.head_nlines > 0u) {
                if (nl == 0u) W = Q.head_first;          // the record's first sequence line
                else {
                    // the line that ended the previous part is no longer the last one: the next line is compared with it
                    const bool only_line_is_last = closed && Q.head_nlines == 1u;
                    if (only_line_is_last) irr |= !(Q.head_first >= 1u && Q.head_first <= last);
                    else irr |= Q.head_first != last;
                }
                nl += Q.head_nlines;
                last = Q.head_last;
                irr |= (f & IP_HEAD_IRR) != 0;
            }
            bases += Q.head_bases;
            if (closed) {
                const uint64_t seq0 = t.start[g] + t.l_head[g] + 1;
                const uint64_t region = Q.head_end_abs > seq0 ? Q.head_end_abs - seq0 : 0;
                // the record table keeps sequence and region lengths in 32 bits: a record of 2^32 bytes or more is
                // refused (ERR_LINE_TOO_LONG -> BSK_ERR_UNSUPPORTED), never truncated
                if (bases > 0xFFFFFFFFull || region > 0xFFFFFFFFull) atomicOr((unsigned long long*)&status[0], (unsigned long long)ERR_LINE_TOO_LONG);
                t.l_seq[g] = (uint32_t)bases;
                t.aux[g] = (uint32_t)region;
                t.text_w[g] = nl <= 1u ? 0u : ((irr || W < 16u) ? 0xFFFFFFFFu : W);
                return;
            }
        }
        if (f & IP_TAIL_OPEN) return;  // another record is open from here on
    }
}

// one block per range: copy its slice of the sparse table to its dense position
__global__ __launch_bounds__(256) void k_index_compact(RecordTable sp, uint64_t sparse_cap,
                                                       const uint64_t* __restrict__ range_count,
                                                       const uint64_t* __restrict__ range_base, RecordTable dn) {
    const uint32_t r = blockIdx.x;
    const uint64_t cnt = range_count[r], src = (uint64_t)r * sparse_cap, dst = range_base[r];
    for (uint64_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        dn.start[dst + i] = sp.start[src + i];
        dn.l_head[dst + i] = sp.l_head[src + i];
        dn.l_seq[dst + i] = sp.l_seq[src + i];
        dn.aux[dst + i] = sp.aux[src + i];
        dn.text_w[dst + i] = sp.text_w[src + i];
    }
}

}  // namespace

hipError_t launch_index(bool fastq, bool dpp, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors,
                        uint32_t nranges, uint32_t* queue, const IndexDev& D, hipStream_t st, uint64_t skip_chunk) {
    const dim3 b(WAVES_PER_BLOCK * WAVE);
    if (fastq) {
        if (dpp) hipLaunchKernelGGL((k_index<true, true>), dim3(blocks), b, 0, st, buf, n, anchors, nranges, queue, D, 0ull);
        else hipLaunchKernelGGL((k_index<true, false>), dim3(blocks), b, 0, st, buf, n, anchors, nranges, queue, D, 0ull);
    } else {
        if (dpp) hipLaunchKernelGGL((k_index<false, true>), dim3(blocks), b, 0, st, buf, n, anchors, nranges, queue, D, skip_chunk);
        else hipLaunchKernelGGL((k_index<false, false>), dim3(blocks), b, 0, st, buf, n, anchors, nranges, queue, D, skip_chunk);
    }
    return hipGetLastError();
}

int index_max_blocks_per_cu(bool fastq, bool dpp) {
    int nb = 0;
    const void* f = fastq ? (dpp ? (const void*)k_index<true, true> : (const void*)k_index<true, false>)
                          : (dpp ? (const void*)k_index<false, true> : (const void*)k_index<false, false>);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, WAVES_PER_BLOCK * WAVE, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

hipError_t launch_scan_small(const uint64_t* in, uint64_t* out, uint32_t n, hipStream_t st, uint64_t* total_at) {
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(256), 0, st, in, out, n, total_at);
    return hipGetLastError();
}

hipError_t launch_scan_small2(const uint64_t* in0, uint64_t* out0, uint64_t* total0, const uint64_t* in1, uint64_t* out1, uint64_t* total1,
                              uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_scan_small2, dim3(2), dim3(1024), 0, st, in0, out0, total0, in1, out1, total1, n);
    return hipGetLastError();
}

hipError_t launch_scan_u32(const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, hipStream_t st) {
    if (n == 0) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(256), 0, st, (const uint64_t*)nullptr, out, 0u, (uint64_t*)nullptr);
        return hipGetLastError();
    }
    const uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    // tmp[0..nb): block sums, then scanned in place into tmp2 = tmp + nb + 1 ... keep simple: two regions
    uint64_t* sums = tmp;
    uint64_t* offs = tmp + nb;  // [nb + 1]
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(256), 0, st, in, n, sums);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(256), 0, st, (const uint64_t*)sums, offs, (uint32_t)nb, out + n);  // (+ the total)
    hipLaunchKernelGGL(k_scan_down, dim3((unsigned)nb), dim3(256), 0, st, in, n, (const uint64_t*)offs, out);
    return hipGetLastError();
}

// exclusive scan of the output sizes + the summary of the size pass: fin[0] = total bytes, fin[1] = records with output,
// fin[2] / fin[3] = number / largest size of the records with >= thresh bytes (listed in long_list, any order).
// tmp: 3 * ceil(n / 2048) + 2 words.
hipError_t launch_scan_u32_fin(const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, uint32_t thresh, uint32_t* long_list,
                               uint64_t* fin, hipStream_t st) {
    if (n == 0) {
        hipLaunchKernelGGL(k_fin_zero, dim3(1), dim3(1), 0, st, out, fin);
        return hipGetLastError();
    }
    const uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    uint64_t* sums = tmp;
    uint64_t* offs = tmp + nb;          // [nb + 1]
    uint64_t* cnts = tmp + 2 * nb + 1;  // [nb]
    hipLaunchKernelGGL(k_scan_reduce_fin, dim3((unsigned)nb), dim3(256), 0, st, in, n, sums, cnts);
    hipLaunchKernelGGL(k_scan_small_fin, dim3(1), dim3(256), 0, st, (const uint64_t*)sums, offs, (uint32_t)nb, (const uint64_t*)cnts,
                       out + n, fin);
    hipLaunchKernelGGL(k_scan_down_fin, dim3((unsigned)nb), dim3(256), 0, st, in, n, (const uint64_t*)offs, out, thresh, long_list,
                       (unsigned long long*)fin);
    return hipGetLastError();
}

hipError_t launch_index_stitch(const RecordTable& dense, const RangePart* parts, const uint64_t* range_count,
                               const uint64_t* range_base, uint32_t nranges, uint64_t* status, hipStream_t st) {
    hipLaunchKernelGGL(k_index_stitch, dim3((nranges + 255u) / 256u), dim3(256), 0, st, dense, parts, range_count, range_base, nranges, status);
    return hipGetLastError();
}

hipError_t launch_index_compact(const RecordTable& sparse, uint64_t sparse_cap, const uint64_t* range_count,
                                const uint64_t* range_base, uint32_t nranges, const RecordTable& dense, hipStream_t st) {
    hipLaunchKernelGGL(k_index_compact, dim3(nranges), dim3(256), 0, st, sparse, sparse_cap, range_count, range_base, dense);
    return hipGetLastError();
}

hipError_t launch_reset_queue(uint32_t* queue, hipStream_t st) {
    hipLaunchKernelGGL(k_reset_queue, dim3(1), dim3(1), 0, st, queue);
    return hipGetLastError();
}

}  // namespace bsk
