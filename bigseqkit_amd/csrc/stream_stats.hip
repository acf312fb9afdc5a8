// ============================================================================
// stream_stats.hip -- `stats` on gfx950: the StatsSink of the streaming skeleton
// (stream_core_dev.hpp) + k_prep (range anchors) + launchers.
//
// Replaces, for one shard resident in HBM, the reference's
//   ReadFixer.Call   /root/reference/bigseqkit-lib/helper.go:41-66   (k_prep)
//   SeqParser.Read   bigseqkit-lib/helper.go:219-325   (line structure, stream_core_dev.hpp)
//   Stats.Call       bigseqkit-lib/stats.go:48-117     (StatsSink)
// One pass, every byte read once; lengths go to an LDS histogram (bins < 2048)
// flushed once per block; Q20/Q30/gap are differences of running counters taken
// at newline events.  See DESIGN.md section 3.
// ============================================================================
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "anchor.hpp"
#include "anchor_wave_dev.hpp"
#include "index.hpp"
#include "stream_core_dev.hpp"
#include "stream_fasta2_dev.hpp"
#include "stream_stats.hpp"

namespace bsk {

namespace {

using namespace stream;

// Lengths >= LDS_HIST (long reads, genes, chromosomes): a small per-block cache of (length, count) pairs behind the
// LDS histogram, flushed with it.  Without it a file of equally long records (50 GB of 5 kb CDS) sends every record
// to ONE global counter: 9.8 M atomics on one address = 119 ms, against 20 ms for the data pass.
constexpr int BIG_SLOTS = 64;
constexpr uint32_t BIG_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ void add_big(uint32_t len, uint32_t c, uint32_t* s_hist, const StatsDev& D) {
    uint32_t* keys = s_hist + LDS_HIST;
    uint32_t* cnts = keys + BIG_SLOTS;
    const uint32_t slot = (len * 2654435761u) >> 26;
    const uint32_t old = atomicCAS(&keys[slot], BIG_EMPTY, len);
    if (old == BIG_EMPTY || old == len) atomicAdd(&cnts[slot], c);
    else atomicAdd((unsigned long long*)&D.vec[STATS_HDR + len], (unsigned long long)c);  // slot taken by another length
}

// length histogram update with wave-level aggregation of the common case
// "every active lane saw the same length" (fixed-length reads)
__device__ __forceinline__ void add_length(bool has, uint32_t len, uint32_t* s_hist, const StatsDev& D) {
    const uint64_t act = __ballot(has);
    if (act == 0) return;
    const int leader = __ffsll((long long)act) - 1;
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)len, leader);
    const uint64_t same = __ballot(has && len == first);
    if (same == act) {
        if ((int)(threadIdx.x & 63) == leader) {
            const uint32_t c = (uint32_t)__popcll(act);
            if (first < (uint32_t)LDS_HIST) atomicAdd(&s_hist[first], c);
            else if (first < D.hist_cap) add_big(first, c, s_hist, D);
            else {
                atomicAdd((unsigned long long*)&D.vec[5], (unsigned long long)c);  // travels with the vector (all-reduce)
                for (uint32_t k = 0; k < c; ++k) {
                    unsigned long long i = atomicAdd((unsigned long long*)&D.status[1], 1ull);
                    if (i < D.overflow_cap) D.overflow[i] = first;
                }
            }
        }
        return;
    }
    if (has) {
        if (len < (uint32_t)LDS_HIST) atomicAdd(&s_hist[len], 1u);
        else if (len < D.hist_cap) add_big(len, 1u, s_hist, D);
        else {
            atomicAdd((unsigned long long*)&D.vec[5], 1ull);
            unsigned long long i = atomicAdd((unsigned long long*)&D.status[1], 1ull);
            if (i < D.overflow_cap) D.overflow[i] = len;
        }
    }
}


// ---------------------------------------------------------------------------
// StatsSink: Stats.Call (bigseqkit-lib/stats.go:48-117) on newline events
// ---------------------------------------------------------------------------
// ROLES (`stats -a` on FASTQ): Q20 / Q30 / gap are counted by the skeleton per line role (sink_role_counts), the events
// carry no running counters and the sink works as for the default row.
template <bool ROLES>
struct StatsSinkT {
    static constexpr bool TILE_HOOK = false;
    static constexpr bool ROLE_COUNTS = ROLES;
    static constexpr bool RECORDS4 = true;  // FASTQ on the sparse path: whole records, 64 at a time (records() below)
    // both rows run their sink when the window of 256 is full (stream_core_dev.hpp): the default row has LDS and registers
    // for nothing else at 7 waves per SIMD, and `-a` measured 21.0 ms that way against 22.6 at the end of tiles with 512
    static constexpr bool REC_TILE_END = false;
    static constexpr bool TILE_NT = !ROLES;  // the default row reads every byte once (non-temporal tile loads: stream_core_dev.hpp)
    uint32_t* s_hist;
    StatsDev D;
    // per-lane accumulators, reduced once per wave at kernel end
    uint64_t q20 = 0, q30 = 0, gap = 0, nrec = 0, sumlen = 0;
    // ROLES: the skeleton counts into these per lane and range (a lane sees 1/64 of a range: no overflow below 256 GB);
    // k_stats folds them into wave totals after every range
    uint32_t rq20 = 0, rq30 = 0, rgap = 0;
    uint32_t err = 0;
    // FASTA: header-end of the record that is open at the start of a batch
    uint32_t open_key = 0, open_sg = 0;
    // FASTA: ranges begin on line starts, possibly inside a record (see StatsDev::r_head)
    uint32_t range_id = 0;
    bool open_is_header = false;  // open_key belongs to a header of THIS range (else: to the range start)
    bool any_event = false, last_closing = true;
    bool last_line_hdr = false;   // FASTA: the line the last event ended is a header line
    uint32_t last_key = 0, last_a = 0;

    __device__ __forceinline__ void begin_range(uint32_t r) {
        open_key = 0; open_sg = 0;
        range_id = r;
        open_is_header = false;
        any_event = false; last_closing = true; last_line_hdr = false;
        last_key = 0; last_a = 0;
        last_pub_pos = 0xFFFFFFFFu;
    }

    // FASTA: what this range leaves open (lane 0 records it for k_stats_stitch)
    template <bool ALL>
    __device__ __forceinline__ void end_range() {
        if ((threadIdx.x & 63) != 0) return;
        uint32_t f = RF_VISITED | (open_is_header ? RF_HAS_HEADER : 0u);
        if (any_event && !last_closing) {
            const uint64_t bases = (uint64_t)(uint32_t)(last_key - (open_is_header ? open_key : 0u));
            if (open_is_header) { D.r_tail[range_id] = bases; f |= RF_TAIL_OPEN; }
            else D.r_head[range_id] = bases;  // no header and no closing line: the whole range is inside one record
            if constexpr (ALL) { if (!f2) gap += (uint32_t)(last_a - (open_is_header ? open_sg : 0u)); }  // (f2: the gap count is a sum over bytes, not over records)
        }
        if constexpr (ALL) {
            if (any_event && !last_line_hdr) f |= RF_SKIP_SEQ;  // (the RF_MID ranges that follow, if any, are sequence bytes)
        }
        atomicOr(&D.r_flags[range_id], f);
    }

    // Stats.Call on WHOLE FASTQ records, lane j = record j of the window (stream_core_dev.hpp sink_records4): the four line
    // ends of the record in one 16-byte LDS read, the rules of batch() below for its four events at once -- line 2 begins
    // with '+', the bases do not, as many qualities as bases, a '@' behind the record -- and one histogram update per 64
    // records (fixed-length reads: one LDS atomic)
    template <class LDS>
    __device__ __forceinline__ void records(LDS& L, uint32_t R, uint32_t wb, uint64_t tile_idx, uint32_t tile_rel, uint64_t rs,
                                            uint64_t re, const uint8_t* __restrict__ buf) {
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t end_rel = (uint32_t)(re - rs);
        // the byte behind the newline at relative position p: 0 past the range, the byte the emitting lane kept, else memory
        auto next_of = [&](uint32_t v16, uint32_t p) -> uint32_t {
            if (p + 1u >= end_rel) return 0u;
            if (v16 & 0x100u) return v16 & 0xFFu;
            return buf[rs + p + 1u];
        };
        for (uint32_t r0 = 0; r0 < R; r0 += WAVE) {
            const uint32_t j = r0 + lane;
            const bool on = j < R;
            const uint32_t s = HISTORY + 4u * (on ? j : 0u);
            // ends of the header and bases lines first, of the plus and quality lines when their turn comes (the values are
            // read where they are used: this runs in the middle of a tile with its 16 data registers live, and two more
            // registers held across the histogram update were two spilled ones at 7 waves per SIMD)
            const uint32_t ph = L.pos[s], pb = L.pos[s + 1u];
            const uint32_t ls = pb - ph - 1u;
            add_length(on, ls, s_hist, D);
            if (on) {
                sumlen += ls;
                nrec += 1;
                if (next_of(L.nc[s], ph) == '+') err |= ERR_BAD_PLUS;        // a non-empty sequence line must not start with '+'
                if (next_of(L.nc[s + 1u], pb) != '+') err |= ERR_BAD_PLUS;
                const uint32_t pp = L.pos[s + 2u], pq = L.pos[s + 3u];
                if (pq - pp - 1u != ls) err |= ERR_LEN_MISMATCH;
                if (pq + 1u < end_rel && next_of(L.nc[s + 3u], pq) != '@') err |= ERR_BAD_HEADER;
            }
        }
        (void)wb; (void)tile_idx; (void)tile_rel;
    }

    // FASTA, round 5 (stream_fasta2_dev.hpp): E <= 64 PUBLISHED newlines -- the ones in front of a '>' (closing) and the ones
    // that end a header line -- with key = position - newline index; the rules of batch()'s FASTA branch on them.
    // GAPS (`stats -a`): the skeleton counts the gap letters among ALL bytes of the range (rgap); the letters of the HEADER
    // lines are no sequence bytes -- every header-end event counts those of its line (from the newline before it, which
    // closed a record and is therefore the event before this one, to its own) into rsub.  Headers are a few bytes per
    // record: one lane each, byte loads.
    bool f2 = false;                 // this range ran on the pass of stream_fasta2_dev.hpp (end_range: no per-record gap sums)
    uint32_t rsub = 0;               // GAPS: gap letters inside header lines (per lane and range)
    uint32_t last_pub_pos = 0xFFFFFFFFu;  // position of the last published newline of the range (~0: none, the range start is a line start)
    const uint8_t* f2_buf = nullptr;
    uint64_t f2_rs = 0, f2_n = 0, f2_skip_lo = ~0ull, f2_skip_hi = 0;   // bytes at [skip_lo, skip_hi) are counted by other ranges (RF_MID)
    template <bool GAPS>
    __device__ __forceinline__ void events2(LdsF2T<GAPS>& L, uint32_t first, uint32_t E) {
        const uint32_t lane = threadIdx.x & 63;
        const bool on = lane < E;
        const uint32_t ei = (first + lane) & (F2_EVENTS - 1u);
        uint32_t key = 0, fl = 0;
        if (on) { key = L.ekey[ei]; fl = L.eflag[ei]; }
        const bool closing = (fl & F2_CLOSING) != 0u, hdr_end = (fl & F2_HDR_END) != 0u;
        if constexpr (GAPS) {
            uint32_t pos = 0;
            if (on) pos = L.epos[ei];
            uint32_t prev = (uint32_t)__shfl_up((int)pos, 1, 64);
            if (lane == 0) prev = last_pub_pos;
            if (hdr_end) {
                // the gap letters of the header line [prev + 1, pos): sixteen bytes per load (until round 5 one byte per
                // load, ten dependent loads for `>r0000001` on the few lanes that hold a header); a chunk without a byte
                // at or below the largest gap letter -- a name without blanks -- is done after the test
                const PredConsts& P = D.pred;
                uint32_t x = prev + 1u;   // (prev + 1 wraps to 0 at the range start)
                while (x < pos) {
                    const uint64_t a = f2_rs + x;
                    const uint32_t len = pos - x < 16u ? pos - x : 16u;
                    if (a + 16 <= f2_n && (a + 16 <= f2_skip_lo || a >= f2_skip_hi)) {
                        uint32_t w[4];
                        __builtin_memcpy(w, f2_buf + a, 16);
                        uint32_t low = 0, m[4];
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const uint32_t nv = len > 4u * d ? (len - 4u * d >= 4u ? 4u : len - 4u * d) : 0u;
                            m[d] = nv >= 4u ? 0x80808080u : (0x80808080u & ((1u << (8u * nv)) - 1u));
                            low |= ~ge_bytes(w[d], P.kgap) & m[d];
                        }
                        if (low != 0u || P.kgap == 0xFFFFFFFFu) {
                            // (static indices: a run-time index into the sink's copy of the constants would put the whole sink into scratch memory)
#pragma unroll
                            for (int k = 0; k < MAX_GAP_LETTERS; ++k) {
                                if (k < P.ngap) {
                                    const uint32_t rep = P.gap_rep[k];
                                    rsub += (uint32_t)__popc((zero_bytes(w[0] ^ rep) & m[0]) >> 7) + (uint32_t)__popc((zero_bytes(w[1] ^ rep) & m[1]) >> 7) +
                                            (uint32_t)__popc((zero_bytes(w[2] ^ rep) & m[2]) >> 7) + (uint32_t)__popc((zero_bytes(w[3] ^ rep) & m[3]) >> 7);
                                }
                            }
                        }
                    } else {  // at the end of the shard, or across bytes that other ranges count: byte by byte
                        for (uint32_t y = x; y < x + len; ++y) {
                            const uint64_t b = f2_rs + y;
                            if (b >= f2_skip_lo && b < f2_skip_hi) continue;
                            const uint32_t c = f2_buf[b];
#pragma unroll
                            for (int k = 0; k < MAX_GAP_LETTERS; ++k) rsub += (k < P.ngap && (P.gap_rep[k] & 0xFFu) == c) ? 1u : 0u;
                        }
                    }
                    x += len;
                }
            }
            if (E) last_pub_pos = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)(E - 1u));
        }
        const uint64_t hb = __ballot(hdr_end);
        const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
        const uint64_t hh = hb & upto;
        const int src = hh ? 63 - __clzll((long long)hh) : (int)lane;
        const uint32_t key_h = (uint32_t)__shfl((int)key, src, 64);
        uint32_t seqlen = 0;
        bool whole = true;  // header and closing line of the record are both in this range
        if (closing) {
            uint32_t key_i;
            if (hh) key_i = key_h;
            else { key_i = open_key; whole = open_is_header; }  // (else the record began in an earlier range: only its tail is here)
            seqlen = key - key_i;
            if (whole) { sumlen += seqlen; nrec += 1; }
            else { D.r_head[range_id] = seqlen; atomicOr(&D.r_flags[range_id], RF_HEAD_CLOSED); }
        }
        add_length(closing && whole, seqlen, s_hist, D);
        if (hb) {  // carry the last header end of this group (wave-uniform)
            const int last = 63 - __clzll((long long)hb);
            open_key = (uint32_t)__builtin_amdgcn_readlane((int)key, last);
            open_is_header = true;
        }
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
                const uint32_t p1 = L.pos[s - 1];
                const uint32_t len = p - p1 - 1u;
                const bool is_seq = on && role == 1u;
                add_length(is_seq, len, s_hist, D);
                if (on) {
                    if (role == 1u) {
                        sumlen += len;
                        if constexpr (ALL) gap += (uint32_t)(L.c[s] - L.c[s - 1]);
                        if (next_char(L, s, abs_next, re, buf) != '+') err |= ERR_BAD_PLUS;
                    } else if (role == 3u) {
                        const uint32_t slen = L.pos[s - 2] - L.pos[s - 3] - 1u;
                        if (len != slen) err |= ERR_LEN_MISMATCH;
                        if constexpr (ALL) {
                            q20 += (uint32_t)(L.a[s] - L.a[s - 1]);
                            q30 += (uint32_t)(L.b[s] - L.b[s - 1]);
                        }
                        nrec += 1;
                        if (abs_next < re && next_char(L, s, abs_next, re, buf) != '@') err |= ERR_BAD_HEADER;
                    } else if (role == 0u) {
                        // a non-empty sequence line must not start with '+'
                        if (next_char(L, s, abs_next, re, buf) == '+') err |= ERR_BAD_PLUS;
                    }
                }
            } else {
                const bool closing = on && L.flag[s] != 0;
                const bool hdr_end = on && L.flag[s - 1] != 0;  // the line after a closing line is a header
                const uint32_t key = p - rank;                  // bases before this newline
                uint32_t sg = 0;
                if constexpr (ALL) sg = L.a[s];
                // the header-end event of the record a lane closes: the last one at or below it in this group of 64
                // events (ballot + shuffle), else the one carried in open_key (a walk back through the LDS events cost
                // a 5 kb record ~85 dependent reads on one lane)
                const uint64_t hb = __ballot(hdr_end);
                const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
                const uint64_t hh = hb & upto;
                const int src = hh ? 63 - __clzll((long long)hh) : lane;
                const uint32_t key_h = (uint32_t)__shfl((int)key, src, 64);
                uint32_t sg_h = 0;
                if constexpr (ALL) sg_h = (uint32_t)__shfl((int)sg, src, 64);
                uint32_t seqlen = 0;
                bool whole = true;  // header and closing line of the record are both in this range
                if (closing) {
                    uint32_t key_i, sg_i = 0;
                    if (hh) {
                        key_i = key_h;
                        sg_i = sg_h;
                    } else {
                        key_i = open_key;
                        sg_i = open_sg;
                        whole = open_is_header;  // else the record began in an earlier range: only its tail is here
                    }
                    seqlen = key - key_i;
                    if (whole) {
                        sumlen += seqlen;
                        nrec += 1;
                    } else {
                        D.r_head[range_id] = seqlen;
                        atomicOr(&D.r_flags[range_id], RF_HEAD_CLOSED);
                    }
                    if constexpr (ALL) gap += (uint32_t)(sg - sg_i);
                }
                add_length(closing && whole, seqlen, s_hist, D);
                if (hb) {  // carry the last header end of this group (wave-uniform)
                    const int last = 63 - __clzll((long long)hb);
                    open_key = (uint32_t)__builtin_amdgcn_readlane((int)key, last);
                    if constexpr (ALL) open_sg = (uint32_t)__builtin_amdgcn_readlane((int)sg, last);
                    open_is_header = true;
                }
            }
        }
        if constexpr (!FASTQ) {
            if (E > 0) {  // the last event of the batch (uniform LDS reads)
                const uint32_t sl = HISTORY + (E - 1u);
                last_key = L.pos[sl] - (wb + (E - 1u));
                last_closing = L.flag[sl] != 0;
                last_line_hdr = L.flag[sl - 1u] != 0;
                if constexpr (ALL) last_a = L.a[sl];
                any_event = true;
            }
        }
    }
};

// 7 waves per SIMD (<= 72 VGPRs): measured 18.27 -> 17.72 ms (stats) against the compiler's own choice (74 VGPRs,
// 6 waves).  The FASTQ -a kernel on the dense path (BSK_STATS_A=dense) needs ~90 VGPRs: at 7 waves it spills 17 of
// them (24 GB of scratch writes per 100 GB pass, PMC) -- 5 waves without spills: 40.7 -> 37.4 ms.
#ifndef BSK_STATS_WAVES
#define BSK_STATS_WAVES 7
#endif
// FASTQ -a: by line roles on the sparse path (ROLES_T, the default) or by running counters on the dense path
// (BSK_STATS_A=dense at run time: the two count the same bytes in unrelated ways, tests hold them against each other)
#ifndef BSK_STATS_WAVES_ALL
#define BSK_STATS_WAVES_ALL 5
#endif
// FASTQ default row: with non-temporal tile loads the pass is bound by HBM alone -- 15.42 ms at 6, 7 and 8 waves per SIMD
// (scripts/history/r04_nt_waves.sh); at 6 nothing spills (at 7: two registers)
#ifndef BSK_STATS_WAVES_FQ
#define BSK_STATS_WAVES_FQ 6
#endif
#if BSK_STATS_WAVES
#ifndef BSK_STATS_WAVES_FA_ALL
#define BSK_STATS_WAVES_FA_ALL 5   // FASTA -a on the pass of stream_fasta2_dev.hpp (ROLES_T): ~100 VGPRs wanted
#endif
#define BSK_STATS_ATTR __attribute__((amdgpu_waves_per_eu((ALL && FASTQ) ? BSK_STATS_WAVES_ALL : (FASTQ ? BSK_STATS_WAVES_FQ : ((ALL && ROLES_T) ? BSK_STATS_WAVES_FA_ALL : BSK_STATS_WAVES)), 8)))
#else
#define BSK_STATS_ATTR
#endif

template <bool FASTQ, bool ALL, bool DPP, bool ROLES_T = true>
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) BSK_STATS_ATTR void k_stats(const uint8_t* __restrict__ buf, uint64_t n,
                                                                   const uint64_t* __restrict__ anchors,
                                                                   uint32_t nranges, uint32_t* __restrict__ queue,
                                                                   StatsDev D, uint64_t chunk) {
    __shared__ uint32_t s_hist[LDS_HIST + 2 * BIG_SLOTS];  // dense bins, then the (length, count) cache of add_big
    constexpr bool ROLES = FASTQ && ALL && ROLES_T;
    constexpr bool SALL = ALL && !ROLES;  // what the skeleton and the events see
    // FASTQ on the sparse path: whole records, deferred.  The default row runs at 7 waves per SIMD = 7 blocks per CU: 22.8 KB
    // of LDS each, of which the length histogram takes 8.7 -- a window of 256 events fits, 512 would cost two waves per SIMD
    // (measured before: 7 waves against 6 is 3 % of the kernel); `-a` by line roles runs at 5 waves and takes the full window
    constexpr int WINDOW = (FASTQ && !SALL) ? 256 : CAP;
    // FASTA default row, round 5: only the newlines the sink acts on become events (stream_fasta2_dev.hpp); ROLES_T = false keeps
    // the pass that publishes every newline (stats_fasta=events: the tests hold the two against each other)
    constexpr bool F2 = !FASTQ && ROLES_T;   // (`-a` too: the gap letters are counted over all bytes, the header lines taken off)
    __shared__ Lds<FASTQ, SALL, F2 ? 4 : WINDOW> s_l[F2 ? 1 : WAVES_PER_BLOCK];
    __shared__ LdsF2T<F2 && ALL> s_f2[F2 ? WAVES_PER_BLOCK : 1];
    for (int i = threadIdx.x; i < LDS_HIST + 2 * BIG_SLOTS; i += blockDim.x)
        s_hist[i] = (i >= LDS_HIST && i < LDS_HIST + BIG_SLOTS) ? BIG_EMPTY : 0u;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Lds<FASTQ, SALL, F2 ? 4 : WINDOW>& L = s_l[F2 ? 0 : wave];
    StatsSinkT<ROLES> sink;
    sink.s_hist = s_hist;
    sink.D = D;
    const uint64_t n_eff = anchors[nranges];
    uint64_t w20 = 0, w30 = 0, wgap = 0;  // ROLES: wave totals
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(queue, 1u);
        r = wave_first(r);
        if (r >= nranges) break;
        uint64_t rs = anchors[r], re = anchors[r + 1];
        rs = rs < n_eff ? rs : n_eff;
        re = re < n_eff ? re : n_eff;
        if (rs >= re) {
            if constexpr (!FASTQ && ALL) {
                // an empty range inside a long line (k_prep found no line start in its nominal chunk): `-a` counts the gap
                // letters of the chunk here; k_stats_stitch adds them if the line is a sequence line
                if (chunk && r > 0u) {
                    const uint64_t lo = (uint64_t)r * chunk, hi = lo + chunk < n_eff ? lo + chunk : n_eff;
                    if (lo < hi) {
                        uint64_t cnt = 0;
                        for (uint64_t i = lo + (uint64_t)lane * 16u; i < hi; i += 64u * 16u) {
                            const uint4 v = load16(buf, hi, i);  // (bytes past hi read as 0)
                            for (int k = 0; k < D.pred.ngap; ++k) {
                                const uint32_t rep = D.pred.gap_rep[k];
                                if (rep == 0u && i + 16u > hi) continue;  // (a NUL gap letter against the zero padding: never in text)
                                cnt += popc4(zero_bytes(v.x ^ rep), zero_bytes(v.y ^ rep), zero_bytes(v.z ^ rep), zero_bytes(v.w ^ rep));
                            }
                        }
                        cnt = wave_sum_u64(cnt);
                        if (lane == 0) { D.r_head[r] = cnt; atomicOr(&D.r_flags[r], RF_MID); }
                    }
                }
            }
            continue;
        }
        sink.begin_range(r);
        // FASTA: a range that reaches beyond its nominal chunk ends with ONE long line (k_prep found no line start in the
        // chunks it covers): the default row needs that line's end, not its bytes; `-a` leaves the gap letters of those
        // chunks to their own ranges (above)
        const uint64_t skip_from = (!FASTQ && chunk) ? (uint64_t)(r + 1u) * chunk : ~0ull;
        const uint64_t count_resume = (!FASTQ && chunk) ? (re == n_eff ? re : (re / chunk) * chunk) : 0ull;
        if constexpr (F2) {
            sink.f2 = true;
            if constexpr (ALL) { sink.f2_buf = buf; sink.f2_rs = rs; sink.f2_n = n; sink.f2_skip_lo = skip_from; sink.f2_skip_hi = count_resume; }
            const F2Tail T = stream_range_fasta2<DPP, ALL>(s_f2[wave], buf, n, rs, re, re == n_eff, sink, D.pred, skip_from, count_resume);
            sink.any_event = T.lines != 0u;
            sink.last_closing = T.last_closing;
            sink.last_key = T.last_key;
            sink.last_line_hdr = T.last_hdr;
            if constexpr (ALL) {
                // (per-lane differences may be negative: the sum over the wave, modulo 2^64, is what counts)
                sink.gap += (uint64_t)sink.rgap - (uint64_t)sink.rsub;
                sink.rgap = 0; sink.rsub = 0;
            }
            (void)L;
        } else {
            stream_range<FASTQ, SALL, DPP>(L, buf, n, rs, re, re == n_eff, D.pred, sink, skip_from, count_resume);
        }
        if constexpr (!FASTQ) sink.template end_range<ALL>();
        if constexpr (ROLES) {
            // wave totals of the range (uniform: they live in scalar registers between ranges)
            const uint64_t s20 = wave_sum_u64(sink.rq20), s30 = wave_sum_u64(sink.rq30), sgap = wave_sum_u64(sink.rgap);
            w20 += ((uint64_t)wave_first((uint32_t)(s20 >> 32)) << 32) | wave_first((uint32_t)s20);
            w30 += ((uint64_t)wave_first((uint32_t)(s30 >> 32)) << 32) | wave_first((uint32_t)s30);
            wgap += ((uint64_t)wave_first((uint32_t)(sgap >> 32)) << 32) | wave_first((uint32_t)sgap);
            sink.rq20 = sink.rq30 = sink.rgap = 0;
        }
    }
    // flush ---------------------------------------------------------------
    const uint64_t q20 = ROLES ? w20 : wave_sum_u64(sink.q20), q30 = ROLES ? w30 : wave_sum_u64(sink.q30),
                   gap = ROLES ? wgap : wave_sum_u64(sink.gap);
    const uint64_t nrec = wave_sum_u64(sink.nrec), sumlen = wave_sum_u64(sink.sumlen);
    const uint32_t err = wave_or_u32(sink.err);
    if (lane == 0) {
        if (q20) atomicAdd((unsigned long long*)&D.vec[0], (unsigned long long)q20);
        if (q30) atomicAdd((unsigned long long*)&D.vec[1], (unsigned long long)q30);
        if (gap) atomicAdd((unsigned long long*)&D.vec[2], (unsigned long long)gap);
        if (nrec) atomicAdd((unsigned long long*)&D.vec[3], (unsigned long long)nrec);
        if (sumlen) atomicAdd((unsigned long long*)&D.vec[6], (unsigned long long)sumlen);
        if (err) atomicOr((unsigned long long*)&D.status[0], (unsigned long long)err);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LDS_HIST; i += blockDim.x) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd((unsigned long long*)&D.vec[STATS_HDR + i], (unsigned long long)c);
    }
    for (int i = threadIdx.x; i < BIG_SLOTS; i += blockDim.x) {
        const uint32_t k = s_hist[LDS_HIST + i], c = s_hist[LDS_HIST + BIG_SLOTS + i];
        if (k != BIG_EMPTY && c) atomicAdd((unsigned long long*)&D.vec[STATS_HDR + k], (unsigned long long)c);
    }
}

// ---------------------------------------------------------------------------
// k_prep: range anchors (ReadFixer equivalent) + effective end of the shard
// anchors[0] = 0, anchors[r] = first record start >= r * chunk, anchors[nranges] = n_eff
// ---------------------------------------------------------------------------
template <bool FASTQ>
__global__ __launch_bounds__(256) void k_prep(const uint8_t* __restrict__ buf, uint64_t n, uint64_t chunk, uint32_t nranges,
                                              uint64_t* __restrict__ anchors, uint32_t* __restrict__ queue, int line_mode) {
    // one WAVE per range boundary (anchor_wave_dev.hpp): the line ends are searched 1 KiB per step
    const uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool lane0 = (threadIdx.x & 63u) == 0u;
    if (r > nranges) return;
    if (r == 0) {
        if (lane0) { anchors[0] = 0; *queue = 0; }
        return;
    }
    if (r == nranges) {
        if (lane0) anchors[nranges] = effective_end(buf, n);
        return;
    }
    const uint64_t from = (uint64_t)r * chunk;
    if (!FASTQ && line_mode == 2) {
        // a line start inside this boundary's OWN chunk, or none: k_prep_fill gives a boundary without one the next
        // boundary's anchor (an empty range), and the range that holds the long line skips the chunks in between
        const uint64_t a = wave_anchor::find_line_start_within(buf, n, from, from + chunk);
        if (lane0) anchors[r] = a;  // (ANCHOR_NONE stays until k_prep_fill)
        return;
    }
    const uint64_t a = FASTQ ? wave_anchor::find_fastq_start(buf, n, from, from + ANCHOR_SEARCH_BYTES)
                             : (line_mode ? wave_anchor::find_line_start(buf, n, from) : wave_anchor::find_fasta_start(buf, n, from));
    // No record start within reach (text that is not FASTQ): the range begins at the raw boundary.  The streaming pass
    // validates every line it reads, so it reports the malformed text itself -- a check of the anchors inside its range loop
    // cost k_stats two spilled registers and 1 ms at 100 GB.
    if (lane0) anchors[r] = a == ANCHOR_NONE ? from : a;
}

// FASTA: raw[r] = the line start k_prep found in chunk r, or ANCHOR_NONE -> anchors[r] = the first anchor at or after r
// (raw[nranges] = the effective end is always one).  Runs of chunks without a line start are one long line.
__global__ __launch_bounds__(256) void k_prep_fill(const uint64_t* __restrict__ raw, uint32_t nranges, uint64_t* __restrict__ anchors) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nranges) return;
    uint32_t j = r;
    while (j < nranges && raw[j] == ANCHOR_NONE) ++j;
    anchors[r] = raw[j];
}

// Records that cross range boundaries (FASTA, line-start ranges).  A record that is open at the end of range r (its
// header is in r) is finished by the thread of r: it walks the following ranges, adding the parts that have no header
// of their own, up to the range where the record ends.  The chains are disjoint, so reads (every range leaves one short
// chain) run fully parallel and a chromosome costs one thread a walk over the ranges it spans.  (A single thread over
// all ranges cost 6 ms for 15 000 ranges -- more than the 1 GB pass it followed.)
__global__ __launch_bounds__(256) void k_stats_stitch(uint32_t nranges, StatsDev D) {
    // (bins below 2 048 and, behind them, the (length, count) cache of add_big: 95 000 ranges of a file of equally long 5 kb
    // records each added 1 to ONE global counter -- 1.15 ms of same-address atomics behind a 7.9 ms pass, round 5)
    __shared__ uint32_t s_hist[LDS_HIST + 2 * BIG_SLOTS];
    __shared__ unsigned long long s_nrec, s_sum;
    static_assert(LDS_HIST == 2048, "the flush below");
    for (uint32_t k = threadIdx.x; k < (uint32_t)(LDS_HIST + 2 * BIG_SLOTS); k += blockDim.x)
        s_hist[k] = (k >= (uint32_t)LDS_HIST && k < (uint32_t)(LDS_HIST + BIG_SLOTS)) ? BIG_EMPTY : 0u;
    if (threadIdx.x == 0) { s_nrec = 0; s_sum = 0; }
    __syncthreads();
    const uint32_t r0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (r0 < nranges && (D.r_flags[r0] & (RF_VISITED | RF_TAIL_OPEN)) == (RF_VISITED | RF_TAIL_OPEN)) {
        uint64_t len = D.r_tail[r0];
        for (uint32_t r = r0 + 1; r < nranges; ++r) {
            const uint32_t f = D.r_flags[r];
            if (!(f & RF_VISITED)) continue;
            len += D.r_head[r];  // has a header: the bases before it; no header: the whole range
            if (f & (RF_HAS_HEADER | RF_HEAD_CLOSED)) break;
        }
        atomicAdd(&s_nrec, 1ull);
        atomicAdd(&s_sum, (unsigned long long)len);
        if (len < 2048u) atomicAdd(&s_hist[len], 1u);
        else if (len < D.hist_cap) add_big((uint32_t)len, 1u, s_hist, D);
        else {
            atomicAdd((unsigned long long*)&D.vec[5], 1ull);
            const uint64_t i = atomicAdd((unsigned long long*)&D.status[1], 1ull);
            if (i < D.overflow_cap) D.overflow[i] = len;
        }
    }
    // `-a`: the gap letters of the chunks that a long SEQUENCE line covers (counted by their own, otherwise empty, ranges)
    if (r0 < nranges && (D.r_flags[r0] & (RF_VISITED | RF_SKIP_SEQ)) == (RF_VISITED | RF_SKIP_SEQ)) {
        unsigned long long g = 0;
        for (uint32_t r = r0 + 1; r < nranges && (D.r_flags[r] & RF_MID); ++r) g += D.r_head[r];
        if (g) atomicAdd((unsigned long long*)&D.vec[2], g);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 2048u; k += blockDim.x)
        if (s_hist[k] && k < D.hist_cap) atomicAdd((unsigned long long*)&D.vec[STATS_HDR + k], (unsigned long long)s_hist[k]);
    for (uint32_t k = threadIdx.x; k < (uint32_t)BIG_SLOTS; k += blockDim.x) {
        const uint32_t key = s_hist[LDS_HIST + k], c = s_hist[LDS_HIST + BIG_SLOTS + k];
        if (key != BIG_EMPTY && c) atomicAdd((unsigned long long*)&D.vec[STATS_HDR + key], (unsigned long long)c);
    }
    if (threadIdx.x == 0 && s_nrec) {
        atomicAdd((unsigned long long*)&D.vec[3], s_nrec);
        atomicAdd((unsigned long long*)&D.vec[6], s_sum);
    }
}

// Pure streaming read with the access pattern of k_stats (same tiles, same queue, no
// per-byte work): calibrates FETCH_SIZE for this pattern and gives the read ceiling.
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) void k_stream_read(const uint8_t* __restrict__ buf, uint64_t n,
                                                                         uint64_t chunk, uint32_t nranges,
                                                                         uint32_t* __restrict__ queue,
                                                                         uint32_t* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(queue, 1u);
        r = wave_first(r);
        if (r >= nranges) break;
        const uint64_t rs = (uint64_t)r * chunk, re = (rs + chunk < n) ? rs + chunk : n;
        for (uint64_t t = rs; t < re; t += TILE) {
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) {
                const uint64_t idx = t + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16;
                if (idx + 16 <= re) {
                    const uint4 v = *reinterpret_cast<const uint4*>(buf + idx);
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                }
            }
        }
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;  // keeps the loads alive; practically never true
}

// self-test of the wave scan (run by tests on the GPU)
template <bool DPP>
__global__ void k_scan_selftest(const uint32_t* in, uint32_t* out) {
    out[threadIdx.x] = wave_incl_scan<DPP>(in[threadIdx.x]);
}

}  // namespace

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
hipError_t launch_prep(bool fastq, const uint8_t* buf, uint64_t n, uint64_t chunk, uint32_t nranges,
                       uint64_t* anchors, uint32_t* queue, hipStream_t st, bool line_mode, uint64_t* raw) {
    const int threads = 256;  // four boundaries per block, a wave each
    const int blocks = (int)(((uint64_t)nranges + 1 + 3) / 4);
    if (fastq) hipLaunchKernelGGL(k_prep<true>, dim3(blocks), dim3(threads), 0, st, buf, n, chunk, nranges, anchors, queue, 0);
    else if (line_mode && raw) {
        // every boundary searches its own chunk only (raw), then the boundaries without a line start take the next anchor
        hipLaunchKernelGGL(k_prep<false>, dim3(blocks), dim3(threads), 0, st, buf, n, chunk, nranges, raw, queue, 2);
        hipLaunchKernelGGL(k_prep_fill, dim3((nranges + 1 + 255) / 256), dim3(256), 0, st, (const uint64_t*)raw, nranges, anchors);
    } else hipLaunchKernelGGL(k_prep<false>, dim3(blocks), dim3(threads), 0, st, buf, n, chunk, nranges, anchors, queue, line_mode ? 1 : 0);
    return hipGetLastError();
}

// one past the highest non-empty bin of the length histogram (bsk_stats_collect copies only that much to the host)
__global__ __launch_bounds__(1024) void k_hist_extent(const uint64_t* __restrict__ hist, uint32_t cap, uint64_t* __restrict__ out) {
    __shared__ uint32_t s_hi;
    if (threadIdx.x == 0) s_hi = 0;
    __syncthreads();
    uint32_t hi = 0;
    for (uint32_t i = threadIdx.x; i < cap; i += 1024u)
        if (hist[i]) hi = i + 1;
    if (hi) atomicMax(&s_hi, hi);
    __syncthreads();
    if (threadIdx.x == 0) *out = s_hi;
}

hipError_t launch_hist_extent(const uint64_t* hist, uint32_t cap, uint64_t* out, hipStream_t st) {
    hipLaunchKernelGGL(k_hist_extent, dim3(1), dim3(1024), 0, st, hist, cap, out);
    return hipGetLastError();
}

namespace {
struct GapSet { uint32_t w[8]; };
__global__ __launch_bounds__(256) void k_gap_set_count(const uint8_t* __restrict__ buf, RecordTable t, GapSet S, int fastq,
                                                       unsigned long long* __restrict__ gap_slot) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    uint64_t acc = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < t.n; r += nwaves) {
        const uint64_t from = t.start[r] + t.l_head[r] + 1;
        // FASTQ: the one sequence line; FASTA: the text up to the next record (its line breaks are no sequence bytes)
        const uint64_t to = fastq ? from + t.l_seq[r] : t.start[r + 1];
        for (uint64_t i = from + lane; i < to; i += 64) {
            const uint8_t c = buf[i];
            if (c != '\n' && ((S.w[c >> 5] >> (c & 31u)) & 1u)) ++acc;
        }
    }
    acc = wave_sum_u64(acc);
    if (lane == 0 && acc) atomicAdd(gap_slot, (unsigned long long)acc);
}
}  // namespace

hipError_t launch_gap_set_count(const uint8_t* buf, const RecordTable& t, const uint32_t (&set)[8], bool fastq, uint64_t* gap_slot,
                                hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    GapSet S;
    for (int i = 0; i < 8; ++i) S.w[i] = set[i];
    const uint64_t blocks = std::min<uint64_t>((t.n + 3) / 4, 256ull * 32ull);
    hipLaunchKernelGGL(k_gap_set_count, dim3((unsigned)blocks), dim3(256), 0, st, buf, t, S, fastq ? 1 : 0, (unsigned long long*)gap_slot);
    return hipGetLastError();
}

hipError_t launch_stats_stitch(uint32_t nranges, const StatsDev& D, hipStream_t st) {
    hipLaunchKernelGGL(k_stats_stitch, dim3((nranges + 255u) / 256u), dim3(256), 0, st, nranges, D);
    return hipGetLastError();
}

template <bool FASTQ, bool ALL, bool ROLES_T = true>
static hipError_t launch_stats_t(bool dpp, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors,
                                 uint32_t nranges, uint32_t* queue, const StatsDev& D, hipStream_t st, uint64_t chunk) {
    const dim3 b(WAVES_PER_BLOCK * WAVE);
    if (dpp) hipLaunchKernelGGL((k_stats<FASTQ, ALL, true, ROLES_T>), dim3(blocks), b, 0, st, buf, n, anchors, nranges, queue, D, chunk);
    else hipLaunchKernelGGL((k_stats<FASTQ, ALL, false, ROLES_T>), dim3(blocks), b, 0, st, buf, n, anchors, nranges, queue, D, chunk);
    return hipGetLastError();
}

hipError_t launch_stats(bool fastq, bool all, bool dpp, int blocks, const uint8_t* buf, uint64_t n,
                        const uint64_t* anchors, uint32_t nranges, uint32_t* queue, const StatsDev& D,
                        hipStream_t st, bool a_dense, uint64_t skip_chunk) {
    if (fastq && all && a_dense) return launch_stats_t<true, true, false>(dpp, blocks, buf, n, anchors, nranges, queue, D, st, 0);
    if (fastq) return all ? launch_stats_t<true, true>(dpp, blocks, buf, n, anchors, nranges, queue, D, st, 0)
                          : launch_stats_t<true, false>(dpp, blocks, buf, n, anchors, nranges, queue, D, st, 0);
    // (FASTA: a_dense = the pass that publishes every newline -- with `-a` the dense path with running counters --, stats_fasta=events)
    if (all) return a_dense ? launch_stats_t<false, true, false>(dpp, blocks, buf, n, anchors, nranges, queue, D, st, skip_chunk)
                            : launch_stats_t<false, true>(dpp, blocks, buf, n, anchors, nranges, queue, D, st, skip_chunk);
    return a_dense ? launch_stats_t<false, false, false>(dpp, blocks, buf, n, anchors, nranges, queue, D, st, skip_chunk)
                   : launch_stats_t<false, false>(dpp, blocks, buf, n, anchors, nranges, queue, D, st, skip_chunk);
}

int stats_max_blocks_per_cu(bool fastq, bool all, bool dpp, bool a_dense) {
    int nb = 0;
    const void* f = nullptr;
#define BSK_PICK(FQ, AL, RO)                                                             \
    f = dpp ? (const void*)k_stats<FQ, AL, true, RO> : (const void*)k_stats<FQ, AL, false, RO>;
    if (fastq && all && a_dense) { BSK_PICK(true, true, false) }
    else if (fastq && all) { BSK_PICK(true, true, true) }
    else if (fastq) { BSK_PICK(true, false, true) }
    else if (all && a_dense) { BSK_PICK(false, true, false) }
    else if (all) { BSK_PICK(false, true, true) }
    else if (a_dense) { BSK_PICK(false, false, false) }
    else { BSK_PICK(false, false, true) }
#undef BSK_PICK
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, WAVES_PER_BLOCK * WAVE, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

hipError_t launch_stream_read(int blocks, const uint8_t* buf, uint64_t n, uint64_t chunk, uint32_t nranges,
                              uint32_t* queue, uint32_t* sink, hipStream_t st) {
    hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(WAVES_PER_BLOCK * WAVE), 0, st, buf, n, chunk, nranges, queue,
                       sink);
    return hipGetLastError();
}

hipError_t launch_scan_selftest(bool dpp, const uint32_t* in, uint32_t* out, hipStream_t st) {
    if (dpp) hipLaunchKernelGGL(k_scan_selftest<true>, dim3(1), dim3(64), 0, st, in, out);
    else hipLaunchKernelGGL(k_scan_selftest<false>, dim3(1), dim3(64), 0, st, in, out);
    return hipGetLastError();
}

}  // namespace bsk
