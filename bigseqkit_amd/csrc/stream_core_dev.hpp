// ============================================================================
// stream_core_dev.hpp -- the single-pass streaming skeleton shared by every kernel
// that has to find records in raw FASTA/FASTQ text on gfx950 (device code only).
//
// Replaces the reference's line-by-line SeqParser.Read
// (/root/reference/bigseqkit-lib/helper.go:219-325) for one shard in HBM:
//   * persistent 64-lane waves stream byte RANGES that begin on a record
//     (anchors from k_prep, the ReadFixer equivalent, helper.go:41-66);
//   * 4 KiB tiles of coalesced 16-byte loads, every byte read once;
//   * SWAR byte predicates -> popcounts -> packed DPP prefix scans;
//   * newline EVENTS (position + running counters) in a per-wave LDS window;
//   * a Sink consumes the events, one lane per event (stats: histogram and
//     counters; index: record table).
// HBM-bound integer/byte work: no MFMA anywhere.
// ============================================================================
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "stream_stats.hpp"  // error flags, PredConsts

namespace bsk {

namespace stream {

constexpr int WAVE = 64;
#ifndef BSK_NPIECE
#define BSK_NPIECE 4
#endif
#ifndef BSK_NL_PREFILTER
#define BSK_NL_PREFILTER 1  // 16.83 -> 16.65 ms per 100 GB (k_stats), 25.2 -> 24.95 (stats -a): scripts/r03_statsvar.sh
#endif
#ifndef BSK_NL_SGPR
#define BSK_NL_SGPR 1  // k_stats unchanged, stats -a 24.95 -> 24.67 ms (scripts/r03_statsvar.sh)
#endif
#ifndef BSK_PREFETCH
#define BSK_PREFETCH 0  // measured: occupancy (7-8 waves/SIMD) hides HBM latency better than a register prefetch
#endif
constexpr int NPIECE = BSK_NPIECE;        // 16-byte pieces per lane per tile
constexpr int PIECE_BYTES = WAVE * 16;    // 1 KiB per wave-piece
constexpr int TILE = PIECE_BYTES * NPIECE;  // 4 KiB per wave-tile
constexpr int CAP = 128;                  // newline events per LDS batch
constexpr int REC_WINDOW = 512;           // ... of the sinks that take whole FASTQ records (sink_records4): the window holds the
                                          // events of several tiles; the sink runs at the END of a tile once 256 are pending
constexpr int HISTORY = 4;                // events kept from the previous batch
constexpr int SLOTS = HISTORY + CAP;
constexpr int LDS_HIST = 2048;
constexpr int WAVES_PER_BLOCK = 4;

// ---------------------------------------------------------------------------
// wave primitives
// ---------------------------------------------------------------------------
template <bool DPP>
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    if constexpr (DPP) {
        // Hillis-Steele inside each row of 16 lanes (row_shr 1,2,4,8; lanes
        // without a source read 0 via bound_ctrl), then row_bcast:15 into rows
        // 1 and 3, then row_bcast:31 into rows 2 and 3.
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
        return v;
    } else {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = (uint32_t)__shfl_up((int)v, d, 64);
            if (lane >= d) v += t;
        }
        return v;
    }
}

__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ uint32_t wave_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64);
        uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

// make one wave's LDS writes visible to its other lanes (DS ops of a wave are
// executed in order; this only stops the compiler from reordering them)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// A kernel argument that is needed once per range, read from the kernel's argument block WHEN it is needed instead of
// living in a scalar register through the pass.  Round 6: every streaming pass wants ~150 scalar registers (the arguments,
// the state of range / tile / round, four ballots, one exec mask per level of divergent control flow) where a wave has
// 94 - 102, and the compiler keeps the rest in lanes of a vector register: a v_writelane per definition and a v_readlane
// per use -- VALU instructions in passes that are bound by VALU issue.  The kernel takes ONE struct; BSK_KARG(Args, member)
// is a scalar load from the argument block behind an opaque copy of its address (so that the load is neither hoisted out of
// the loop nor merged with the loads at kernel entry).  Measured on k_names (scripts/r06_ab5.sh): 68 -> 55 spilled scalar
// registers, 410 -> 404 VALU per tile -- the spills that cost are those of the tile / round state, not of the arguments;
// kept there, not worth carrying to the other passes.
template <class T>
__device__ __forceinline__ T karg_load(uint32_t offset) {
    using cptr = __attribute__((address_space(4))) const char*;
    cptr p = (cptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *reinterpret_cast<__attribute__((address_space(4))) const T*>(p + offset);
}
#define BSK_KARG(ARGS, member) (::bsk::stream::karg_load<decltype(ARGS::member)>((uint32_t)offsetof(ARGS, member)))

// ---------------------------------------------------------------------------
// SWAR byte predicates on a dword -> 4-bit mask (bit b = byte b matches)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) {  // exact: 0x80 in every zero byte
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
__device__ __forceinline__ uint32_t nibble(uint32_t t) {  // 0x80-per-byte mask -> 4 bits
    return ((t >> 7) | (t >> 14) | (t >> 21) | (t >> 28)) & 0xFu;
}
__device__ __forceinline__ uint32_t eq_mask16(const uint4& v, uint32_t rep) {
    return nibble(zero_bytes(v.x ^ rep)) | (nibble(zero_bytes(v.y ^ rep)) << 4) | (nibble(zero_bytes(v.z ^ rep)) << 8) |
           (nibble(zero_bytes(v.w ^ rep)) << 12);
}
// bytes >= thr (unsigned compare, thr in 1..128): k = (0x80 - thr) replicated
__device__ __forceinline__ uint32_t ge_bytes(uint32_t x, uint32_t k) {
    return (((x & 0x7F7F7F7Fu) + k) | x) & 0x80808080u;
}
__device__ __forceinline__ uint32_t ge_mask16(const uint4& v, uint32_t k) {
    return nibble(ge_bytes(v.x, k)) | (nibble(ge_bytes(v.y, k)) << 4) | (nibble(ge_bytes(v.z, k)) << 8) |
           (nibble(ge_bytes(v.w, k)) << 12);
}

// ---- packed flags of a 16-byte piece (dense path) -------------------------------------------------------------
// The 0x80-per-byte flag words of the four dwords are merged WITHOUT being compressed to one bit per byte in byte
// order (that compression was half of the VALU work of `stats -a`): byte k of dword d lands on bit 8k + d, so a piece
// uses the low nibble of every byte of one word and a second predicate fits in the high nibbles.
__device__ __forceinline__ uint32_t pack_flags(uint32_t tx, uint32_t ty, uint32_t tz, uint32_t tw) {
    return (tx >> 7) | (ty >> 6) | (tz >> 5) | (tw >> 4);
}
// "byte != rep" as 0x80 flags plus garbage below (AND-accumulate over several letters, then zero_from_nonzero)
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) { return ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x; }
__device__ __forceinline__ uint32_t zero_from_nonzero(uint32_t nz) { return ~nz & 0x80808080u; }
// packed positions of the bytes j' < j of a piece (j = 4 d + k)
__device__ __forceinline__ uint32_t packed_below(uint32_t j) {
    const uint32_t d = j >> 2, k = j & 3u;
    return (0x01010101u * ((1u << d) - 1u)) | ((0x00010101u & ((1u << (8u * k)) - 1u)) << d);
}
// 16-bit byte-order mask -> packed positions (edge tiles only)
__device__ __forceinline__ uint32_t packed_from_mask16(uint32_t m16) {
    uint32_t r = 0;
    for (uint32_t j = 0; j < 16u; ++j)
        if ((m16 >> j) & 1u) r |= 1u << (8u * (j & 3u) + (j >> 2));
    return r;
}

// 16 bytes at buf[idx..idx+16) with idx % 16 == 0; bytes outside [0, n) read as 0
__device__ __forceinline__ uint4 load16(const uint8_t* __restrict__ buf, uint64_t n, uint64_t idx) {
    if (idx + 16 <= n) return *reinterpret_cast<const uint4*>(buf + idx);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int b = 0; b < 16; ++b)
        if (idx + b < n) w[b >> 2] |= (uint32_t)buf[idx + b] << ((b & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// the same as a non-temporal load -- for the passes that read every byte of the shard once and come back to none
// (sink_tile_nt below)
__device__ __forceinline__ uint4 load16_nt(const uint8_t* __restrict__ buf, uint64_t n, uint64_t idx) {
    if (idx + 16 <= n) {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(buf + idx));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    uint32_t w[4] = {0, 0, 0, 0};
    for (int b = 0; b < 16; ++b)
        if (idx + b < n) w[b >> 2] |= (uint32_t)buf[idx + b] << ((b & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
template <bool NT>
__device__ __forceinline__ uint4 load16_tile(const uint8_t* __restrict__ buf, uint64_t n, uint64_t idx) {
    if constexpr (NT) return load16_nt(buf, n, idx);
    else return load16(buf, n, idx);
}

struct Piece {  // what a lane keeps of one 16-byte piece (dense path; packed flags, see pack_flags)
    uint32_t m_nl_a;  // newlines in the low nibbles, a in the high nibbles   (a = q>=20 for FASTQ, gap for FASTA)
    uint32_t m_b_c;   // b in the low nibbles, c in the high nibbles          (b = q>=30, c = gap; FASTQ -a only)
    uint32_t ex_lo;   // exclusive prefix: nl | a << 16
    uint32_t ex_hi;   // exclusive prefix: b  | c << 16
};

// CAPV: newline events the window holds (CAP; 256 for the sinks that take whole FASTQ records, see RECORDS4 below)
template <bool FASTQ, bool ALL, int CAPV = CAP>
struct Lds {
    static constexpr int CAPW = CAPV;
    static constexpr int NSLOTS = HISTORY + CAPV;
    __attribute__((aligned(16))) uint32_t pos[NSLOTS];
    uint32_t a[(ALL) ? NSLOTS : 1];
    uint32_t b[(ALL && FASTQ) ? NSLOTS : 1];
    uint32_t c[(ALL && FASTQ) ? NSLOTS : 1];
    uint8_t flag[(!FASTQ) ? NSLOTS : 4];
    // FASTQ: the byte that follows the newline when the emitting lane had it in registers
    // (0x100 | byte), 0 = unknown -> the sink probes memory (1 event in 16 on the sparse path)
    __attribute__((aligned(8))) uint16_t nc[(FASTQ) ? NSLOTS : 4];
    // sparse path (ALL == false): the 16-byte pieces that contain a newline, compacted in byte order
    __attribute__((aligned(16))) uint4 sdata[(ALL) ? 1 : WAVE];
    uint16_t stag[(ALL) ? 2 : WAVE];  // piece * 64 + lane of the owner
    // sparse FASTQ path with role counts: role[k] = (lines before the byte after the k-th flagged piece of the tile) & 3,
    // role[0] = the same at the tile start -- what a piece WITHOUT a newline needs to know about itself
    uint8_t role[(FASTQ && !ALL) ? WAVE * NPIECE + 4 : 4];
    // the same path: below[j] = packed_below(j), j = 0..16 (the bytes before byte j of a piece in the layout of pack_flags)
    uint32_t below[(FASTQ && !ALL) ? 18 : 1];
};


// keep the last HISTORY events of (history ++ batch) for the next batch
template <bool FASTQ, bool ALL, int CV>
__device__ __forceinline__ void keep_history(Lds<FASTQ, ALL, CV>& L, uint32_t E) {
    const int lane = threadIdx.x & 63;
    wave_lds_fence();
    uint32_t hp = 0, ha = 0, hb = 0, hc = 0;
    uint8_t hf = 0;
    uint16_t hn = 0;
    if (lane < HISTORY) {
        hp = L.pos[E + lane];
        if constexpr (FASTQ) hn = L.nc[E + lane];
        if constexpr (ALL) ha = L.a[E + lane];
        if constexpr (ALL && FASTQ) { hb = L.b[E + lane]; hc = L.c[E + lane]; }
        if constexpr (!FASTQ) hf = L.flag[E + lane];
    }
    wave_lds_fence();
    if (lane < HISTORY) {
        L.pos[lane] = hp;
        if constexpr (FASTQ) L.nc[lane] = hn;
        if constexpr (ALL) L.a[lane] = ha;
        if constexpr (ALL && FASTQ) { L.b[lane] = hb; L.c[lane] = hc; }
        if constexpr (!FASTQ) L.flag[lane] = hf;
    }
    wave_lds_fence();
}

// the byte after newline event `s` (its absolute index is abs_next); 0 when past the range
template <bool FASTQ, bool ALL, int CV>
__device__ __forceinline__ uint8_t next_char(const Lds<FASTQ, ALL, CV>& L, uint32_t s, uint64_t abs_next, uint64_t re,
                                             const uint8_t* __restrict__ buf) {
    if (abs_next >= re) return 0;
    if constexpr (FASTQ) {
        const uint32_t v = L.nc[s];
        if (v & 0x100u) return (uint8_t)v;
    }
    return buf[abs_next];
}

// absolute byte index of an event from its range-relative position; valid for
// events within +-2^31 bytes of the current tile
__device__ __forceinline__ uint64_t abs_of(uint32_t rel, uint64_t tile_idx, uint32_t tile_rel) {
    return tile_idx + (uint64_t)(int64_t)(int32_t)(rel - tile_rel);
}

// A sink with `static constexpr bool ROLE_COUNTS = true` (FASTQ, sparse path) receives Q20 / Q30 / gap counts in its
// rq20 / rq30 / rgap members (uint32_t per lane, the caller folds them after every range): they are sums over the
// file, so they are counted per 16-byte piece by the ROLE of its bytes
// (line index & 3: 0 header, 1 bases, 2 plus, 3 qualities) instead of as differences of running counters at the events.
template <class S, class = void>
struct sink_role_counts { static constexpr bool value = false; };
template <class S>
struct sink_role_counts<S, decltype((void)S::ROLE_COUNTS)> { static constexpr bool value = S::ROLE_COUNTS; };

// A sink with `static constexpr bool RECORDS4 = true` (FASTQ, sparse path) takes WHOLE RECORDS, one lane per record:
//     sink.records<LDS>(L, R, wb, tile_idx, tile_rel, rs, re, buf)
// R records = events [wb, wb + 4 R) (wb % 4 == 0) in slots [HISTORY, HISTORY + 4 R): lane j reads the five line ends of record
// j with one ds_read_b128 + one ds_read_b32 (and the three "next byte" words with one ds_read_b64) and does the work of the
// four event lanes of batch().  The skeleton DEFERS: the events of the tiles pile up in a window of 256 and the sink runs
// when 64 records are complete -- one sink iteration per ~5 tiles of 150-base reads with all 64 lanes busy, where batch()
// ran once per tile with 52 event lanes of which a quarter (the record ends) did the sink's real work.  Round 4: the
// streaming sinks were at their instruction issue time (HISTORY.md §7), and this is the part of it that scales with
// events instead of bytes.  Events that do not fill a record at the end of a range (a truncated file) go through
// batch(), whose per-event rules and error flags are the reference for both.
template <class S, class = void>
struct sink_records4 { static constexpr bool value = false; };
template <class S>
struct sink_records4<S, decltype((void)S::RECORDS4)> { static constexpr bool value = S::RECORDS4; };

// Non-temporal tile loads: a sink says TILE_NT = true when its pass never comes back to a byte of the shard.  Measured at
// C2 / C5 (scripts/history/r04_nt_loads.sh, r04_nt_more.sh): k_stats 16.42 -> 15.39 ms (0.761 -> 0.812 of the HBM peak),
// k_rmdup_stream 7.45 -> 7.15.  NOT for the sinks that fetch header bytes again (k_names 20.0 -> 24.5 ms with it), k_filter
// (3.17 -> 3.29), k_index / k_subseq_stream (no change), `stats -a` (20.5 -> 20.3 but one spilled register), any FASTA pass
// (`stats -a` at 20 GB 6.75 -> 7.80).
template <class S, class = void>
struct sink_tile_nt { static constexpr bool value = false; };
template <class S>
struct sink_tile_nt<S, decltype((void)S::TILE_NT)> { static constexpr bool value = S::TILE_NT; };

// A RECORDS4 sink with `static constexpr bool HEAD16 = true` is handed the 16 bytes BEHIND every record-end newline while
// they are in LDS -- the marker and the first 15 bytes of the NEXT record's header line (round 6, k_names):
//     sink.head16(s, p, valid)          s = window slot of the record-end event; p = the byte behind the newline inside the
//                                       compacted flagged pieces (L.sdata, LDS: consecutive slots are consecutive 16-byte
//                                       pieces of the tile when they are neighbours in it; an unaligned ds_read_b128 is fine
//                                       on gfx950); valid = what the sink may use of the 16 bytes at p is the text (the header
//                                       line ends inside this piece, or the next slot holds the tile's next piece)
//     sink.shift16(done, keep)          the window moved: slots [done, done + HISTORY + keep) are now [0, HISTORY + keep)
// `seq -n` fetched 1.43 x its input (profiles/r06_ops_traffic.json): one more 128-byte line per record for a 12-byte header
// that had gone through the wave's registers ~5 tiles earlier -- a wave's tiles are long out of L2 by then (28 waves x 8 KiB
// in flight per CU and 32 CUs share 4 MiB).
template <class S, class = void>
struct sink_head16 { static constexpr bool value = false; };
template <class S>
struct sink_head16<S, decltype((void)S::HEAD16)> { static constexpr bool value = S::HEAD16; };

// WHEN the deferred sink runs.  REC_TILE_END = true (default): at the end of a tile once the window is half full -- the 16
// data registers of the tile are dead there, which is what lets k_index / k_names keep 72 registers without a spill.
// false: inside the rounds, when the window is full (64 records of a 256 window) -- what k_stats' default row needs: it
// has LDS for 7 blocks per CU only with the small window, and the compiler fits THAT shape into its 72 registers
// (the tile-end shape: 82, nine of them spilled).  The register allocator decides; tests/test_kernel_resources_cpu.py holds it.
template <class S, class = void>
struct sink_rec_tile_end { static constexpr bool value = true; };
template <class S>
struct sink_rec_tile_end<S, decltype((void)S::REC_TILE_END)> { static constexpr bool value = S::REC_TILE_END; };

// the window after `done` events were consumed: slots [done, done + HISTORY + keep) move to the front (keep = events that
// stay pending; HISTORY + keep <= 64)
template <bool FASTQ, bool ALL, int CV>
__device__ __forceinline__ void shift_window(Lds<FASTQ, ALL, CV>& L, uint32_t done, uint32_t keep) {
    static_assert(FASTQ && !ALL, "deferred windows exist on the sparse FASTQ path only");
    const uint32_t lane = threadIdx.x & 63;
    wave_lds_fence();
    uint32_t hp = 0;
    uint16_t hn = 0;
    const bool mine = lane < (uint32_t)HISTORY + keep;
    if (mine) { hp = L.pos[done + lane]; hn = L.nc[done + lane]; }
    wave_lds_fence();
    if (mine) { L.pos[lane] = hp; L.nc[lane] = hn; }
    wave_lds_fence();
}

// set 0x80 flags of four dwords counted on top of acc: a chain of four v_bcnt_u32_b32 (count + accumulator)
// (written as instructions: the compiler turns the sum of four popcounts into three v_bcnt with a zero accumulator, one
// chained and a v_add3)
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t popc4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t acc = 0u) {
    return bcnt_acc(d, bcnt_acc(c, bcnt_acc(b, bcnt_acc(a, acc))));
}

// ---------------------------------------------------------------------------
// stream one range [rs, re) of the shard through `sink`.
//   sink.err            uint32_t error flags (per lane)
//   Sink::TILE_HOOK     static constexpr bool; true: sink.tile(cur, tile_idx, rs, re, buf) is called once per tile
//                       with the tile's 4 x 16 bytes per lane, before the batches of that tile
//   sink.batch<FASTQ, ALL>(L, E, wb, tile_idx, tile_rel, re, buf)
//       E events sit in LDS slots [HISTORY, HISTORY+E); wb = rank (0-based line
//       index inside the range) of the first; slots [0, HISTORY) hold the
//       previous HISTORY events.
// returns the number of lines (newline events) in the range
// ---------------------------------------------------------------------------
//   skip_from (FASTA, sparse path): the caller knows that [skip_from - 1, re - 1) holds no newline -- the nominal chunks
//       behind the range's own one in which k_prep found no line start (a chromosome on one line).  Tiles that begin at
//       or after skip_from are not read, except the last one (it holds the line's newline at re - 1): what the events
//       need of a line is where it ends, not what is in it.  ~0: read everything.
//   count_resume (FASTA -a, dense path): the gap letters of the 16-byte pieces in [skip_from, count_resume) are NOT counted
//       here -- those bytes belong to nominal chunks without a line start, whose own (otherwise empty) ranges count them
//       (k_stats) and k_stats_stitch adds them when the long line is a sequence line.
template <bool FASTQ, bool ALL, bool DPP, class Sink, int CV>
__device__ __forceinline__ uint32_t stream_range(Lds<FASTQ, ALL, CV>& L, const uint8_t* __restrict__ buf, uint64_t n,
                                                 uint64_t rs, uint64_t re, bool is_last, const PredConsts& P,
                                                 Sink& sink, uint64_t skip_from = ~0ull, uint64_t count_resume = 0) {
    const int lane = threadIdx.x & 63;
    constexpr bool ROLES = FASTQ && !ALL && sink_role_counts<Sink>::value;
    constexpr bool REC4 = FASTQ && !ALL && sink_records4<Sink>::value;  // whole records, deferred (see sink_records4)
    constexpr bool REC_TE = REC4 && sink_rec_tile_end<Sink>::value;     // ... at the end of a tile / when the window is full
    constexpr bool HEAD16 = REC4 && sink_head16<Sink>::value;           // ... and is handed the 16 bytes behind every record end
    constexpr uint32_t CAPW = (uint32_t)CV;
    static_assert(!REC4 || (CV % 4 == 0 && CV >= 256), "a window of whole records with room for a tile behind the ones that wait");
    uint32_t pend_base = 0;  // REC4: rank of the event in slot HISTORY (a multiple of 4)
    // virtual events before the range: a newline at relative position -1
    if (lane < HISTORY) {
        L.pos[lane] = 0xFFFFFFFFu;
        if constexpr (FASTQ) L.nc[lane] = 0;
        if constexpr (ALL) L.a[lane] = 0;
        if constexpr (ALL && FASTQ) { L.b[lane] = 0; L.c[lane] = 0; }
        // FASTA ranges may begin inside a record (on any line start): the virtual event before the range closed a
        // record only if the range begins with a header
        if constexpr (!FASTQ) L.flag[lane] = buf[rs] == '>' ? 1 : 0;
    }
    if constexpr (ROLES) {
        if (lane <= 16) L.below[lane] = lane == 16 ? 0x0F0F0F0Fu : packed_below((uint32_t)lane);
    }
    wave_lds_fence();
    if (lane == 0) {
        const uint8_t c0 = buf[rs];
        if constexpr (FASTQ) { if (c0 != '@') sink.err |= ERR_BAD_HEADER; }
        else { if (rs == 0 && c0 != '>') sink.err |= ERR_BAD_HEADER; }
    }
#if BSK_NL_SGPR
    uint32_t k_ctl;
    asm volatile("s_mov_b32 %0, 0x20202020" : "=s"(k_ctl));
#endif
    if constexpr (!FASTQ) {
        // positions are 32-bit and relative to the range: a range of 2^31 bytes (one line longer than that) cannot be measured
        if (re - rs > 0x7FFFFFFFull) sink.err |= ERR_LINE_TOO_LONG;
    }
    uint32_t line_base = 0;                    // newlines seen so far in this range
    uint32_t run_a = 0, run_b = 0, run_c = 0;  // running counters (mod 2^32)
    uint32_t quiet_tiles = 0;                  // consecutive tiles without a newline

    const uint64_t idx0 = rs & ~(uint64_t)15;
    const uint64_t ntiles = (re - idx0 + TILE - 1) / TILE;

    constexpr bool TILE_NT = sink_tile_nt<Sink>::value && FASTQ && !ALL;
    uint4 cur[NPIECE], nxt[NPIECE];
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) cur[p] = load16_tile<TILE_NT>(buf, n, idx0 + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16);

    for (uint64_t t = 0; t < ntiles; ++t) {
        const uint64_t tile_idx = idx0 + t * TILE;
        const uint32_t tile_rel = (uint32_t)(tile_idx - rs);
        // prefetch the next tile while this one is processed
        if (BSK_PREFETCH && t + 1 < ntiles) {
#pragma unroll
            for (int p = 0; p < NPIECE; ++p)
                nxt[p] = load16_tile<TILE_NT>(buf, n, tile_idx + TILE + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16);
        }
        const bool edge = (tile_idx < rs) || (tile_idx + TILE > re);  // wave-uniform

        // sinks that look at the raw text of a tile (the fused pattern filter, stream_filter.hip) see it here, while
        // it is in registers and before the newline events of the tile are handed over
        if constexpr (Sink::TILE_HOOK) sink.tile(cur, tile_idx, rs, re, buf);

        if constexpr (!ALL) {
            // ---- sparse path -------------------------------------------------------------
            // Only ~1 piece in 5 holds a newline (4 per 317 B).  Every lane runs a cheap exact
            // "does my piece contain one" filter; the flagged pieces are compacted in byte order
            // into LDS and only they pay for exact byte masks, the prefix scan and the events.
            const uint32_t tile_rank_base = line_base;
            uint32_t slot_p[NPIECE];
            bool flagged[NPIECE];
            uint32_t tot_slots = 0;
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) {
                const uint4 v = cur[p];
#if BSK_NL_PREFILTER == 1
                // "any byte below 0x20 in these 16 bytes": (x - 0x20..) & ~x has bit 7 set in every byte < 0x20 (a false
                // positive needs such a byte below it in the same dword).  A superset of "holds a newline" -- text has no
                // other control characters but the odd tab -- at one instruction less per dword than the exact test (no
                // xor with 0x0A..); a piece flagged without a newline simply yields no event (the exact masks decide).
#if BSK_NL_SGPR
                // the constant in a scalar register: a VOP2 with a 32-bit literal is an 8-byte instruction
                const uint32_t h = (((v.x - k_ctl) & ~v.x) | ((v.y - k_ctl) & ~v.y) | ((v.z - k_ctl) & ~v.z) | ((v.w - k_ctl) & ~v.w)) & 0x80808080u;
#else
                const uint32_t h = (((v.x - 0x20202020u) & ~v.x) | ((v.y - 0x20202020u) & ~v.y) |
                                    ((v.z - 0x20202020u) & ~v.z) | ((v.w - 0x20202020u) & ~v.w)) & 0x80808080u;
#endif
#else
                const uint32_t y0 = v.x ^ 0x0A0A0A0Au, y1 = v.y ^ 0x0A0A0A0Au, y2 = v.z ^ 0x0A0A0A0Au,
                               y3 = v.w ^ 0x0A0A0A0Au;
                // (y - 0x01..) & ~y has bit 7 set in every zero byte; a false positive needs a true
                // zero byte below it in the same dword, so "any newline in these 16 bytes" is exact
                const uint32_t h = (((y0 - 0x01010101u) & ~y0) | ((y1 - 0x01010101u) & ~y1) |
                                    ((y2 - 0x01010101u) & ~y2) | ((y3 - 0x01010101u) & ~y3)) & 0x80808080u;
#endif
                // role counts: on an edge tile every piece takes the masked path of the flagged ones
                const bool f = h != 0u || (ROLES && edge);
                const uint64_t bal = __ballot(f);
                flagged[p] = f;
                // for a piece that is not flagged: the number of flagged pieces before it
                slot_p[p] = tot_slots + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                tot_slots += (uint32_t)__popcll(bal);
            }
            if constexpr (ROLES) {
                if (lane == 0) L.role[0] = (uint8_t)(tile_rank_base & 3u);
            }
            for (uint32_t sbase = 0; sbase < tot_slots; sbase += WAVE) {
                // owners publish their flagged pieces of this round
#pragma unroll
                for (int p = 0; p < NPIECE; ++p) {
                    const uint32_t sl = slot_p[p] - sbase;
                    // the byte BEHIND the piece -- the first byte of the next lane's piece (DPP wave_shl:1, every lane takes
                    // part) -- travels in the high byte of the tag: a newline in the last byte of a piece then knows what
                    // follows it without a load.  Round 4: with the sinks deferred those loads (1 event in 16) came 5 tiles
                    // late and missed the caches -- 6 GB of fetches per 100 GB, the whole distance of k_stats to a plain read.
                    // (lane 63 has no neighbour: 1 event in 1 000 still asks memory)
                    const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur[p].x, 0x130, 0xf, 0xf, true) & 0xFFu;
                    if (flagged[p] && sl < (uint32_t)WAVE) {
                        L.sdata[sl] = cur[p];
                        L.stag[sl] = (uint16_t)((uint32_t)(p * WAVE + lane) | (nx << 8));
                    }
                }
                wave_lds_fence();
                const bool have = sbase + (uint32_t)lane < tot_slots;
                uint4 v = make_uint4(0, 0, 0, 0);
                uint32_t tag = 0;
                if (have) { v = L.sdata[lane]; tag = L.stag[lane]; }
                const uint32_t nxb = tag >> 8;  // the byte behind the piece (valid unless lane 63 owns it)
                tag &= 0xFFu;
                const uint32_t off0 = (tag >> 6) * (uint32_t)PIECE_BYTES + (tag & 63u) * 16u;
                uint32_t nl = have ? eq_mask16(v, 0x0A0A0A0Au) : 0u;
                uint32_t vmask = have ? 0xFFFFu : 0u;  // bytes of the piece that belong to the range
                if (edge) {
                    const uint64_t I = tile_idx + off0;
                    int64_t lo = (int64_t)rs - (int64_t)I, hi = (int64_t)re - (int64_t)I;
                    lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
                    hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
                    vmask &= hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
                    nl &= vmask;
                }
                const uint32_t cnt = (uint32_t)__popc(nl);
                const uint32_t incl = wave_incl_scan<DPP>(cnt);
                const uint32_t round_base = line_base;
                const uint32_t rank0 = round_base + incl - cnt;
                line_base += wave_last(incl);
                if constexpr (ROLES) {
                    if (have) L.role[1u + sbase + (uint32_t)lane] = (uint8_t)((rank0 + cnt) & 3u);
                    // pieces with a newline (and every piece of an edge tile): the bytes between its newlines take the
                    // roles rank0, rank0 + 1, ...; cq / cs = the bytes of quality / sequence lines.  Flags and masks stay
                    // in the layout of pack_flags (a byte-order mask costs 31 instructions per predicate, a packed one
                    // 13); the packed "bytes before byte b" come from LDS, both ends of a segment in one read.
                    uint32_t m = nl, r = rank0 & 3u, cq = 0, cs = 0, pl = 0;
                    while (m) {
                        const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
                        m &= m - 1u;
                        const uint32_t seg = L.below[b] & ~pl;
                        pl = L.below[b + 1u];
                        cq |= r == 3u ? seg : 0u;
                        cs |= r == 1u ? seg : 0u;
                        r = (r + 1u) & 3u;
                    }
                    const uint32_t seg = 0x0F0F0F0Fu & ~pl;
                    cq |= r == 3u ? seg : 0u;
                    cs |= r == 1u ? seg : 0u;
                    if (edge) {
                        const uint32_t vp = packed_from_mask16(vmask);
                        cq &= vp;
                        cs &= vp;
                    } else if (!have) {
                        cq = cs = 0;
                    }
                    constexpr uint32_t HI = 0x80808080u;
                    const uint32_t a0 = v.x & 0x7F7F7F7Fu, a1 = v.y & 0x7F7F7F7Fu, a2 = v.z & 0x7F7F7F7Fu,
                                   a3 = v.w & 0x7F7F7F7Fu;
                    sink.rq20 += (uint32_t)__popc(pack_flags(((a0 + P.k20) | v.x) & HI, ((a1 + P.k20) | v.y) & HI,
                                                             ((a2 + P.k20) | v.z) & HI, ((a3 + P.k20) | v.w) & HI) & cq);
                    sink.rq30 += (uint32_t)__popc(pack_flags(((a0 + P.k30) | v.x) & HI, ((a1 + P.k30) | v.y) & HI,
                                                             ((a2 + P.k30) | v.z) & HI, ((a3 + P.k30) | v.w) & HI) & cq);
                    uint32_t low = cs;  // sequence bytes that may be gap letters
                    if (P.kgap != 0xFFFFFFFFu)
                        low &= ~pack_flags(((a0 + P.kgap) | v.x) & HI, ((a1 + P.kgap) | v.y) & HI, ((a2 + P.kgap) | v.z) & HI,
                                           ((a3 + P.kgap) | v.w) & HI);
                    if (__ballot(low != 0u)) {
#pragma nounroll
                        for (int k = 0; k < P.ngap; ++k) {
                            const uint32_t rep = P.gap_rep[k];
                            sink.rgap += (uint32_t)__popc(pack_flags(zero_bytes(v.x ^ rep), zero_bytes(v.y ^ rep),
                                                                     zero_bytes(v.z ^ rep), zero_bytes(v.w ^ rep)) & cs);
                        }
                    }
                }
                // HEAD16: is the piece in the next slot the tile's next piece?  (lane l + 1's offset over DPP wave_shl:1)
                bool next_adjacent = false;
                if constexpr (HEAD16) {
                    const uint32_t offn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(have ? off0 + 1u : 0u), 0x130, 0xf, 0xf, true);
                    next_adjacent = have && lane != 63 && offn == off0 + 16u + 1u;
                }
                // this lane's newlines -> events in the window whose first slot is the event of rank `wb`
                auto emit_events = [&](uint32_t wb) {
                    uint32_t m = nl, k = 0;
                    while (m) {
                        const uint32_t bpos = (uint32_t)__ffs((int)m) - 1u;
                        m &= m - 1u;
                        const uint32_t w = rank0 + k - wb;
                        ++k;
                        if (w < CAPW) {
                            const uint32_t s = HISTORY + w;
                            const uint32_t off = off0 + bpos;
                            L.pos[s] = tile_rel + off;
                            if constexpr (HEAD16) {
                                if (((rank0 + k - 1u) & 3u) == 3u) {  // a record ends here: what follows is the next record's header
                                    sink.head16(s, reinterpret_cast<const uint8_t*>(&L.sdata[0]) + (uint32_t)lane * 16u + bpos + 1u,
                                                m != 0u || next_adjacent);
                                }
                            }
                            // the byte after the newline sits in the same 16 bytes 15 times out of 16
                            uint32_t nc16 = 0;
                            if (bpos < 15u) {
                                const uint32_t d = (bpos + 1u) >> 2, sh8 = ((bpos + 1u) & 3u) * 8u;
                                const uint32_t wd = d == 0 ? v.x : (d == 1 ? v.y : (d == 2 ? v.z : v.w));
                                nc16 = 0x100u | ((wd >> sh8) & 0xFFu);
                            } else if ((tag & 63u) != 63u) {
                                nc16 = 0x100u | nxb;  // ... or is the first byte of the neighbour's piece
                            }
                            if constexpr (FASTQ) {
                                L.nc[s] = (uint16_t)nc16;
                            } else {
                                // the line closes a record when a header follows, or at the end of the shard
                                const uint64_t an = tile_idx + off + 1;
                                const uint8_t nc = nc16 ? (uint8_t)nc16 : (an < n ? buf[an] : (uint8_t)0);
                                L.flag[s] = (nc == '>' || (an >= re && is_last) || an >= n) ? 1 : 0;
                            }
                        }
                    }
                };
                if constexpr (REC4) {
                    // deferred: the events join the window and wait for the end of a tile (below), which leaves room for
                    // CAPW - 256 more.  A tile with more newlines than that (lines of a few bytes) sends the full window
                    // through the per-event rules right here -- a whole number of records, so the window keeps beginning on one.
                    for (;;) {
                        emit_events(pend_base);
                        if (line_base - pend_base <= CAPW) break;  // (wave-uniform) everything of this round is in
                        wave_lds_fence();
                        // (batch() takes positions relative to the tile it is handed and expects none before it: the window
                        // holds events of earlier tiles, so it is handed the range itself -- tile (rs, 0))
                        if constexpr (REC_TE) sink.template batch<FASTQ, ALL>(L, CAPW, pend_base, rs, 0u, re, buf);
                        else sink.template records<Lds<FASTQ, ALL, CV>>(L, CAPW / 4u, pend_base, tile_idx, tile_rel, rs, re, buf);
                        shift_window(L, CAPW, 0u);
                        if constexpr (HEAD16) sink.shift16(CAPW, 0u);
                        pend_base += CAPW;
                    }
                } else {
                    for (uint32_t wb = round_base; wb < line_base; wb += CAPW) {
                        emit_events(wb);
                        wave_lds_fence();
                        const uint32_t E = (line_base - wb) < CAPW ? (line_base - wb) : CAPW;
                        sink.template batch<FASTQ, ALL>(L, E, wb, tile_idx, tile_rel, re, buf);
                        keep_history<FASTQ, ALL>(L, E);
                    }
                }
                wave_lds_fence();  // the next round overwrites sdata / stag
            }
            if constexpr (ROLES) {
                if (!edge) {
                    // pieces without a newline: all 16 bytes have ONE role, the one published by the flagged piece before
                    wave_lds_fence();
#pragma unroll
                    for (int p = 0; p < NPIECE; ++p) {
                        const uint4 v = cur[p];
                        const uint32_t r = L.role[slot_p[p]];
                        const bool isq = !flagged[p] && r == 3u, iss = !flagged[p] && r == 1u;
                        const uint32_t a0 = v.x & 0x7F7F7F7Fu, a1 = v.y & 0x7F7F7F7Fu, a2 = v.z & 0x7F7F7F7Fu,
                                       a3 = v.w & 0x7F7F7F7Fu;
                        constexpr uint32_t HI = 0x80808080u;
                        // one family of "byte >= t" counts with a lane-selected threshold: quality bytes >= Q20 on a
                        // quality line; on a sequence line bytes above the largest gap letter (bases are letters, so
                        // 16 of 16 settle the gap count without looking at the letters)
                        const uint32_t ka = r == 1u ? P.kgap : P.k20;
                        const uint32_t ca = popc4(((a0 + ka) | v.x) & HI, ((a1 + ka) | v.y) & HI, ((a2 + ka) | v.z) & HI,
                                                  ((a3 + ka) | v.w) & HI);
                        // Q30: the flags of lanes that are not on a quality line are masked away, the counts chain into
                        // the accumulator
                        const uint32_t mq = isq ? HI : 0u;
                        sink.rq30 = popc4(((a0 + P.k30) | v.x) & mq, ((a1 + P.k30) | v.y) & mq, ((a2 + P.k30) | v.z) & mq,
                                          ((a3 + P.k30) | v.w) & mq, sink.rq30);
                        sink.rq20 += isq ? ca : 0u;
                        if (__ballot(iss && (ca != 16u || P.kgap == 0xFFFFFFFFu))) {
                            // gap letters are distinct: a byte equals at most one of them, the counts add up
                            uint32_t cg = 0;
#pragma nounroll
                            for (int k = 0; k < P.ngap; ++k) {
                                const uint32_t rep = P.gap_rep[k];
                                cg += popc4(zero_bytes(v.x ^ rep), zero_bytes(v.y ^ rep), zero_bytes(v.z ^ rep),
                                            zero_bytes(v.w ^ rep));
                            }
                            sink.rgap += iss ? cg : 0u;
                        }
                    }
                }
            }
            if (line_base == tile_rank_base) {
                // lines longer than 2^31 bytes cannot be measured with 32-bit relative positions
                if (++quiet_tiles >= (1u << 31) / TILE) sink.err |= ERR_LINE_TOO_LONG;
            } else {
                quiet_tiles = 0;
            }
            if constexpr (REC_TE) {
                // the whole records of the window, at the END of a tile: its 16 data registers are dead here (inside the
                // rounds the same call cost k_stats two spilled registers at 7 waves per SIMD).  Run once 256 events are
                // pending -- 64 .. 128 records per call, every fifth tile of 150-base reads -- and at the end of the range,
                // where the virtual newline of a file that stops inside its last quality line completes that record first.
                const bool last_tile = t + 1 == ntiles;
                uint32_t pending = line_base - pend_base;
                if (last_tile && is_last && (line_base & 3u) == 3u && pending < CAPW) {
                    if (lane == 0) { L.pos[HISTORY + pending] = (uint32_t)(re - rs); L.nc[HISTORY + pending] = 0; }
                    pending += 1;
                    line_base += 1;
                }
                // (a window of 512: 256 wait, 256 are room for the next tile; a smaller window -- k_stats at 7 waves per SIMD has
                // LDS for 320 -- keeps room for 128)
                constexpr uint32_t FLUSH_AT = CV >= 512 ? 256u : (uint32_t)CV - 128u;
                if (pending >= FLUSH_AT || last_tile) {  // (wave-uniform)
                    const uint32_t R = pending >> 2;
                    wave_lds_fence();
                    if (R) sink.template records<Lds<FASTQ, ALL, CV>>(L, R, pend_base, tile_idx, tile_rel, rs, re, buf);
                    shift_window(L, 4u * R, pending & 3u);
                    if constexpr (HEAD16) sink.shift16(4u * R, pending & 3u);
                    pend_base += 4u * R;
                }
            }
        } else {
            Piece pc[NPIECE];
            uint32_t base_nl[NPIECE], base_a[NPIECE], base_b[NPIECE], base_c[NPIECE];
            const uint32_t tile_rank_base = line_base;
    #pragma unroll
            for (int p = 0; p < NPIECE; ++p) {
                const uint4 v = cur[p];
                uint32_t nl = pack_flags(zero_bytes(v.x ^ 0x0A0A0A0Au), zero_bytes(v.y ^ 0x0A0A0A0Au),
                                         zero_bytes(v.z ^ 0x0A0A0A0Au), zero_bytes(v.w ^ 0x0A0A0A0Au));
                uint32_t ma = 0, mb = 0, mc = 0;
                if constexpr (ALL) {
                    // gap letters: AND the "differs" words of all letters, one pack for the set
                    uint32_t n0 = ~0u, n1 = ~0u, n2 = ~0u, n3 = ~0u;
    #pragma unroll
                    for (int k = 0; k < MAX_GAP_LETTERS; ++k)
                        if (k < P.ngap) {
                            const uint32_t rep = P.gap_rep[k];
                            n0 &= nonzero_bytes(v.x ^ rep); n1 &= nonzero_bytes(v.y ^ rep);
                            n2 &= nonzero_bytes(v.z ^ rep); n3 &= nonzero_bytes(v.w ^ rep);
                        }
                    const uint32_t g = pack_flags(zero_from_nonzero(n0), zero_from_nonzero(n1), zero_from_nonzero(n2),
                                                  zero_from_nonzero(n3));
                    if constexpr (FASTQ) {
                        ma = pack_flags(ge_bytes(v.x, P.k20), ge_bytes(v.y, P.k20), ge_bytes(v.z, P.k20), ge_bytes(v.w, P.k20));
                        mb = pack_flags(ge_bytes(v.x, P.k30), ge_bytes(v.y, P.k30), ge_bytes(v.z, P.k30), ge_bytes(v.w, P.k30));
                        mc = g;
                    } else {
                        // (pieces of the chunks that a long line covers are counted by those chunks' ranges)
                        const uint64_t I = tile_idx + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16;
                        ma = (I >= skip_from && I < count_resume) ? 0u : g;
                    }
                }
                if (edge) {
                    const uint64_t I = tile_idx + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16;
                    int64_t lo = (int64_t)rs - (int64_t)I, hi = (int64_t)re - (int64_t)I;
                    lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
                    hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
                    const uint32_t valid = packed_from_mask16(hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u);
                    nl &= valid; ma &= valid; mb &= valid; mc &= valid;
                }
                pc[p].m_nl_a = nl | (ma << 4);
                pc[p].m_b_c = mb | (mc << 4);
                const uint32_t lo_cnt = (uint32_t)__popc(nl) | ((uint32_t)__popc(ma) << 16);
                const uint32_t incl_lo = wave_incl_scan<DPP>(lo_cnt);
                pc[p].ex_lo = incl_lo - lo_cnt;
                const uint32_t tot_lo = wave_last(incl_lo);
                base_nl[p] = line_base;
                base_a[p] = run_a;
                line_base += tot_lo & 0xFFFFu;
                run_a += tot_lo >> 16;
                if constexpr (ALL && FASTQ) {
                    const uint32_t hi_cnt = (uint32_t)__popc(mb) | ((uint32_t)__popc(mc) << 16);
                    const uint32_t incl_hi = wave_incl_scan<DPP>(hi_cnt);
                    pc[p].ex_hi = incl_hi - hi_cnt;
                    const uint32_t tot_hi = wave_last(incl_hi);
                    base_b[p] = run_b;
                    base_c[p] = run_c;
                    run_b += tot_hi & 0xFFFFu;
                    run_c += tot_hi >> 16;
                } else {
                    pc[p].ex_hi = 0;
                    base_b[p] = 0;
                    base_c[p] = 0;
                }
            }
            const uint32_t tile_events = line_base - tile_rank_base;
            if (tile_events == 0) {
                // lines longer than 2^31 bytes cannot be measured with 32-bit relative positions
                if (++quiet_tiles >= (1u << 31) / TILE) sink.err |= ERR_LINE_TOO_LONG;
            } else {
                quiet_tiles = 0;
            }

            // events of this tile, CAP at a time (one batch for ordinary data).  The newline masks of
            // the pieces are merged into one 64-bit word so that ONE loop visits every newline of the
            // lane (~3 iterations per tile for 150 bp reads instead of ~2 per piece).
            // word h of a merged mask holds pieces 2h (low nibbles) and 2h + 1 (high nibbles)
            static_assert(NPIECE == 4, "the dense path merges exactly four pieces");
            constexpr uint32_t LOW = 0x0F0F0F0Fu;
            const uint32_t nl_w[2] = {(pc[0].m_nl_a & LOW) | ((pc[1].m_nl_a & LOW) << 4), (pc[2].m_nl_a & LOW) | ((pc[3].m_nl_a & LOW) << 4)};
            uint32_t a_w[2] = {0, 0}, b_w[2] = {0, 0}, c_w[2] = {0, 0};
            if constexpr (ALL) {
                a_w[0] = ((pc[0].m_nl_a >> 4) & LOW) | (pc[1].m_nl_a & ~LOW);
                a_w[1] = ((pc[2].m_nl_a >> 4) & LOW) | (pc[3].m_nl_a & ~LOW);
            }
            if constexpr (ALL && FASTQ) {
                b_w[0] = (pc[0].m_b_c & LOW) | ((pc[1].m_b_c & LOW) << 4);
                b_w[1] = (pc[2].m_b_c & LOW) | ((pc[3].m_b_c & LOW) << 4);
                c_w[0] = ((pc[0].m_b_c >> 4) & LOW) | (pc[1].m_b_c & ~LOW);
                c_w[1] = ((pc[2].m_b_c >> 4) & LOW) | (pc[3].m_b_c & ~LOW);
            }
            const uint64_t m64 = (uint64_t)nl_w[0] | ((uint64_t)nl_w[1] << 32);
            for (uint32_t wb = tile_rank_base; wb < line_base; wb += CAPW) {
                uint64_t m = m64;
                while (m) {
                    // (the newlines of a lane are visited in bit order, not byte order: every event computes its own rank)
                    const uint32_t q = (uint32_t)__ffsll((long long)m) - 1u;
                    m &= m - 1ull;
                    const uint32_t h = q >> 5, rbit = q & 31u;
                    const uint32_t psel = (rbit >> 2) & 1u;
                    const uint32_t p = 2u * h + psel;
                    const uint32_t bpos = 4u * (rbit & 3u) + (rbit >> 3);  // byte of the piece
                    const uint32_t below = packed_below(bpos) << (4u * psel);
                    // per-piece values selected by p (static unrolled compare chain keeps them in registers)
                    uint32_t r0 = base_nl[0] + (pc[0].ex_lo & 0xFFFFu);
                    uint32_t sa = base_a[0] + (pc[0].ex_lo >> 16), sb = base_b[0] + (pc[0].ex_hi & 0xFFFFu),
                             sc = base_c[0] + (pc[0].ex_hi >> 16);
    #pragma unroll
                    for (int pp = 1; pp < NPIECE; ++pp) {
                        if (p == (uint32_t)pp) {
                            r0 = base_nl[pp] + (pc[pp].ex_lo & 0xFFFFu);
                            if constexpr (ALL) sa = base_a[pp] + (pc[pp].ex_lo >> 16);
                            if constexpr (ALL && FASTQ) {
                                sb = base_b[pp] + (pc[pp].ex_hi & 0xFFFFu);
                                sc = base_c[pp] + (pc[pp].ex_hi >> 16);
                            }
                        }
                    }
                    const uint32_t rank = r0 + (uint32_t)__popc((h ? nl_w[1] : nl_w[0]) & below);
                    const uint32_t w = rank - wb;
                    if (w < CAPW) {
                        const uint32_t s = HISTORY + w;
                        const uint32_t off = p * (uint32_t)PIECE_BYTES + (uint32_t)lane * 16u + bpos;
                        L.pos[s] = tile_rel + off;
                        if constexpr (FASTQ) L.nc[s] = 0;  // dense path: the sink probes memory
                        if constexpr (ALL) L.a[s] = sa + (uint32_t)__popc((h ? a_w[1] : a_w[0]) & below);
                        if constexpr (ALL && FASTQ) {
                            L.b[s] = sb + (uint32_t)__popc((h ? b_w[1] : b_w[0]) & below);
                            L.c[s] = sc + (uint32_t)__popc((h ? c_w[1] : c_w[0]) & below);
                        }
                        if constexpr (!FASTQ) {
                            const uint64_t an = tile_idx + off + 1;  // byte after the newline
                            L.flag[s] = (an >= n || buf[an] == '>' || (an >= re && is_last)) ? 1 : 0;
                        }
                    }
                }
                wave_lds_fence();
                const uint32_t E = (line_base - wb) < CAPW ? (line_base - wb) : CAPW;
                sink.template batch<FASTQ, ALL>(L, E, wb, tile_idx, tile_rel, re, buf);
                keep_history<FASTQ, ALL>(L, E);
            }
        }
        if (t + 1 < ntiles) {
            if constexpr (!FASTQ && !BSK_PREFETCH) {
                // the newline-free stretch of a line that is longer than the range's nominal chunk: on to its last tile
                // (-a: the bytes from count_resume on are this range's to count -- the tile that holds them is the target)
                uint64_t tgt = ntiles - 1;
                if constexpr (ALL) {
                    if (count_resume > idx0 && (count_resume - idx0) / TILE < tgt) tgt = (count_resume - idx0) / TILE;
                }
                if (tile_idx + TILE >= skip_from && t + 1 < tgt) {
                    t = tgt - 1;
                    quiet_tiles = 0;
                    const uint64_t tgt_idx = idx0 + tgt * TILE;
#pragma unroll
                    for (int p = 0; p < NPIECE; ++p) cur[p] = load16_tile<TILE_NT>(buf, n, tgt_idx + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16);
                    continue;
                }
            }
#pragma unroll
            for (int p = 0; p < NPIECE; ++p)
                cur[p] = BSK_PREFETCH ? nxt[p]
                                      : load16_tile<TILE_NT>(buf, n, tile_idx + TILE + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16);
        }
    }

    // end of range --------------------------------------------------------
    const uint32_t end_rel = (uint32_t)(re - rs);
    const uint64_t end_tile = re & ~(uint64_t)(TILE - 1);
    bool virt = false;
    if constexpr (FASTQ) {
        // EOF inside a quality line (no final '\n', or an empty last quality line)
        virt = is_last && (line_base & 3u) == 3u;
    } else {
        virt = is_last && buf[re - 1] != '\n';
    }
    if constexpr (REC4) {
        if constexpr (!REC_TE) {
            // (sinks that run when their window is full) what is still in the window: the virtual newline of a file that
            // stops inside its last quality line completes that record, then the whole records go to the sink
            uint32_t pending = line_base - pend_base;
            if (virt && pending < CAPW) {
                if (lane == 0) { L.pos[HISTORY + pending] = end_rel; L.nc[HISTORY + pending] = 0; }
                pending += 1;
                line_base += 1;
            }
            wave_lds_fence();
            const uint32_t R = pending >> 2;
            if (R) sink.template records<Lds<FASTQ, ALL, CV>>(L, R, pend_base, end_tile, (uint32_t)(end_tile - rs), rs, re, buf);
            shift_window(L, 4u * R, pending & 3u);
            if constexpr (HEAD16) sink.shift16(4u * R, pending & 3u);
            pend_base += 4u * R;
        }
        // the whole records have gone to the sink (tile-end sinks: at the end of the last tile, virtual newline included);
        // what is left are the events of an incomplete last record -- a truncated file, a range that does not end on a
        // record: batch()'s per-event rules and error flags
        const uint32_t left = line_base - pend_base;
        if (left) {
            wave_lds_fence();
            sink.template batch<FASTQ, ALL>(L, left, pend_base, rs, 0u, re, buf);  // (events of earlier tiles: tile (rs, 0))
        }
        (void)virt; (void)end_rel; (void)end_tile;
    } else if (virt) {
        if (lane == 0) {
            L.pos[HISTORY] = end_rel;
            if constexpr (FASTQ) L.nc[HISTORY] = 0;
            if constexpr (ALL) L.a[HISTORY] = run_a;
            if constexpr (ALL && FASTQ) { L.b[HISTORY] = run_b; L.c[HISTORY] = run_c; }
            if constexpr (!FASTQ) L.flag[HISTORY] = 1;
        }
        wave_lds_fence();
        // place the pseudo tile so that (pos - tile_rel) stays small.  The byte after the
        // virtual newline is at re + 1 > re, so no sink probes memory for it.
        const uint32_t vt_rel = (uint32_t)(end_tile - rs);
        sink.template batch<FASTQ, ALL>(L, 1u, line_base, end_tile, vt_rel, re, buf);
        keep_history<FASTQ, ALL>(L, 1u);
        line_base += 1;
    }
    if constexpr (FASTQ) {
        if ((line_base & 3u) != 0u) sink.err |= is_last ? ERR_TRUNCATED : ERR_ANCHOR;
    }
    return line_base;
}

}  // namespace stream
}  // namespace bsk
