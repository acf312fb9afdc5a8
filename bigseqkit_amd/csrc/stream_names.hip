// ============================================================================
// stream_names.hip -- `seq -n` (and `seq -n -i`) on FASTQ inside the streaming pass.
//
// SeqTransform.Call with only Name set prints head (or ID) + "\n" per record
// (/root/reference/bigseqkit-lib/seq.go:143-175): nothing of the sequence is needed
// beyond what SeqParser.Read checks (helper.go:252-311: line roles, len(seq) ==
// len(qual)).  The record-end event of the skeleton knows the header line, so the
// names leave from the pass itself -- no 24 B/record table, no size / scan / emit.
// Every range writes into its own slice of a scratch buffer (sized from the header
// density of the shard head); k_names_compact gathers the slices in range order
// (= file order).  A slice that overflows raises ERR_CAPACITY and the caller takes
// the record-table path.
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

#include "anchor.hpp"
#include "stream_core_dev.hpp"
#include "stream_names.hpp"
#include "text_dev.hpp"

#ifndef BSK_NAMES_EXP
#define BSK_NAMES_EXP 0
#endif
#ifndef BSK_NAMES_WINDOW
#define BSK_NAMES_WINDOW 320  // events the deferred window holds.  Round 4 (header bytes loaded again per record): 384 / tile end
                              // 20.25 ms, 256 / window full 20.38, 512 / tile end 22.07 at C2 (scripts/history/r04_names.sh: the
                              // later the sink ran, the colder the header lines it loaded).  Round 6 (header bytes from LDS):
                              // 256 .. 384 are one figure, 18.7 .. 20.7 ms from box to box (scripts/r06_ab4e.sh, r06_ab4f.sh);
                              // 320 is the least LDS that still hands the sink 48 records at a time
#endif
#ifndef BSK_NAMES_TE
#define BSK_NAMES_TE 1               // the sink runs at the end of a tile (1) / when the window is full (0)
#endif
#ifndef BSK_NAMES_HEAD16
#define BSK_NAMES_HEAD16 1           // header bytes from the pass's LDS (stream_core_dev.hpp sink_head16) instead of a load per record:
                                     // FETCH_SIZE x 2 of the pass 144.8 -> 103.1 GB per 100 GB (scripts/r06_ab4.sh)
#endif
#ifndef BSK_NAMES_NT
#define BSK_NAMES_NT 0               // non-temporal tile loads: no gain once the header bytes come from LDS, and a wider spread
                                     // (19.3 .. 21.2 ms against 18.7 .. 19.6, scripts/r06_ab4e.sh)
#endif
#ifndef BSK_NAMES_DIAG
#define BSK_NAMES_DIAG 0             // experiments (wrong results): 1 no scan / stores, 2 no sink at all, 3 no stores
#endif
#ifndef BSK_NAMES_WAVES
#define BSK_NAMES_WAVES 7
#endif

namespace bsk {

namespace {

using namespace stream;

template <bool DPP>
struct NamesSink {
    static constexpr bool TILE_HOOK = false;
    static constexpr bool RECORDS4 = true;  // whole records, 64 at a time (records() below)
    static constexpr bool REC_TILE_END = BSK_NAMES_TE != 0;
    static constexpr bool TILE_NT = BSK_NAMES_NT != 0 && BSK_NAMES_HEAD16 != 0;
    int only_id = 0, id_mode = 0;  // (NamesDev)
    uint32_t slice_cap = 0;        // bytes of a slice (< 2^32)
    uint8_t* slice = nullptr;      // this range's output slice
    uint32_t cursor = 0;       // bytes written to it so far (wave-uniform)
    uint32_t nrec = 0;         // records seen in this range (wave-uniform)
    uint32_t err = 0;
    const uint8_t* lim = nullptr;  // one past the last byte of the shard
    // HEAD16: entry j = the 16 bytes behind the newline in window slot HISTORY + 4 j - 1 -- '@' and the first 15 bytes of the
    // header of record j of the window (entry 0: the record whose header began before the window did); zeros = not known
    static constexpr bool HEAD16 = BSK_NAMES_HEAD16 != 0;
    uint4* h16 = nullptr;

    __device__ __forceinline__ void head16(uint32_t s, const uint8_t* p, bool valid) {
        uint4* e = &h16[(s - (uint32_t)HISTORY + 1u) >> 2];
        if (valid) {
            uint4 v;
            __builtin_memcpy(&v, p, 16);
            *e = v;
        } else {
            e->x = 0;  // (no '@' in its first byte: not known)
        }
    }
    __device__ __forceinline__ void shift16(uint32_t done, uint32_t) {
        // (the events that stay pending are the first three of a record at most: only the header of THAT record moves)
        uint4 v = make_uint4(0, 0, 0, 0);
        const bool mine = (threadIdx.x & 63) == 0;
        if (mine) v = h16[done >> 2];
        wave_lds_fence();
        if (mine) h16[0] = v;
        wave_lds_fence();
    }

    __device__ __forceinline__ void begin_range(uint8_t* slice_of_range) {
        slice = slice_of_range;
        cursor = 0;
        nrec = 0;
        if constexpr (HEAD16) {
            if ((threadIdx.x & 63) == 0) h16[0] = make_uint4(0, 0, 0, 0);  // (the range's first header has no record end before it)
            wave_lds_fence();
        }
    }

    // exactly olen <= 16 bytes of w to dst: 16 / 8 / 4 / 2 / 1-byte stores (neighbouring lanes own the bytes around them).
    // (Four dwords by value and selects, not an array: indexed by "dwords consumed" the array went to scratch memory -- a
    // 16-byte scratch store and two scratch loads per record, most of the 10.4 GB k_names wrote for 3.8 GB of names at C2.)
    __device__ __forceinline__ void store_small(uint8_t* dst, uint4 w, uint32_t olen) {
        if (olen & 16u) { __builtin_memcpy(dst, &w, 16); return; }
        uint32_t a = w.x, b = w.y, c = w.z, d = w.w;  // what is left to write begins at a
        if (olen & 8u) {
            const uint2 v = make_uint2(a, b);
            __builtin_memcpy(dst, &v, 8);
            dst += 8;
            a = c; b = d;
        }
        if (olen & 4u) {
            __builtin_memcpy(dst, &a, 4);
            dst += 4;
            a = b;
        }
        if (olen & 2u) {
            const uint16_t h2 = (uint16_t)a;
            __builtin_memcpy(dst, &h2, 2);
            dst += 2;
            a >>= 16;
        }
        if (olen & 1u) dst[0] = (uint8_t)a;
    }
    // byte m (< 16) of w becomes '\n'
    __device__ __forceinline__ uint4 newline_at(uint4 w, uint32_t m) {
        const uint32_t sh = (m & 3u) * 8u, d = m >> 2;
        const uint32_t keep = ~(0xFFu << sh), nl = 0x0Au << sh;
        w.x = d == 0u ? (w.x & keep) | nl : w.x;
        w.y = d == 1u ? (w.y & keep) | nl : w.y;
        w.z = d == 2u ? (w.z & keep) | nl : w.z;
        w.w = d == 3u ? (w.w & keep) | nl : w.w;
        return w;
    }

    // olen = m + 1 bytes of a name (m from src, then '\n') to dst
    __device__ __forceinline__ void copy_name(uint8_t* dst, const uint8_t* src, uint32_t m, uint32_t olen) {
        if (olen <= 16u && src + 16 <= lim) {
            // one 16-byte load; the '\n' is put at byte m in registers (it IS byte m of the text for a whole head)
            uint4 w;
            __builtin_memcpy(&w, src, 16);
            if (only_id) w = newline_at(w, m);
            store_small(dst, w, olen);
        } else {
            uint32_t i = 0;
            for (; i + 16u <= m; i += 16u) {
                uint4 v;
                __builtin_memcpy(&v, src + i, 16);
                __builtin_memcpy(dst + i, &v, 16);
            }
            if (m & 8u) {
                uint2 v;
                __builtin_memcpy(&v, src + i, 8);
                __builtin_memcpy(dst + i, &v, 8);
                i += 8u;
            }
            if (m & 4u) {
                uint32_t v;
                __builtin_memcpy(&v, src + i, 4);
                __builtin_memcpy(dst + i, &v, 4);
                i += 4u;
            }
            if (m & 2u) {
                uint16_t v;
                __builtin_memcpy(&v, src + i, 2);
                __builtin_memcpy(dst + i, &v, 2);
                i += 2u;
            }
            if (m & 1u) { dst[i] = src[i]; }
            dst[m] = (uint8_t)'\n';
        }
    }

    // whole records, lane j = record j of the window (stream_core_dev.hpp sink_records4): the rules of batch() for the four
    // events of the record, then ONE scan over 64 name lengths and 64 names written back to back -- where batch() ran
    // the scan and the copy once per tile for the 13 record ends among its 52 event lanes
    template <class LDS>
    __device__ __forceinline__ void records(LDS& L, uint32_t R, uint32_t wb, uint64_t tile_idx, uint32_t tile_rel, uint64_t rs,
                                            uint64_t re, const uint8_t* __restrict__ buf) {
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t end_rel = (uint32_t)(re - rs);
        auto next_of = [&](uint32_t v16, uint32_t p) -> uint32_t {
            if (p + 1u >= end_rel) return 0u;
            if (v16 & 0x100u) return v16 & 0xFFu;
            return buf[rs + p + 1u];
        };
#if BSK_NAMES_DIAG == 2
        nrec += R; return;
#endif
        for (uint32_t r0 = 0; r0 < R; r0 += WAVE) {
            const uint32_t j = r0 + lane;
            const bool on = j < R;
            const uint32_t s = HISTORY + 4u * (on ? j : 0u);
            const uint8_t* src = nullptr;
            uint32_t m = 0, olen = 0;
            if (on) {
                const uint32_t p0 = L.pos[s - 1u], ph = L.pos[s], pb = L.pos[s + 1u];
                if (next_of(L.nc[s], ph) == '+') err |= ERR_BAD_PLUS;
                if (next_of(L.nc[s + 1u], pb) != '+') err |= ERR_BAD_PLUS;
                const uint32_t pp = L.pos[s + 2u], pq = L.pos[s + 3u];
                if (pq - pp != pb - ph) err |= ERR_LEN_MISMATCH;
                if (pq + 1u < end_rel && next_of(L.nc[s + 3u], pq) != '@') err |= ERR_BAD_HEADER;
                const uint32_t lh = ph - p0 - 1u;
                const uint8_t* h = buf + rs + (uint64_t)(p0 + 1u) + 1u;  // the header without its marker (p0 + 1 wraps to 0 at the range start)
                m = lh ? lh - 1u : 0u;
                src = h;
                if (only_id) {
                    uint32_t off = 0;
                    m = id_span_of(h, m, id_mode, &off, lim);
                    src = h + off;
                }
                olen = m + 1u;
            }
            // HEAD16: a whole head of up to 15 bytes sits behind its '@' in the 16 bytes the pass kept in LDS
            bool in_lds = false;
            uint4 hw = make_uint4(0, 0, 0, 0);
            if constexpr (HEAD16) {
                if (on && !only_id && olen <= 16u) {
                    const uint4 hd = h16[j];
                    if ((hd.x & 0xFFu) == (uint32_t)'@') {
                        in_lds = true;
                        hw = newline_at(make_uint4(__builtin_amdgcn_alignbyte(hd.y, hd.x, 1), __builtin_amdgcn_alignbyte(hd.z, hd.y, 1),
                                                   __builtin_amdgcn_alignbyte(hd.w, hd.z, 1), hd.w >> 8), m);
                    }
                }
            }
#if BSK_NAMES_DIAG == 1
            err |= (olen + (uint32_t)(uintptr_t)src) == 0x12345678u; nrec += (uint32_t)__popcll(__ballot(on)); continue;
#endif
            const uint32_t incl = wave_incl_scan<DPP>(olen);
            const uint32_t tot = wave_last(incl);
#if BSK_NAMES_DIAG == 3
            cursor += tot; err |= (hw.x ^ hw.y ^ hw.z ^ hw.w) == 0x12345678u; nrec += (uint32_t)__popcll(__ballot(on)); continue;
#endif
            if (olen) {
                const uint32_t at = cursor + incl - olen;
                if ((uint64_t)at + olen > slice_cap) err |= ERR_CAPACITY;
                else if (in_lds) store_small(slice + at, hw, olen);
                else copy_name(slice + at, src, m, olen);
            }
            cursor += tot;
            nrec += (uint32_t)__popcll(__ballot(olen != 0u));
        }
        (void)wb; (void)tile_idx; (void)tile_rel;
    }

    template <bool FASTQ, bool ALL, class LDS>
    __device__ __forceinline__ void batch(LDS& L, uint32_t E, uint32_t wb, uint64_t tile_idx,
                                          uint32_t tile_rel, uint64_t re, const uint8_t* __restrict__ buf) {
        static_assert(FASTQ && !ALL, "the names sink runs on the sparse FASTQ path");
        const int lane = threadIdx.x & 63;
        for (uint32_t e0 = 0; e0 < E; e0 += WAVE) {
            const uint32_t e = e0 + lane;
            const bool on = e < E;
            const uint32_t s = HISTORY + (on ? e : 0);
            const uint32_t rank = wb + e;
            const uint32_t p = L.pos[s];
            const uint64_t abs_next = tile_idx + (uint64_t)(uint32_t)(p - tile_rel) + 1;
            const uint32_t role = rank & 3u;
            const uint8_t* src = nullptr;
            uint32_t m = 0;       // bytes to copy (the '\n' comes on top)
            uint32_t olen = 0;    // m + 1 for a record end, else 0
            if (on) {
                // the structural validation of the stats / index kernels (strict 4-line FASTQ)
                if (role == 1u) {
                    if (next_char(L, s, abs_next, re, buf) != '+') err |= ERR_BAD_PLUS;
                } else if (role == 0u) {
                    if (next_char(L, s, abs_next, re, buf) == '+') err |= ERR_BAD_PLUS;
                } else if (role == 3u) {
                    const uint32_t p1 = L.pos[s - 1], p2 = L.pos[s - 2], p3 = L.pos[s - 3], p4 = L.pos[s - 4];
                    const uint32_t lq = p - p1 - 1u, ls = p2 - p3 - 1u, lh = p3 - p4 - 1u;
                    if (lq != ls) err |= ERR_LEN_MISMATCH;
                    if (abs_next < re && next_char(L, s, abs_next, re, buf) != '@') err |= ERR_BAD_HEADER;
                    const uint8_t* h = buf + abs_of(p4, tile_idx, tile_rel) + 2;  // the header without its marker
                    m = lh ? lh - 1u : 0u;
                    src = h;
                    if (only_id) {
                        uint32_t off = 0;
                        m = id_span_of(h, m, id_mode, &off, lim);
                        src = h + off;
                    }
                    olen = m + 1u;
                }
            }
            const uint32_t incl = wave_incl_scan<DPP>(olen);
            const uint32_t tot = wave_last(incl);
            if (olen) {
                const uint32_t at = cursor + incl - olen;
                if ((uint64_t)at + olen <= slice_cap) {
                    uint8_t* dst = slice + at;
                    copy_name(dst, src, m, olen);
                } else {
                    err |= ERR_CAPACITY;
                }
            }
            cursor += tot;
            nrec += (uint32_t)__popcll(__ballot(olen != 0u));
        }
    }
};

// (one argument struct: what a wave needs once per range is read from it there -- BSK_KARG, stream_core_dev.hpp)
struct NamesArgs {
    const uint8_t* buf;
    uint64_t n;
    const uint64_t* anchors;
    uint32_t nranges;
    uint32_t* queue;
    NamesDev D;
};

template <bool DPP>
__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) __attribute__((amdgpu_waves_per_eu(BSK_NAMES_WAVES, 8)))
void k_names(NamesArgs a) {
    __shared__ Lds<true, false, BSK_NAMES_WINDOW> s_l[WAVES_PER_BLOCK];  // 64 whole records per sink call (NamesSink::records)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Lds<true, false, BSK_NAMES_WINDOW>& L = s_l[wave];
    NamesSink<DPP> sink;
    if constexpr (NamesSink<DPP>::HEAD16) {
        __shared__ uint4 s_h16[WAVES_PER_BLOCK][BSK_NAMES_WINDOW / 4 + 1];
        sink.h16 = s_h16[wave];
    }
    const uint8_t* __restrict__ buf = a.buf;
    const uint64_t n = a.n;
    sink.only_id = a.D.only_id;
    sink.id_mode = a.D.id_mode;
    sink.lim = buf + n;
    PredConsts P;  // unused (sparse path)
    P.k20 = P.k30 = 0;
    P.ngap = 0;
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(BSK_KARG(NamesArgs, queue), 1u);
        r = wave_first(r);
        const uint32_t nranges = BSK_KARG(NamesArgs, nranges);
        if (r >= nranges) break;
        const uint64_t* anchors = BSK_KARG(NamesArgs, anchors);
        const uint64_t n_eff = anchors[nranges];
        uint64_t rs = anchors[r], re = anchors[r + 1];
        rs = rs < n_eff ? rs : n_eff;
        re = re < n_eff ? re : n_eff;
        if (rs >= re) {
            if (lane == 0) { BSK_KARG(NamesArgs, D.range_bytes)[r] = 0; BSK_KARG(NamesArgs, D.range_count)[r] = 0; }
            continue;
        }
        sink.slice_cap = (uint32_t)BSK_KARG(NamesArgs, D.slice_cap);
        sink.begin_range(BSK_KARG(NamesArgs, D.slices) + (uint64_t)r * sink.slice_cap);
        stream_range<true, false, DPP>(L, buf, n, rs, re, re == n_eff, P, sink);
        if (lane == 0) { BSK_KARG(NamesArgs, D.range_bytes)[r] = sink.cursor; BSK_KARG(NamesArgs, D.range_count)[r] = sink.nrec; }
    }
    const uint32_t err = wave_or_u32(sink.err);
    if (lane == 0 && err) atomicOr((unsigned long long*)&BSK_KARG(NamesArgs, D.status)[0], (unsigned long long)err);
}

// slices -> one text: block (r, k) copies the k-th 16 KiB of range r's slice to its place.  A slice begins 16-byte
// aligned, its destination anywhere: aligned 16-byte loads, unaligned 16-byte stores.
constexpr uint32_t COMPACT_CHUNK = 16384;

__global__ __launch_bounds__(256) void k_names_compact(const uint8_t* __restrict__ slices, uint64_t slice_cap,
                                                       const uint64_t* __restrict__ range_bytes,
                                                       const uint64_t* __restrict__ range_base, uint32_t chunks_per_range,
                                                       uint8_t* __restrict__ out) {
    const uint32_t r = blockIdx.x / chunks_per_range;
    const uint32_t k = blockIdx.x % chunks_per_range;
    const uint64_t nb = range_bytes[r];
    const uint64_t c0 = (uint64_t)k * COMPACT_CHUNK;
    if (c0 >= nb) return;
    const uint64_t c1 = c0 + COMPACT_CHUNK < nb ? c0 + COMPACT_CHUNK : nb;
    const uint8_t* src = slices + (uint64_t)r * slice_cap;
    uint8_t* dst = out + range_base[r];
    for (uint64_t i = c0 + 16ull * threadIdx.x; i < c1; i += 16ull * 256) {
        if (i + 16 <= c1) {
            const uint4 v = *reinterpret_cast<const uint4*>(src + i);
            __builtin_memcpy(dst + i, &v, 16);
        } else {
            for (uint64_t j = i; j < c1; ++j) dst[j] = src[j];
        }
    }
}

}  // namespace

hipError_t launch_names(bool dpp, int blocks, const uint8_t* buf, uint64_t n, const uint64_t* anchors, uint32_t nranges,
                        uint32_t* queue, const NamesDev& D, hipStream_t st) {
    const dim3 b(WAVES_PER_BLOCK * WAVE);
    const NamesArgs a{buf, n, anchors, nranges, queue, D};
    if (dpp) hipLaunchKernelGGL((k_names<true>), dim3(blocks), b, 0, st, a);
    else hipLaunchKernelGGL((k_names<false>), dim3(blocks), b, 0, st, a);
    return hipGetLastError();
}

int names_max_blocks_per_cu(bool dpp) {
    int nb = 0;
    const void* f = dpp ? (const void*)k_names<true> : (const void*)k_names<false>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, WAVES_PER_BLOCK * WAVE, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

hipError_t launch_names_compact(const NamesDev& D, const uint64_t* range_base, uint32_t nranges, uint8_t* out, hipStream_t st) {
    const uint32_t cpr = (uint32_t)((D.slice_cap + COMPACT_CHUNK - 1) / COMPACT_CHUNK);
    if (nranges == 0 || cpr == 0) return hipSuccess;
    hipLaunchKernelGGL(k_names_compact, dim3(nranges * cpr), dim3(256), 0, st, D.slices, D.slice_cap, D.range_bytes, range_base,
                       cpr, out);
    return hipGetLastError();
}

}  // namespace bsk
