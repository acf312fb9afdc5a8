// ============================================================================
// stream_fasta_light.hip -- the record table of a FASTA shard from its '>' bytes alone.
//
// k_index<FASTA> finds every line of the shard (67 newline events per 4 KiB tile of 60-column text, ~330 vector
// instructions per tile: 11.7 ms per 50 GB) to learn three things per record: where it begins, how long its header is,
// and whether its lines are equally long.  `translate` needs the first two exactly and can have the third VERIFIED for
// free: k_translate_wide checks every window of the text against the layout it was told (one line break, at the expected
// place, everything else ACGT) and flags a record that does not fit.  So for text that looks regular:
//   k_fasta_starts  a streaming read that looks for '>' only (sequence text holds none: ~60 vector instructions per
//                   tile, the speed of the read), keeps those that begin a line, per nominal chunk -- no anchors, no
//                   newline events;
//   k_fasta_heads   one thread per record: the header's end and the first sequence line's end (two short searches),
//                   the region up to the next record, and l_seq / text_w AS IF all lines but the last had the length of
//                   the first: L = q W + max(rem - 1, 0) for a region of q (W + 1) + rem bytes.
// The caller (translate_run_device) falls back to k_index whenever the wide kernel flags a record, a record is
// chromosome-sized, a first line is shorter than 16 bases, or a slice overflows -- never a different answer.
// SeqParser.Read, /root/reference/bigseqkit-lib/helper.go:219-250.  HBM-bound byte work; no MFMA.
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

#include "stream_core_dev.hpp"
#include "stream_fasta_light.hpp"
#include "text_dev.hpp"

namespace bsk {

namespace {

using namespace stream;

__global__ __launch_bounds__(WAVES_PER_BLOCK * WAVE) void k_fasta_starts(const uint8_t* __restrict__ buf, uint64_t n_eff, uint64_t chunk,
                                                                          uint32_t nranges, uint32_t* __restrict__ queue,
                                                                          uint64_t* __restrict__ sparse, uint64_t sparse_cap,
                                                                          uint64_t* __restrict__ range_count,
                                                                          uint64_t* __restrict__ status) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t err = 0;
    for (;;) {
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(queue, 1u);
        r = wave_first(r);
        if (r >= nranges) break;
        const uint64_t lo = (uint64_t)r * chunk, hi = lo + chunk < n_eff ? lo + chunk : n_eff;
        uint64_t* slice = sparse + (uint64_t)r * sparse_cap;
        uint32_t cursor = 0;  // (wave-uniform)
        for (uint64_t tile = lo; tile < hi; tile += TILE) {
            uint4 v[NPIECE];
            bool any = false;
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) {
                v[p] = load16(buf, hi, tile + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16);
                const uint32_t y0 = v[p].x ^ 0x3E3E3E3Eu, y1 = v[p].y ^ 0x3E3E3E3Eu, y2 = v[p].z ^ 0x3E3E3E3Eu, y3 = v[p].w ^ 0x3E3E3E3Eu;
                any |= ((((y0 - 0x01010101u) & ~y0) | ((y1 - 0x01010101u) & ~y1) | ((y2 - 0x01010101u) & ~y2) | ((y3 - 0x01010101u) & ~y3)) &
                        0x80808080u) != 0u;
            }
            if (__ballot(any) == 0ull) continue;  // (a tile of sequence text)
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) {
                const uint64_t at = tile + (uint64_t)p * PIECE_BYTES + (uint64_t)lane * 16;
                uint32_t m = eq_mask16(v[p], 0x3E3E3E3Eu);
                if (m) {
                    // keep the '>' that begin a line: the byte before is a newline (or the shard begins here)
                    const uint32_t nl = eq_mask16(v[p], 0x0A0A0A0Au);
                    uint32_t ok = m & (nl << 1);
                    if (m & 1u) {
                        if (at == 0 || buf[at - 1] == '\n') ok |= 1u;
                    }
                    m = ok & 0xFFFFu;
                }
                const uint64_t bal = __ballot(m != 0u);
                if (bal == 0ull) continue;
                const uint32_t c = (uint32_t)__popc(m);
                const uint32_t incl = wave_incl_scan<true>(c);
                uint32_t k = cursor + incl - c;
                while (m) {
                    const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
                    m &= m - 1u;
                    if ((uint64_t)k < sparse_cap) slice[k] = at + b; else err |= ERR_CAPACITY;
                    ++k;
                }
                cursor += wave_last(incl);
            }
        }
        if (lane == 0) range_count[r] = cursor < sparse_cap ? cursor : sparse_cap;
    }
    err = wave_or_u32(err);
    if (lane == 0 && err) atomicOr((unsigned long long*)&status[0], (unsigned long long)err);
}

__global__ __launch_bounds__(256) void k_fasta_starts_compact(const uint64_t* __restrict__ sparse, uint64_t sparse_cap,
                                                              const uint64_t* __restrict__ range_count,
                                                              const uint64_t* __restrict__ range_base, uint64_t n_eff, uint64_t total,
                                                              RecordTable t) {
    const uint32_t r = blockIdx.x;
    const uint64_t cnt = range_count[r], src = (uint64_t)r * sparse_cap, dst = range_base[r];
    for (uint64_t i = threadIdx.x; i < cnt; i += blockDim.x) t.start[dst + i] = sparse[src + i];
    if (r == 0 && threadIdx.x == 0) t.start[total] = n_eff;
}

// Eight lanes per record: lane j loads bytes [16 j, 16 j + 16) of the record, so the first 128 bytes arrive with ONE load
// per lane and no dependent chain (one thread per record walking header and first line 16 bytes at a time took 2.6 ms
// for 9.8 M records -- six dependent, uncoalesced loads each); the two newlines wanted -- end of the header, end of the
// first sequence line -- are the first two set bits of the group's 128-bit newline mask.  A record whose first 128 bytes
// do not hold both falls back to the serial search on lane 0 of its group.
__global__ __launch_bounds__(256) void k_fasta_heads(const uint8_t* __restrict__ buf, uint64_t n_eff, RecordTable t,
                                                     uint64_t* __restrict__ status) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const uint32_t j = threadIdx.x & 7u, lane = threadIdx.x & 63u;
    const bool live = i < t.n;
    const uint64_t s = live ? t.start[i] : 0, e = live ? t.start[i + 1] : 0;  // (start[n] = n_eff)
    const uint8_t* lim = buf + n_eff;
    const uint64_t span = e - s;
    uint32_t nl = 0;
    if (live && s + 16ull * j < e) {
        uint4 v = make_uint4(0, 0, 0, 0);
        const uint64_t at = s + 16ull * j;
        if (at + 16 <= n_eff) __builtin_memcpy(&v, buf + at, 16);   // (any alignment)
        else {
            uint32_t wv[4] = {0, 0, 0, 0};
            for (uint32_t b = 0; at + b < n_eff; ++b) wv[b >> 2] |= (uint32_t)buf[at + b] << ((b & 3u) * 8u);
            v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
        nl = eq_mask16(v, 0x0A0A0A0Au);
        const uint64_t left = e - (s + 16ull * j);
        if (left < 16) nl &= (1u << left) - 1u;                // bytes of the next record do not count
    }
    // the group's mask: bit 16 j + b; its two lowest set bits
    const uint32_t gbase = lane & ~7u;
    uint64_t mlo = 0, mhi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint64_t m = (uint64_t)(uint32_t)__shfl((int)nl, (int)(gbase + k), 64);
        if (k < 4) mlo |= m << (16 * k); else mhi |= m << (16 * (k - 4));
    }
    if (!live || j != 0) return;
    if (span > 0xFFFFFFFFull) { atomicOr((unsigned long long*)&status[0], (unsigned long long)ERR_LINE_TOO_LONG); return; }
    auto first_bit = [](uint64_t lo, uint64_t hi) -> uint32_t { return lo ? (uint32_t)__ffsll((long long)lo) - 1u : (hi ? 64u + (uint32_t)__ffsll((long long)hi) - 1u : 128u); };
    uint32_t lh = first_bit(mlo, mhi);
    uint32_t second = 128u;
    if (lh < 128u) {
        uint64_t lo2 = mlo, hi2 = mhi;
        if (lh < 64u) lo2 &= lo2 - 1ull; else hi2 &= hi2 - 1ull;
        second = first_bit(lo2, hi2);
    }
    if (lh >= 128u) lh = span <= 128u ? (uint32_t)span : find_byte_in(buf + s, (uint32_t)span, '\n', lim);   // header line without its newline (== span: none)
    const uint64_t region = span > (uint64_t)lh + 1 ? span - lh - 1 : 0;                                        // bytes after the header's newline
    uint32_t w = 0, lseq = 0, tw = 0;
    if (region) {
        if (second < 128u) w = second - lh - 1u;
        else w = span <= 128u ? (uint32_t)region : find_byte_in(buf + s + lh + 1, (uint32_t)region, '\n', lim);   // first sequence line
        // the last record of a shard that does not end with a newline: as if it did
        const uint64_t R = (i + 1 == t.n && buf[n_eff - 1] != '\n') ? region + 1 : region;
        if (R <= (uint64_t)w + 1) {       // one line
            lseq = w;
            tw = 0;
        } else if (w < 16u) {             // (the text views want lines of 16 bases and more: not this path's layout)
            atomicOr((unsigned long long*)&status[0], (unsigned long long)ERR_LIGHT_UNFIT);
        } else {
            const uint64_t qn = R / ((uint64_t)w + 1), rem = R % ((uint64_t)w + 1);
            lseq = (uint32_t)(qn * w + (rem ? rem - 1 : 0));
            tw = w;
        }
    }
    t.l_head[i] = lh;
    t.l_seq[i] = lseq;
    t.aux[i] = (uint32_t)region;
    t.text_w[i] = tw;
}

}  // namespace

hipError_t launch_fasta_starts(int blocks, const uint8_t* buf, uint64_t n_eff, uint64_t chunk, uint32_t nranges, uint32_t* queue,
                               uint64_t* sparse, uint64_t sparse_cap, uint64_t* range_count, uint64_t* status, hipStream_t st) {
    hipLaunchKernelGGL(k_fasta_starts, dim3(blocks), dim3(WAVES_PER_BLOCK * WAVE), 0, st, buf, n_eff, chunk, nranges, queue, sparse,
                       sparse_cap, range_count, status);
    return hipGetLastError();
}

int fasta_starts_max_blocks_per_cu() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_fasta_starts, WAVES_PER_BLOCK * WAVE, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

hipError_t launch_fasta_starts_compact(const uint64_t* sparse, uint64_t sparse_cap, const uint64_t* range_count,
                                       const uint64_t* range_base, uint32_t nranges, uint64_t n_eff, uint64_t total, RecordTable t,
                                       hipStream_t st) {
    hipLaunchKernelGGL(k_fasta_starts_compact, dim3(nranges), dim3(256), 0, st, sparse, sparse_cap, range_count, range_base, n_eff, total, t);
    return hipGetLastError();
}

hipError_t launch_fasta_heads(const uint8_t* buf, uint64_t n_eff, RecordTable t, uint64_t* status, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fasta_heads, dim3((unsigned)((t.n * 8 + 255) / 256)), dim3(256), 0, st, buf, n_eff, t, status);
    return hipGetLastError();
}

}  // namespace bsk
