// Segmented copy (see ops_segcopy.hpp).  Measured on 8 GB of 317-byte records, 80 % kept
// (scripts/experiments/copy_rates.hip): 4 lanes per record with unaligned partial-line stores 2.9 TB/s (read + write),
// this kernel 4.9 TB/s, a plain aligned copy 5.0-5.2 TB/s.
#include <hip/hip_runtime.h>

#include "ops_segcopy.hpp"

namespace bsk {
namespace {

__global__ __launch_bounds__(256) void k_seg_build_fastq(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t,
                                                         const uint32_t* __restrict__ out_len, uint64_t* __restrict__ seg_src,
                                                         unsigned long long* __restrict__ n_other, const uint32_t* __restrict__ ren_ord) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint32_t n = out_len[i];
    uint64_t s = 0;
    if (n) {
        const uint64_t st = t.start[i];
        // (rename: a record with an ordinal gets a new head and stays with the record-wise emit)
        if (t.aux[i] == 1u && st + n <= buf_n && !(ren_ord && ren_ord[i])) s = (uint64_t)(uintptr_t)(buf + st);
        else atomicAdd(n_other, 1ull);  // rare: a '+' line that repeats the name, or the last record of a shard without '\n'
    }
    seg_src[i] = s;
}

// whole-record operators (range / head): the element is the record text, written back followed by '\n' -- a verbatim
// segment when that newline is the byte after the text in the shard (always, except for a last record without one)
__global__ __launch_bounds__(256) void k_seg_build_text(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t,
                                                        const uint32_t* __restrict__ out_len, uint64_t* __restrict__ seg_src,
                                                        unsigned long long* __restrict__ n_other) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint32_t n = out_len[i];
    uint64_t s = 0;
    if (n) {
        const uint64_t st = t.start[i];
        if (st + n <= buf_n && buf[st + n - 1] == '\n') s = (uint64_t)(uintptr_t)(buf + st);
        else atomicAdd(n_other, 1ull);
    }
    seg_src[i] = s;
}

// the records k_seg_build_text left out: text + '\n', byte by byte (one thread per record; there is at most one)
__global__ __launch_bounds__(256) void k_seg_fix_text(const uint8_t* __restrict__ buf, RecordTable t, const uint32_t* __restrict__ out_len,
                                                      const uint64_t* __restrict__ out_off, const uint64_t* __restrict__ seg_src,
                                                      uint8_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint32_t n = out_len[i];
    if (n == 0 || seg_src[i] != 0) return;
    const uint8_t* s = buf + t.start[i];
    uint8_t* o = out + out_off[i];
    for (uint32_t k = 0; k + 1 < n; ++k) o[k] = s[k];
    o[n - 1] = (uint8_t)'\n';
}

// sort: segment k is the record perm[k]; seg_rec (by record) tells the record-wise emit what is left to it
__global__ __launch_bounds__(256) void k_seg_build_fastq_perm(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t,
                                                              const uint32_t* __restrict__ out_len, const uint32_t* __restrict__ perm,
                                                              uint64_t* __restrict__ seg_sorted, uint64_t* __restrict__ seg_rec,
                                                              unsigned long long* __restrict__ n_other) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= t.n) return;
    const uint32_t i = perm[k];
    const uint32_t n = out_len[i];
    uint64_t s = 0;
    if (n) {
        const uint64_t st = t.start[i];
        if (t.aux[i] == 1u && st + n <= buf_n) s = (uint64_t)(uintptr_t)(buf + st);
        else atomicAdd(n_other, 1ull);
    }
    seg_sorted[k] = s;
    seg_rec[i] = s;
}

// duplicate: `times` segments per record with output, all from the record's text; seg_off2[i * times + j] = out_off[i] +
// j * (out_len[i] / times); the entry after the last one (= total) is written by the thread of the last record
__global__ __launch_bounds__(256) void k_seg_build_text_times(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t,
                                                              const uint32_t* __restrict__ out_len, const uint64_t* __restrict__ out_off,
                                                              uint32_t times, uint64_t* __restrict__ seg_src, uint64_t* __restrict__ seg_off2,
                                                              unsigned long long* __restrict__ n_other) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint32_t n = out_len[i];
    const uint64_t o = out_off[i];
    const uint32_t unit = n / times;
    uint64_t s = 0;
    if (n) {
        const uint64_t st = t.start[i];
        if (st + unit <= buf_n && buf[st + unit - 1] == '\n') s = (uint64_t)(uintptr_t)(buf + st);
        else atomicAdd(n_other, 1ull);
    }
    for (uint32_t j = 0; j < times; ++j) {
        seg_src[i * times + j] = s;
        seg_off2[i * times + j] = o + (uint64_t)j * unit;
    }
    if (i + 1 == t.n) seg_off2[t.n * times] = out_off[t.n];
}

__global__ __launch_bounds__(256) void k_seg_fix_text_times(const uint8_t* __restrict__ buf, RecordTable t, const uint32_t* __restrict__ out_len,
                                                            const uint64_t* __restrict__ out_off, uint32_t times,
                                                            const uint64_t* __restrict__ seg_src, uint8_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint32_t n = out_len[i];
    if (n == 0 || seg_src[i * times] != 0) return;
    const uint32_t unit = n / times;
    const uint8_t* s = buf + t.start[i];
    for (uint32_t j = 0; j < times; ++j) {
        uint8_t* o = out + out_off[i] + (uint64_t)j * unit;
        for (uint32_t k = 0; k + 1 < unit; ++k) o[k] = s[k];
        o[unit - 1] = (uint8_t)'\n';
    }
}

__global__ __launch_bounds__(256) void k_seg_first(const uint64_t* __restrict__ seg_off, uint64_t nseg, uint32_t* __restrict__ first4k) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseg) return;
    const uint64_t a = seg_off[i], b = seg_off[i + 1];
    for (uint64_t T = (a + SEG_TILE - 1) / SEG_TILE; T * SEG_TILE < b; ++T) first4k[T] = (uint32_t)i;
}

__device__ __forceinline__ uint32_t incl_scan64(uint32_t v) {  // inclusive sum over the 64 lanes
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

constexpr int32_t REL_MIN = -(1 << 30), REL_MAX = (1 << 30);

// The output bytes [tile0 * SEG_TILE, total) of the text whose whole length is `whole`; `out` is the address output byte 0
// WOULD have (a piece of the text gathered into a staging buffer: buffer - tile0 * SEG_TILE).  The whole text: tile0 = 0,
// total = whole.
__global__ __launch_bounds__(256) void k_seg_copy(const uint64_t* __restrict__ seg_src, const uint64_t* __restrict__ seg_off,
                                                  uint64_t nseg, const uint32_t* __restrict__ first4k, uint8_t* __restrict__ out,
                                                  uint64_t total, const uint8_t* lo, const uint8_t* hi, uint64_t tile0, uint64_t whole) {
    __shared__ int32_t s_rel[4][66];     // begin of the wave's segments relative to its tile (clamped); [m] = end of the last
    __shared__ uint64_t s_delta[4][64];  // source address of output byte x = delta + x
    __shared__ uint32_t s_hist[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t tile = tile0 + (uint64_t)blockIdx.x * 4u + wv;
    const uint64_t T0 = tile * SEG_TILE;
    if (T0 >= total) return;
    const uint64_t k0 = first4k[tile];
    const uint64_t k1 = (T0 + SEG_TILE < whole) ? first4k[tile + 1] : nseg - 1;  // last segment that can begin in the tile
    const uint64_t m64 = k1 - k0 + 1;
    if (m64 <= 64) {
        const uint32_t m = (uint32_t)m64;
        uint64_t sa = 0;
        {
            const uint64_t k = k0 + lane;
            uint64_t off = 0, offn = 0;
            if (lane < (int)m) { off = seg_off[k]; sa = seg_src[k]; offn = seg_off[k + 1]; }
            const int64_t rel = (int64_t)off - (int64_t)T0;
            s_rel[wv][lane] = lane < (int)m ? (int32_t)(rel < REL_MIN ? REL_MIN : rel) : REL_MAX;
            if (lane == (int)m - 1) {
                const int64_t reln = (int64_t)offn - (int64_t)T0;
                s_rel[wv][m] = (int32_t)(reln > REL_MAX ? REL_MAX : reln);
            }
            s_delta[wv][lane] = sa - off;
        }
        const uint64_t skipmask = __ballot(lane < (int)m && sa == 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int step = 0; step < (int)(SEG_TILE / 1024u); ++step) {
            const uint64_t T = T0 + 1024ull * step;
            if (T >= total) break;
            s_hist[wv][lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // segment `lane` begins at or before the first byte of chunk c (16 c) and after that of chunk c - 1
            if (lane < (int)m) {
                const int32_t rel = s_rel[wv][lane] - 1024 * step;
                const int32_t c = rel <= 0 ? 0 : (rel + 15) >> 4;
                if (c < 64) atomicAdd(&s_hist[wv][c], 1u);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t r = incl_scan64(s_hist[wv][lane]) - 1u;  // the segment that holds the chunk's first byte
            const uint64_t pos = T + 16ull * lane;
            if (pos >= total) continue;
            const int32_t prel = 1024 * step + 16 * lane;
            const int32_t next = s_rel[wv][r + 1];
            const bool skip_r = (skipmask >> r) & 1ull;
            if (prel + 16 <= next) {  // inside one segment
                if (!skip_r) {
                    uint4 v;
                    __builtin_memcpy(&v, (const uint8_t*)(uintptr_t)(s_delta[wv][r] + pos), 16);
                    *reinterpret_cast<uint4*>(out + pos) = v;
                }
                continue;
            }
            const uint32_t nb = pos + 16 <= total ? 16u : (uint32_t)(total - pos);
            uint32_t rb = r + 1;  // the next segment that is not empty
            while (rb < m && s_rel[wv][rb + 1] == s_rel[wv][rb]) ++rb;
            const uint8_t* pa = (const uint8_t*)(uintptr_t)(s_delta[wv][r] + pos);
            const uint8_t* pb = rb < m ? (const uint8_t*)(uintptr_t)(s_delta[wv][rb] + pos) : pa;
            if (nb == 16u && rb < m && prel + 16 <= s_rel[wv][rb + 1] && !skip_r && !((skipmask >> rb) & 1ull) && pa >= lo &&
                pa + 16 <= hi && pb >= lo && pb + 16 <= hi) {
                // exactly one boundary inside the chunk: two unaligned loads, bytes [0, cut) from the first
                uint32_t A[4], B[4], O[4];
                __builtin_memcpy(A, pa, 16);
                __builtin_memcpy(B, pb, 16);
                const int32_t cut = next - prel;  // 1..15
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int32_t c = cut - 4 * d;
                    const uint32_t mk = c >= 4 ? 0xFFFFFFFFu : c <= 0 ? 0u : ((1u << (8 * c)) - 1u);
                    O[d] = (A[d] & mk) | (B[d] & ~mk);
                }
                *reinterpret_cast<uint4*>(out + pos) = make_uint4(O[0], O[1], O[2], O[3]);
                continue;
            }
            // anything else (several boundaries, skipped segments, the end of the text or of the source buffer): the
            // source of every byte first, then all byte loads together
            uint32_t rr = r;
            const uint8_t* from[16];
            bool all = nb == 16u;
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) {
                from[j] = nullptr;
                if (j >= nb) continue;
                while (rr + 1 < m && prel + (int32_t)j >= s_rel[wv][rr + 1]) ++rr;
                if ((skipmask >> rr) & 1ull) all = false;
                else from[j] = (const uint8_t*)(uintptr_t)(s_delta[wv][rr] + pos + j);
            }
            uint8_t b[16];
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) b[j] = from[j] ? *from[j] : (uint8_t)0;
            if (all) {
                uint4 v;
                __builtin_memcpy(&v, b, 16);
                *reinterpret_cast<uint4*>(out + pos) = v;
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j)
                    if (from[j]) out[pos + j] = b[j];
            }
        }
        return;
    }
    // more than 64 segments begin in this tile (records of a few bytes, or long runs without output): every lane looks its
    // bytes up in the segment arrays themselves
    for (uint32_t ch = lane; ch < SEG_TILE / 16u; ch += 64u) {
        const uint64_t pos = T0 + 16ull * ch;
        if (pos >= total) break;
        uint64_t a = k0, b = k1;  // last k in [k0, k1] with seg_off[k] <= pos
        while (a < b) {
            const uint64_t mid = (a + b + 1) >> 1;
            if (seg_off[mid] <= pos) a = mid; else b = mid - 1;
        }
        uint64_t k = a;
        uint64_t nextk = seg_off[k + 1];
        uint64_t src = seg_src[k], base = seg_off[k];
        const uint32_t nb = pos + 16 <= total ? 16u : (uint32_t)(total - pos);
        for (uint32_t j = 0; j < nb; ++j) {
            const uint64_t x = pos + j;
            while (x >= nextk && k < k1) { ++k; base = nextk; nextk = seg_off[k + 1]; src = seg_src[k]; }
            if (src) out[x] = *(const uint8_t*)(uintptr_t)(src + (x - base));
        }
    }
}

}  // namespace

hipError_t launch_seg_build_fastq(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const uint32_t* out_len,
                                  uint64_t* seg_src, uint64_t* n_other, hipStream_t st, const uint32_t* ren_ord) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_seg_build_fastq, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, out_len, seg_src,
                       (unsigned long long*)n_other, ren_ord);
    return hipGetLastError();
}

hipError_t launch_seg_build_text(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const uint32_t* out_len,
                                 uint64_t* seg_src, uint64_t* n_other, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_seg_build_text, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, out_len, seg_src,
                       (unsigned long long*)n_other);
    return hipGetLastError();
}

hipError_t launch_seg_fix_text(const uint8_t* buf, const RecordTable& t, const uint32_t* out_len, const uint64_t* out_off,
                               const uint64_t* seg_src, uint8_t* out, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_seg_fix_text, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, t, out_len, out_off, seg_src, out);
    return hipGetLastError();
}

hipError_t launch_seg_build_fastq_perm(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const uint32_t* out_len,
                                       const uint32_t* perm, uint64_t* seg_sorted, uint64_t* seg_rec, uint64_t* n_other, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_seg_build_fastq_perm, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, out_len, perm,
                       seg_sorted, seg_rec, (unsigned long long*)n_other);
    return hipGetLastError();
}

hipError_t launch_seg_build_text_times(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const uint32_t* out_len,
                                       const uint64_t* out_off, uint32_t times, uint64_t* seg_src, uint64_t* seg_off2, uint64_t* n_other,
                                       hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_seg_build_text_times, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, out_len, out_off,
                       times, seg_src, seg_off2, (unsigned long long*)n_other);
    return hipGetLastError();
}

hipError_t launch_seg_fix_text_times(const uint8_t* buf, const RecordTable& t, const uint32_t* out_len, const uint64_t* out_off,
                                     uint32_t times, const uint64_t* seg_src, uint8_t* out, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_seg_fix_text_times, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, t, out_len, out_off, times,
                       seg_src, out);
    return hipGetLastError();
}

hipError_t launch_seg_first(const uint64_t* seg_off, uint64_t nseg, uint32_t* first4k, hipStream_t st) {
    if (nseg == 0) return hipSuccess;
    hipLaunchKernelGGL(k_seg_first, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, st, seg_off, nseg, first4k);
    return hipGetLastError();
}

hipError_t launch_seg_copy(const uint64_t* seg_src, const uint64_t* seg_off, uint64_t nseg, const uint32_t* first4k,
                           uint8_t* out, uint64_t total, const uint8_t* lo, const uint8_t* hi, hipStream_t st) {
    if (nseg == 0 || total == 0) return hipSuccess;
    const uint64_t tiles = seg_tiles(total);
    hipLaunchKernelGGL(k_seg_copy, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, seg_src, seg_off, nseg, first4k, out, total,
                       lo, hi, (uint64_t)0, total);
    return hipGetLastError();
}

hipError_t launch_seg_copy_range(const uint64_t* seg_src, const uint64_t* seg_off, uint64_t nseg, const uint32_t* first4k, uint8_t* dst,
                                 uint64_t from, uint64_t to, uint64_t whole, const uint8_t* lo, const uint8_t* hi, hipStream_t st) {
    if (nseg == 0 || to <= from) return hipSuccess;
    if (from % SEG_TILE || ((uintptr_t)dst & 15u)) return hipErrorInvalidValue;
    const uint64_t tiles = seg_tiles(to) - from / SEG_TILE;
    hipLaunchKernelGGL(k_seg_copy, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, seg_src, seg_off, nseg, first4k, dst - from, to,
                       lo, hi, from / SEG_TILE, whole);
    return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void k_slice_srcs(const uint8_t* slices, uint64_t slice_cap, uint32_t nranges, uint64_t* __restrict__ src) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nranges) src[r] = (uint64_t)(uintptr_t)(slices + (uint64_t)r * slice_cap);
}
}  // namespace

hipError_t launch_slice_srcs(const uint8_t* slices, uint64_t slice_cap, uint32_t nranges, uint64_t* src, hipStream_t st) {
    if (nranges == 0) return hipSuccess;
    hipLaunchKernelGGL(k_slice_srcs, dim3((nranges + 255) / 256), dim3(256), 0, st, slices, slice_cap, nranges, src);
    return hipGetLastError();
}

}  // namespace bsk
