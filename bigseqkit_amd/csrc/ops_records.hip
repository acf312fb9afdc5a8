// range / head / duplicate: verbatim copies of whole records (see ops_records.hpp).
#include <hip/hip_runtime.h>

#include "ops_records.hpp"

namespace bsk {
namespace {

// Text bytes of record i without its final newline.
//   FASTQ: exactly the four lines (blank lines after the last record are not part of it)
//   FASTA: everything up to the next record start / the shard end, minus ONE trailing '\n' (helper.go:51-56)
__device__ __forceinline__ uint64_t record_text_len(const uint8_t* __restrict__ buf, uint64_t buf_n, const RecordTable& t,
                                                    int fastq, uint64_t i) {
    if (fastq) {
        uint64_t T = (uint64_t)t.l_head[i] + 1u + t.l_seq[i] + 1u + t.aux[i] + 1u + t.l_seq[i];
        const uint64_t s = t.start[i], avail = buf_n - s;
        if (avail <= T) {  // the shard ends with this record and without a final newline: an EMPTY quality line then
            T = avail;     // leaves the newline of the '+' line at the end of the element, and ReadFixer strips it
            if (T && buf[s + T - 1] == '\n') --T;
        }
        return T;
    }
    const uint64_t s = t.start[i];
    uint64_t e = i + 1 == t.n ? buf_n : t.start[i + 1];  // (start[n] stops before blank lines at the end of the shard)
    if (e > s && buf[e - 1] == '\n') --e;
    return e - s;
}

__global__ __launch_bounds__(256) void k_records_size(const uint8_t* __restrict__ buf, uint64_t buf_n, RecordTable t,
                                                      RecordsParams P, uint32_t* __restrict__ out_len,
                                                      uint64_t* __restrict__ status) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const int64_t g = P.first_record + (int64_t)i;
    uint64_t bytes = 0;
    if (P.lo <= g && g < P.hi) bytes = (record_text_len(buf, buf_n, t, P.fastq, i) + 1u) * (uint64_t)P.times;
    if (bytes > 0xFFFFFFFFull) {
        atomicOr((unsigned long long*)&status[0], (unsigned long long)ERR_RECORD_TOO_LARGE);
        bytes = 0;
    }
    out_len[i] = (uint32_t)bytes;
}

constexpr uint32_t COPY_TILE = 16u * 1024u;  // output bytes per block: 256 lanes x 4 steps x 16 B
constexpr uint32_t COPY_STEPS = COPY_TILE / (256u * 16u);
constexpr uint32_t COPY_STAGE = 1024u;      // records of one tile whose offsets are kept in LDS

// tile_first[T] = the record that holds output byte T * COPY_TILE.  One thread per record: a 27-step binary search per
// tile over the offsets of 80 M records (all dependent loads) cost more than the copy itself.
__global__ __launch_bounds__(256) void k_tile_first(const uint64_t* __restrict__ out_off, uint64_t n,
                                                    uint32_t* __restrict__ tile_first) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = out_off[i], b = out_off[i + 1];
    for (uint64_t T = (a + COPY_TILE - 1) / COPY_TILE; T * COPY_TILE < b; ++T) tile_first[T] = (uint32_t)i;
}

__device__ __forceinline__ uint4 load16(const uint8_t* p) {  // any alignment
    uint4 q;
    __builtin_memcpy(&q, p, 16);
    return q;
}

// offsets / starts of the records [i_lo, i_hi] of a tile: from LDS (staged) or straight from memory (more records in
// the tile than the stage holds: tiny records)
struct TileGlobal {
    const uint64_t* off;
    const uint64_t* start;
    uint64_t shift;
    __device__ __forceinline__ uint64_t o(uint64_t i) const { return off[i - shift]; }
    __device__ __forceinline__ uint64_t s(uint64_t i) const { return start[i - shift]; }
};

// last i in [lo, hi] with o(i) <= x
template <class A>
__device__ __forceinline__ uint64_t record_of(const A& a, uint64_t lo, uint64_t hi, uint64_t x) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi + 1) >> 1;
        if (a.o(mid) <= x) lo = mid; else hi = mid - 1;
    }
    return lo;
}

struct TileShared {  // distinct type so that the accesses compile to ds_read (a generic pointer gives flat loads)
    uint64_t shift;
    __device__ __forceinline__ uint64_t o(uint64_t i) const;
    __device__ __forceinline__ uint64_t s(uint64_t i) const;
};
__shared__ uint64_t s_off[COPY_STAGE + 1];
__shared__ uint64_t s_start[COPY_STAGE];
__device__ __forceinline__ uint64_t TileShared::o(uint64_t i) const { return s_off[i - shift]; }
__device__ __forceinline__ uint64_t TileShared::s(uint64_t i) const { return s_start[i - shift]; }

// MULTI: P.times > 1 (a 64-bit division per chunk only then)
template <bool MULTI, class A>
__device__ __forceinline__ void copy_tile(const A& acc, const uint8_t* __restrict__ buf, const RecordsParams& P,
                                          uint8_t* __restrict__ out, uint64_t t0, uint64_t t1, uint64_t i_lo, uint64_t i_hi) {
    // phase 1: where the 16 bytes of every step come from; phase 2: the loads, all in flight together; phase 3: stores
    uint64_t rec[COPY_STEPS];
    uint32_t rr[COPY_STEPS];
    uint4 v[COPY_STEPS];
    uint32_t fast = 0;
#pragma unroll
    for (uint32_t step = 0; step < COPY_STEPS; ++step) {
        const uint64_t x0 = t0 + ((uint64_t)step * 256u + threadIdx.x) * 16u;
        rec[step] = 0; rr[step] = 0;
        if (x0 >= t1) continue;
        const uint64_t i = record_of(acc, i_lo, i_hi, x0);
        const uint64_t base = acc.o(i);
        const uint64_t len = acc.o(i + 1) - base;                                 // < 2^32 (out_len is 32 bits)
        const uint32_t unit = MULTI ? (uint32_t)len / P.times : (uint32_t)len;   // text + '\n'
        const uint32_t r = MULTI ? (uint32_t)(x0 - base) % unit : (uint32_t)(x0 - base);
        rec[step] = i; rr[step] = r;
        if ((uint64_t)r + 16u < unit && x0 + 16u <= t1) fast |= 1u << step;  // inside the text of one copy
    }
#pragma unroll
    for (uint32_t step = 0; step < COPY_STEPS; ++step)
        v[step] = (fast >> step & 1u) ? load16(buf + acc.s(rec[step]) + rr[step]) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (uint32_t step = 0; step < COPY_STEPS; ++step) {
        const uint64_t x0 = t0 + ((uint64_t)step * 256u + threadIdx.x) * 16u;
        if (fast >> step & 1u) *reinterpret_cast<uint4*>(out + x0) = v[step];  // x0 and the output buffer are 16-byte aligned
    }
#pragma unroll 1
    for (uint32_t step = 0; step < COPY_STEPS; ++step) {
        const uint64_t x0 = t0 + ((uint64_t)step * 256u + threadIdx.x) * 16u;
        if ((fast >> step & 1u) || x0 >= t1) continue;
        // a chunk that crosses the end of a copy, of a record or of the output: the source of each byte first, then all
        // byte loads together (a load per loop iteration made this path -- 7 % of the chunks, but some lane of nearly
        // every wave -- cost more than the rest of the kernel)
        uint64_t i = record_of(acc, i_lo, i_hi, x0);
        uint64_t next = acc.o(i + 1);
        uint32_t unit, r;
        {
            const uint64_t base = acc.o(i);
            unit = MULTI ? (uint32_t)(next - base) / P.times : (uint32_t)(next - base);
            r = MULTI ? (uint32_t)(x0 - base) % unit : (uint32_t)(x0 - base);
        }
        const uint8_t* sp = buf + acc.s(i);
        const uint32_t nb = x0 + 16u <= t1 ? 16u : (uint32_t)(t1 - x0);
        const uint8_t* from[16];
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) {
            from[k] = nullptr;
            if (k >= nb) continue;
            const uint64_t x = x0 + k;
            while (x >= next) {  // next record with output
                ++i;
                const uint64_t base = next;
                next = acc.o(i + 1);
                if (next > base) {
                    unit = MULTI ? (uint32_t)(next - base) / P.times : (uint32_t)(next - base);
                    sp = buf + acc.s(i);
                    r = 0;
                }
            }
            if (r + 1u != unit) from[k] = sp + r;
            if (++r == unit) r = 0;
        }
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        uint8_t c[16];
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) c[k] = from[k] ? *from[k] : (uint8_t)'\n';
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) w[k >> 2] |= (uint32_t)c[k] << (8u * (k & 3u));
        if (nb == 16u) {
            *reinterpret_cast<uint4*>(out + x0) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (uint32_t k = 0; k < nb; ++k) out[x0 + k] = (uint8_t)(w[k >> 2] >> (8u * (k & 3u)));
        }
    }
}

template <bool MULTI>
__global__ __launch_bounds__(256) void k_records_copy(const uint8_t* __restrict__ buf, RecordTable t, RecordsParams P,
                                                      const uint64_t* __restrict__ out_off,
                                                      const uint32_t* __restrict__ tile_first, uint8_t* __restrict__ out,
                                                      uint64_t total) {
    const uint64_t t0 = (uint64_t)blockIdx.x * COPY_TILE;
    const uint64_t t1 = t0 + COPY_TILE < total ? t0 + COPY_TILE : total;
    // records of this tile: [i_lo, i_hi] (i_hi may begin exactly at t1; it is then never selected)
    const uint64_t i_lo = tile_first[blockIdx.x];
    const uint64_t i_hi = blockIdx.x + 1u < gridDim.x ? tile_first[blockIdx.x + 1u] : t.n - 1;
    // their offsets and starts go through LDS when they fit (reads: ~52 records per tile); every lane then finds the
    // record of its 16 bytes without touching memory
    if (i_hi - i_lo + 1u <= COPY_STAGE) {
        for (uint64_t k = threadIdx.x; k < i_hi - i_lo + 2u; k += 256u) {
            s_off[k] = out_off[i_lo + k];
            if (k < i_hi - i_lo + 1u) s_start[k] = t.start[i_lo + k];
        }
        __syncthreads();
        copy_tile<MULTI>(TileShared{i_lo}, buf, P, out, t0, t1, i_lo, i_hi);
    } else {
        copy_tile<MULTI>(TileGlobal{out_off, t.start, 0}, buf, P, out, t0, t1, i_lo, i_hi);
    }
}

}  // namespace

hipError_t launch_records_size(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const RecordsParams& P,
                               uint32_t* out_len, uint64_t* status, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_records_size, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, buf_n, t, P, out_len, status);
    return hipGetLastError();
}

hipError_t launch_records_copy(const uint8_t* buf, const RecordTable& t, const RecordsParams& P, const uint64_t* out_off,
                               uint32_t* tile_first, uint8_t* out, uint64_t total, hipStream_t st) {
    if (t.n == 0 || total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tile_first, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, out_off, t.n, tile_first);
    const dim3 grid((unsigned)records_copy_tiles(total));
    if (P.times > 1) hipLaunchKernelGGL(k_records_copy<true>, grid, dim3(256), 0, st, buf, t, P, out_off, tile_first, out, total);
    else hipLaunchKernelGGL(k_records_copy<false>, grid, dim3(256), 0, st, buf, t, P, out_off, tile_first, out, total);
    return hipGetLastError();
}

uint64_t records_copy_tiles(uint64_t total) { return (total + COPY_TILE - 1) / COPY_TILE; }

}  // namespace bsk
