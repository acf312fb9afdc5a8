// Probe (not product code): what a copy costs on gfx950 as a function of alignment and of the record structure.
//   A  aligned 16 B/lane load + aligned 16 B/lane store (the ceiling)
//   B  source misaligned by k bytes (unaligned dwordx4 loads), aligned stores
//   C  aligned loads, destination misaligned by k bytes (unaligned dwordx4 stores)
//   D  aligned loads + funnel shift through the neighbour lane (DPP wave_shl) + aligned stores
//   E  317-byte records, 4 lanes per record, 16-byte unaligned copies (what k_seq_emit<4> does for whole records)
//   F  the same records, one wave per 1 KiB of OUTPUT: unaligned load per lane at (record, offset), aligned store
// Build: hipcc --offload-arch=gfx950 -O3 copy_rates.hip -o bin/copy_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t n16, int so, int d_o) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256ull) {
        uint4 v;
        __builtin_memcpy(&v, src + 16 * i + so, 16);
        __builtin_memcpy(dst + 16 * i + d_o, &v, 16);
    }
}

__global__ __launch_bounds__(256) void k_copy_shift(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t n16, int so) {
    // dst[16 i ..] = src[16 i + so ..]: aligned loads of chunk i and (through the next lane) chunk i + 1
    const int lane = threadIdx.x & 63;
    for (uint64_t i0 = (blockIdx.x * 4ull + (threadIdx.x >> 6)) * 64ull; i0 < n16; i0 += (uint64_t)gridDim.x * 256ull) {
        const uint64_t i = i0 + lane;
        uint4 a = *reinterpret_cast<const uint4*>(src + 16 * i);
        uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.x, 0x130, 0xf, 0xf, false);  // wave_shl:1 -> lane l gets lane l+1
        if (lane == 63) nx = *reinterpret_cast<const uint32_t*>(src + 16 * (i + 1));
        const uint32_t s = (uint32_t)so & 3u;  // (probe: shifts below 4 bytes)
        uint4 o;
        o.x = __builtin_amdgcn_alignbyte(a.y, a.x, s);
        o.y = __builtin_amdgcn_alignbyte(a.z, a.y, s);
        o.z = __builtin_amdgcn_alignbyte(a.w, a.z, s);
        o.w = __builtin_amdgcn_alignbyte(nx, a.w, s);
        *reinterpret_cast<uint4*>(dst + 16 * i) = o;
    }
}

// E: record r (317 bytes at 317 r) kept if keep[r]; out offset off[r]; 4 lanes per record
__global__ __launch_bounds__(256) void k_rec4(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint64_t* __restrict__ off,
                                              const uint8_t* __restrict__ keep, uint64_t nrec) {
    const uint64_t g = (blockIdx.x * 256ull + threadIdx.x) >> 2;
    const uint32_t sub = threadIdx.x & 3;
    if (g >= nrec || !keep[g]) return;
    const uint8_t* s = src + 317 * g;
    uint8_t* d = dst + off[g];
    uint32_t i = 16 * sub;
    for (; i + 16 <= 317; i += 64) {
        uint4 v;
        __builtin_memcpy(&v, s + i, 16);
        __builtin_memcpy(d + i, &v, 16);
    }
    if (sub == 3) for (uint32_t j = 304; j < 317; ++j) d[j] = s[j];
}

// F: one lane per 16 output bytes; record of an output position by division (all kept records have 317 bytes); list[k] = k-th kept record
__global__ __launch_bounds__(256) void k_out16(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint32_t* __restrict__ list,
                                               uint64_t nout16, uint64_t nkept) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nout16; i += (uint64_t)gridDim.x * 256ull) {
        const uint64_t pos = 16 * i;
        const uint64_t k = pos / 317;
        const uint32_t o = (uint32_t)(pos - 317 * k);
        uint4 v;
        if (o + 16 <= 317 || k + 1 >= nkept) {
            __builtin_memcpy(&v, src + 317ull * list[k] + o, 16);
        } else {  // crosses into the next kept record
            uint8_t b[16];
            const uint8_t* s0 = src + 317ull * list[k];
            const uint8_t* s1 = src + 317ull * list[k + 1];
            for (uint32_t j = 0; j < 16; ++j) b[j] = o + j < 317 ? s0[o + j] : s1[o + j - 317];
            __builtin_memcpy(&v, b, 16);
        }
        *reinterpret_cast<uint4*>(dst + pos) = v;
    }
}


// G: segmented copy as the product would do it: segments k = 0..K-1 with source address seg_src[k] and output offset
// seg_off[k] (seg_off[K] = total); first4k[T] = last k with seg_off[k] <= 4096 T.  One wave per 4 KiB of output, four
// steps of 1 KiB; the wave's segments (<= 64) sit in LDS; the segment of a 16-byte chunk = prefix sum of "segments that
// begin in this chunk".
__global__ __launch_bounds__(256) void k_seg_copy(const uint8_t* __restrict__ src, const uint64_t* __restrict__ seg_src,
                                                  const uint64_t* __restrict__ seg_off, uint64_t K,
                                                  const uint32_t* __restrict__ first4k, uint8_t* __restrict__ dst, uint64_t total,
                                                  const uint8_t* lo, const uint8_t* hi) {
    __shared__ uint32_t s_rel[4][66];
    __shared__ uint64_t s_delta[4][64];
    __shared__ uint32_t s_hist[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t tile = blockIdx.x * 4ull + wv;
    const uint64_t T0 = tile * 4096ull;
    if (T0 >= total) return;
    const uint64_t k0 = first4k[tile];
    const uint64_t k1 = (T0 + 4096 < total) ? first4k[tile + 1] : K - 1;   // last segment that can begin inside the tile
    const uint32_t m = (uint32_t)(k1 - k0 + 1);
    // (probe: m <= 64 assumed)
    {
        const uint64_t k = k0 + lane;
        uint64_t off = 0, sa = 0;
        if (lane < (int)m) { off = seg_off[k]; sa = seg_src[k]; }
        const uint64_t offn = (lane < (int)m) ? seg_off[k + 1] : 0;
        int64_t rel = (int64_t)off - (int64_t)T0;
        s_rel[wv][lane] = lane < (int)m ? (uint32_t)(rel < -1000000000ll ? -1000000000ll : rel) : 0x7FFFFFFFu;
        if (lane == (int)m - 1) s_rel[wv][m] = (uint32_t)((int64_t)offn - (int64_t)T0 > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : (int64_t)offn - (int64_t)T0);
        s_delta[wv][lane] = sa - off;   // source address of output byte x = delta + x
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int step = 0; step < 4; ++step) {
        const uint64_t T = T0 + 1024ull * step;
        if (T >= total) break;
        s_hist[wv][lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // record lane: the chunk in which segment `lane` begins (chunk c covers (16 c - 16, 16 c] ... begins at or before
        // byte 16 c of this step); segments that began before this step count for chunk 0
        if (lane < (int)m) {
            const int32_t rel = (int32_t)s_rel[wv][lane] - 1024 * step;
            const int32_t c = rel <= 0 ? 0 : (rel + 15) >> 4;
            if (c < 64) atomicAdd(&s_hist[wv][c], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t cnt = s_hist[wv][lane];
        // inclusive scan over lanes
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)cnt, d, 64);
            if (lane >= d) cnt += o;
        }
        const uint32_t r = cnt - 1u;                       // segment of the chunk's first byte
        const uint64_t pos = T + 16ull * lane;
        if (pos >= total) continue;
        const int32_t prel = (int32_t)(1024 * step + 16 * lane);
        const int32_t next = (int32_t)s_rel[wv][r + 1];
        if (prel + 16 <= next) {
            uint4 v;
            __builtin_memcpy(&v, (const uint8_t*)(s_delta[wv][r] + pos), 16);
            *reinterpret_cast<uint4*>(dst + pos) = v;
        } else {
            const uint32_t nb = pos + 16 <= total ? 16u : (uint32_t)(total - pos);
            const uint8_t* pa = (const uint8_t*)(s_delta[wv][r] + pos);
            const uint8_t* pb = (r + 1 < m) ? (const uint8_t*)(s_delta[wv][r + 1] + pos) : pa;
            const int32_t next2 = (r + 2 <= m) ? (int32_t)s_rel[wv][r + 2 <= m ? r + 2 : m] : 0x7FFFFFF0;
            if (nb == 16 && r + 1 < m && prel + 16 <= next2 && pa >= lo && pa + 16 <= hi && pb >= lo && pb + 16 <= hi) {
                // exactly one boundary inside the chunk: two unaligned loads, bytes [0, cut) from the first
                uint32_t A[4], B[4], O[4];
                __builtin_memcpy(A, pa, 16);
                __builtin_memcpy(B, pb, 16);
                const uint32_t cut = (uint32_t)(next - prel);   // 1..15
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int32_t c = (int32_t)cut - 4 * d;
                    const uint32_t mk = c >= 4 ? 0xFFFFFFFFu : c <= 0 ? 0u : ((1u << (8 * c)) - 1u);
                    O[d] = (A[d] & mk) | (B[d] & ~mk);
                }
                *reinterpret_cast<uint4*>(dst + pos) = make_uint4(O[0], O[1], O[2], O[3]);
            } else {
                uint32_t rr = r;
                const uint8_t* from[16];
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) {
                    from[j] = nullptr;
                    if (j >= nb) continue;
                    while (rr + 1 < m && prel + (int32_t)j >= (int32_t)s_rel[wv][rr + 1]) ++rr;
                    from[j] = (const uint8_t*)(s_delta[wv][rr] + pos + j);
                }
                uint8_t b[16];
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) b[j] = from[j] ? *from[j] : (uint8_t)0;
                if (nb == 16) { uint4 v; __builtin_memcpy(&v, b, 16); *reinterpret_cast<uint4*>(dst + pos) = v; }
                else for (uint32_t j = 0; j < nb; ++j) dst[pos + j] = b[j];
            }
        }
    }
}

int main() {
    const uint64_t N = 8ull << 30;
    uint8_t *a, *b;
    CK(hipMalloc(&a, N + 4096));
    CK(hipMalloc(&b, N + 4096));
    CK(hipMemset(a, 1, N + 4096));
    CK(hipMemset(b, 0, N + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, auto&& launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        printf("%-58s %8.3f ms  %7.1f GB/s (read + write)\n", name, ms, bytes / ms / 1e6);
    };
    const uint64_t n16 = N / 16;
    const int grid = 256 * 8;
    char nm[128];
    for (int so : {0, 1, 4, 5, 8}) {
        snprintf(nm, sizeof nm, "B copy, source + %d (unaligned dwordx4 loads)", so);
        timeit(nm, 2.0 * N, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n16, so, 0); });
    }
    for (int d : {1, 4, 5, 8}) {
        snprintf(nm, sizeof nm, "C copy, destination + %d (unaligned dwordx4 stores)", d);
        timeit(nm, 2.0 * N, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n16, 0, d); });
    }
    for (int grid2 : {256 * 4, 256 * 8, 256 * 16, 256 * 64}) {
        snprintf(nm, sizeof nm, "A aligned copy, grid %d", grid2);
        timeit(nm, 2.0 * N, [&] { hipLaunchKernelGGL(k_copy, dim3(grid2), dim3(256), 0, 0, a, b, n16, 0, 0); });
    }
    timeit("D aligned loads + DPP funnel shift (3) + aligned stores", 2.0 * N,
           [&] { hipLaunchKernelGGL(k_copy_shift, dim3(grid), dim3(256), 0, 0, a, b, n16 - 64, 3); });
    // records
    const uint64_t nrec = N / 317;
    uint64_t* h_off = (uint64_t*)malloc(nrec * 8);
    uint8_t* h_keep = (uint8_t*)malloc(nrec);
    uint32_t* h_list = (uint32_t*)malloc(nrec * 4);
    uint64_t o = 0, nk = 0;
    uint64_t x = 88172645463325252ull;
    for (uint64_t r = 0; r < nrec; ++r) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h_keep[r] = (x % 100) < 80;
        h_off[r] = o;
        if (h_keep[r]) { o += 317; h_list[nk++] = (uint32_t)r; }
    }
    uint64_t* d_off; uint8_t* d_keep; uint32_t* d_list;
    CK(hipMalloc(&d_off, nrec * 8)); CK(hipMalloc(&d_keep, nrec)); CK(hipMalloc(&d_list, nrec * 4));
    CK(hipMemcpy(d_off, h_off, nrec * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_keep, h_keep, nrec, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_list, h_list, nk * 4, hipMemcpyHostToDevice));
    timeit("E 317-byte records, 80 % kept, 4 lanes per record", (double)N + (double)o,
           [&] { hipLaunchKernelGGL(k_rec4, dim3((unsigned)((nrec * 4 + 255) / 256)), dim3(256), 0, 0, a, b, d_off, d_keep, nrec); });
    timeit("F the same, one lane per 16 output bytes (aligned stores)", (double)N + (double)o,
           [&] { hipLaunchKernelGGL(k_out16, dim3(grid * 4), dim3(256), 0, 0, a, b, d_list, o / 16, nk); });
    // G
    {
        uint64_t* h_ssrc = (uint64_t*)malloc((nk + 1) * 8);
        uint64_t* h_soff = (uint64_t*)malloc((nk + 1) * 8);
        for (uint64_t k = 0; k < nk; ++k) { h_ssrc[k] = (uint64_t)a + 317ull * h_list[k]; h_soff[k] = 317ull * k; }
        h_soff[nk] = o; h_ssrc[nk] = 0;
        const uint64_t ntile = (o + 4095) / 4096;
        uint32_t* h_first = (uint32_t*)malloc((ntile + 1) * 4);
        for (uint64_t t = 0; t <= ntile; ++t) { uint64_t k = (4096ull * t) / 317; if (k >= nk) k = nk - 1; h_first[t] = (uint32_t)k; }
        uint64_t *d_ssrc, *d_soff; uint32_t* d_first;
        CK(hipMalloc(&d_ssrc, (nk + 1) * 8)); CK(hipMalloc(&d_soff, (nk + 1) * 8)); CK(hipMalloc(&d_first, (ntile + 1) * 4));
        CK(hipMemcpy(d_ssrc, h_ssrc, (nk + 1) * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_soff, h_soff, (nk + 1) * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_first, h_first, (ntile + 1) * 4, hipMemcpyHostToDevice));
        CK(hipMemset(b, 0, o));
        timeit("G segmented copy: LDS segment table, prefix-sum lookup", (double)N + (double)o,
               [&] { hipLaunchKernelGGL(k_seg_copy, dim3((unsigned)((ntile + 3) / 4)), dim3(256), 0, 0, a, d_ssrc, d_soff, nk, d_first, b, o, a, a + N); });
        // check against a host copy of a sample: source is all 1s -> just verify no zero byte is left
        uint8_t* hb = (uint8_t*)malloc(1 << 20);
        CK(hipMemcpy(hb, b + (o / 2 & ~15ull), 1 << 20, hipMemcpyDeviceToHost));
        uint64_t bad = 0;
        for (int i = 0; i < (1 << 20); ++i) bad += hb[i] != 1;
        printf("G check: %llu bytes not written in a 1 MiB sample\n", (unsigned long long)bad);
    }
    return 0;
}
