// probe: six output streams of 16-byte pieces at byte-unaligned positions -- (a) stored as they are (unaligned 16-byte global
// stores), (b) staged through LDS (unaligned ds_write_b128 of the piece, aligned ds_read_b128 of a chunk) and stored as
// aligned 16-byte chunks.  Also checks that an unaligned ds_write_b128 lands byte-exactly on gfx950.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint4 __attribute__((aligned(1))) u4u;
__global__ __launch_bounds__(256) void k_direct(const uint8_t* src, uint8_t* dst, size_t n48, size_t stream_bytes, int shift) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n48; i += stride) {
        const uint4 a = *reinterpret_cast<const u4u*>(src + i * 48);
        const uint4 b = *reinterpret_cast<const u4u*>(src + i * 48 + 16);
        const uint4 c = *reinterpret_cast<const u4u*>(src + i * 48 + 32);
        uint4 v[6] = {a, b, c, make_uint4(a.y, b.z, c.w, a.x), make_uint4(b.x, c.y, a.z, b.w), make_uint4(c.x, a.w, b.y, c.z)};
        for (int k = 0; k < 6; ++k) *reinterpret_cast<u4u*>(dst + (size_t)k * stream_bytes + shift * (k + 1) + i * 16) = v[k];
    }
}
// the same bytes to the same places, through LDS: the wave's 64 pieces of stream k begin at byte shift*(k+1) of a 1 KiB + 16
// window; lanes then store the aligned chunks (chunk 64 by lane 0; the bytes outside the pieces are whatever LDS held --
// the probe measures rates, the edges of a real kernel are its own business)
__global__ __launch_bounds__(256) void k_staged(const uint8_t* src, uint8_t* dst, size_t n48, size_t stream_bytes, int shift) {
    __shared__ __attribute__((aligned(16))) uint8_t s[4][1024 + 32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t* st = s[wave];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n48; i += stride) {   // (n48 is a multiple of the grid: whole waves)
        const uint4 a = *reinterpret_cast<const u4u*>(src + i * 48);
        const uint4 b = *reinterpret_cast<const u4u*>(src + i * 48 + 16);
        const uint4 c = *reinterpret_cast<const u4u*>(src + i * 48 + 32);
        uint4 v[6] = {a, b, c, make_uint4(a.y, b.z, c.w, a.x), make_uint4(b.x, c.y, a.z, b.w), make_uint4(c.x, a.w, b.y, c.z)};
        const size_t i0 = i - lane;  // the wave's first piece
        for (int k = 0; k < 6; ++k) {
            const int m = shift * (k + 1);
            *reinterpret_cast<u4u*>(st + m + lane * 16) = v[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint8_t* base = dst + (size_t)k * stream_bytes + i0 * 16;  // 16-byte aligned
            const uint4 w = *reinterpret_cast<const uint4*>(st + lane * 16);
            *reinterpret_cast<uint4*>(base + lane * 16) = w;
            // (the wave's 65th chunk is left out of the measurement: 64 aligned chunks against 64 unaligned pieces)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}
__global__ void k_check(uint8_t* out) {  // one wave: piece of lane l = 16 bytes of value l, at byte 3 + 17 l  -> dump
    __shared__ __attribute__((aligned(16))) uint8_t s[2048];
    const int lane = threadIdx.x;
    for (int x = lane; x < 2048; x += 64) s[x] = 0xEE;
    __syncthreads();
    uint4 v = make_uint4(0x01010101u * lane, 0x01010101u * lane, 0x01010101u * lane, 0x01010101u * lane);
    *reinterpret_cast<u4u*>(s + 3 + 17 * lane) = v;
    __syncthreads();
    for (int x = lane; x < 2048; x += 64) out[x] = s[x];
}
template <class F> static float timed(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < 3; ++r) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 3;
}
int main() {
    const size_t GB = 1ull << 30, nin = 16 * GB, nout = 34 * GB;
    uint8_t *src, *dst;
    if (hipMalloc(&src, nin + 256) != hipSuccess || hipMalloc(&dst, nout + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, nin); hipMemset(dst, 0, nout);
    {
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dst);
        std::vector<uint8_t> h(2048);
        hipMemcpy(h.data(), dst, 2048, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int x = 0; x < 2048; ++x) {
            int want = 0xEE;
            if (x >= 3 && (x - 3) / 17 < 64 && (x - 3) % 17 < 16) want = (x - 3) / 17;
            if (h[x] != want) ++bad;
        }
        printf("unaligned ds_write_b128: %d wrong bytes of 2048\n", bad);
    }
    for (int grid : {4096, 16384}) {
        const size_t n48 = ((15 * GB) / 48 / ((size_t)grid * 256)) * ((size_t)grid * 256), sb = n48 * 16 + 4096;
        for (int sh : {0, 1}) {
            float ms = timed([&] { hipLaunchKernelGGL(k_direct, dim3(grid), dim3(256), 0, 0, src, dst, n48, sb, sh); });
            printf("direct grid %5d shift %d: %7.3f ms  %7.1f GB/s total\n", grid, sh, ms, 3.0 * n48 * 48 / ms / 1e6);
            ms = timed([&] { hipLaunchKernelGGL(k_staged, dim3(grid), dim3(256), 0, 0, src, dst, n48, sb, sh); });
            printf("staged grid %5d shift %d: %7.3f ms  %7.1f GB/s total\n", grid, sh, ms, 3.0 * n48 * 48 / ms / 1e6);
        }
    }
    return 0;
}
