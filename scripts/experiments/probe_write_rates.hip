// probe: what does HBM give a kernel that writes more than it reads?  (translate -f 6 reads 1 byte per 2 it writes)
//   W   write only (16 bytes per lane, aligned / shifted by 1 byte)
//   R1W2  read 16 bytes per lane, write them to two places (aligned / destinations shifted by 1 and 5 bytes)
//   R1W6  read 8 bytes, write 6 x 2.67 .. as six streams of 16-byte stores (six output streams like the six frames)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint4 __attribute__((aligned(1))) u4u;
__global__ void k_w(uint8_t* dst, size_t n16, int shift) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 v = make_uint4((uint32_t)i, 2, 3, 4);
    for (; i < n16; i += stride) *reinterpret_cast<u4u*>(dst + shift + i * 16) = v;
}
__global__ void k_r1w2(const uint8_t* src, uint8_t* d0, uint8_t* d1, size_t n16, int s0, int s1) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + i * 16);
        *reinterpret_cast<u4u*>(d0 + s0 + i * 16) = v;
        *reinterpret_cast<u4u*>(d1 + s1 + i * 16) = v;
    }
}
// six output streams, each a third of the input's length: lane reads 48 bytes, writes 16 to each of six streams
__global__ void k_r1w6(const uint8_t* src, uint8_t* dst, size_t n48, size_t stream_bytes, int shift) {
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
    for (int grid : {2048, 8192, 32768}) {
        for (int sh : {0, 1}) {
            float ms = timed([&] { hipLaunchKernelGGL(k_w, dim3(grid), dim3(256), 0, 0, dst, nout / 16 - 1, sh); });
            printf("W    grid %5d shift %d: %7.3f ms  %7.1f GB/s written\n", grid, sh, ms, nout / ms / 1e6);
        }
        for (int sh : {0, 1}) {
            float ms = timed([&] { hipLaunchKernelGGL(k_r1w2, dim3(grid), dim3(256), 0, 0, src, dst, dst + 17 * GB, nin / 16, sh, sh * 5); });
            printf("R1W2 grid %5d shift %d: %7.3f ms  %7.1f GB/s total (read %zu GB, written %zu GB)\n", grid, sh, ms, 3.0 * nin / ms / 1e6, nin / GB, 2 * nin / GB);
        }
        for (int sh : {0, 1}) {
            const size_t n48 = (15 * GB) / 48, sb = n48 * 16 + 4096;
            float ms = timed([&] { hipLaunchKernelGGL(k_r1w6, dim3(grid), dim3(256), 0, 0, src, dst, n48, sb, sh); });
            printf("R1W6 grid %5d shift %d: %7.3f ms  %7.1f GB/s total (read 15 GB, written 30 GB)\n", grid, sh, ms, 3.0 * n48 * 48 / ms / 1e6);
        }
    }
    return 0;
}
