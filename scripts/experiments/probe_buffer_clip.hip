// probe: how do raw-buffer stores on gfx950 clip at the ends of the resource?  (hipcc --offload-arch=gfx950 -O2 -o probe_buffer_clip ...)
// one lane stores 16 bytes (0x11 0x22 ...) at byte offset `off` of a 32-byte resource that sits 64 bytes inside a 256-byte
// array filled with 0xEE; printed: which bytes of the array changed.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint8_t* arr, int off, int nrec, int mode) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(arr + 64, 0, nrec, 0x00020000);
    u32x4 v = {0x14131211u, 0x24232221u, 0x34333231u, 0x44434241u};
    if (mode == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
    else if (mode == 1) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0x77, r, off, 0, 0);
    else if (mode == 2) __builtin_amdgcn_raw_buffer_store_b32(0x54535251u, r, off, 0, 0);
}
int main() {
    uint8_t* d;
    hipMalloc(&d, 256);
    const int offs[] = {-20, -16, -15, -13, -12, -9, -4, -3, -1, 0, 1, 3, 13, 15, 16, 17, 19, 20, 21, 23, 28, 29, 31, 32, 33, 40};
    for (int mode = 0; mode < 3; ++mode)
        for (int off : offs) {
            hipMemset(d, 0xEE, 256);
            hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, off, 32, mode);
            std::vector<uint8_t> h(256);
            hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost);
            printf("mode %d off %3d: changed array bytes (relative to the resource base):", mode, off);
            for (int i = 0; i < 256; ++i) if (h[i] != 0xEE) printf(" %d=%02x", i - 64, h[i]);
            printf("\n");
        }
    return 0;
}
