// Probe (not product code): semantics and issue rate of v_mqsad_pk_u16_u8 / v_qsad_pk_u16_u8 on gfx950, and the
// wave_shr:1 / wave_shl:1 DPP controls.  Build: hipcc --offload-arch=gfx950 -O3 qsad_probe.hip -o qsad_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_sem(const uint64_t* a, const uint32_t* b, uint64_t* o, uint32_t* o2) {
    const int i = threadIdx.x;
    o[3 * i] = __builtin_amdgcn_mqsad_pk_u16_u8(a[i], b[i], 0ull);
    o[3 * i + 1] = __builtin_amdgcn_qsad_pk_u16_u8(a[i], b[i], 0ull);
    o[3 * i + 2] = __builtin_amdgcn_mqsad_pk_u16_u8(a[i], b[i], 0x0001000200030004ull);
    o2[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)0xABCD, (int)(i * 10), 0x138, 0xf, 0xf, false);
    o2[64 + i] = (uint32_t)__builtin_amdgcn_update_dpp((int)0xABCD, (int)(i * 10), 0x130, 0xf, 0xf, false);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(uint64_t* out, uint32_t seed, int iters) {
    uint64_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
    uint32_t p = seed * 2654435761u;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 4 independent mqsad chains
            x0 = __builtin_amdgcn_mqsad_pk_u16_u8(x0, p, x0);
            x1 = __builtin_amdgcn_mqsad_pk_u16_u8(x1, p, x1);
            x2 = __builtin_amdgcn_mqsad_pk_u16_u8(x2, p, x2);
            x3 = __builtin_amdgcn_mqsad_pk_u16_u8(x3, p, x3);
        } else if (MODE == 1) {  // 4 independent qsad chains
            x0 = __builtin_amdgcn_qsad_pk_u16_u8(x0, p, x0);
            x1 = __builtin_amdgcn_qsad_pk_u16_u8(x1, p, x1);
            x2 = __builtin_amdgcn_qsad_pk_u16_u8(x2, p, x2);
            x3 = __builtin_amdgcn_qsad_pk_u16_u8(x3, p, x3);
        } else if (MODE == 2) {  // 8 simple 32-bit VALU ops (xor/add on the halves) as the reference rate
            uint32_t a0 = (uint32_t)x0, a1 = (uint32_t)(x0 >> 32), b0 = (uint32_t)x1, b1 = (uint32_t)(x1 >> 32);
            uint32_t c0 = (uint32_t)x2, c1 = (uint32_t)(x2 >> 32), d0 = (uint32_t)x3, d1 = (uint32_t)(x3 >> 32);
            a0 = (a0 ^ p) + a1; a1 = (a1 ^ p) + b0; b0 = (b0 ^ p) + b1; b1 = (b1 ^ p) + c0;
            c0 = (c0 ^ p) + c1; c1 = (c1 ^ p) + d0; d0 = (d0 ^ p) + d1; d1 = (d1 ^ p) + a0;
            x0 = a0 | ((uint64_t)a1 << 32); x1 = b0 | ((uint64_t)b1 << 32); x2 = c0 | ((uint64_t)c1 << 32); x3 = d0 | ((uint64_t)d1 << 32);
        } else if (MODE == 3) {  // v_alignbyte + xor pairs (the round-1 prefix compare)
            uint32_t a0 = (uint32_t)x0, a1 = (uint32_t)(x0 >> 32), b0 = (uint32_t)x1, b1 = (uint32_t)(x1 >> 32);
            a0 = __builtin_amdgcn_alignbyte(a1, a0, 1) ^ p; a1 = __builtin_amdgcn_alignbyte(b0, a1, 2) ^ p;
            b0 = __builtin_amdgcn_alignbyte(b1, b0, 3) ^ p; b1 = __builtin_amdgcn_alignbyte(a0, b1, 1) ^ p;
            x0 = a0 | ((uint64_t)a1 << 32); x1 = b0 | ((uint64_t)b1 << 32);
        } else if (MODE == 4) {  // v_mul_lo_u32 x 4
            uint32_t a0 = (uint32_t)x0, a1 = (uint32_t)x1, b0 = (uint32_t)x2, b1 = (uint32_t)x3;
            a0 = a0 * p + 1; a1 = a1 * p + 1; b0 = b0 * p + 1; b1 = b1 * p + 1;
            x0 = a0; x1 = a1; x2 = b0; x3 = b1;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}

template <int MODE>
static double rate(const char* name, int ops_per_iter) {
    uint64_t* d;
    const int blocks = 256 * 8, iters = 20000;
    hipMalloc(&d, (size_t)blocks * 256 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_rate<MODE><<<blocks, 256>>>(d, 1, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_rate<MODE><<<blocks, 256>>>(d, 7, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double winstr = (double)blocks * 4 * iters * ops_per_iter;  // wave instructions
    // 256 CUs x 4 SIMDs; clock unknown -> report wave-instructions per SIMD per microsecond
    printf("%-28s %8.3f ms  %.1f wave-instr/SIMD/us\n", name, ms, winstr / 1024.0 / (ms * 1e3));
    hipFree(d);
    return ms;
}

int main() {
    uint64_t ha[64];
    uint32_t hb[64];
    const char* text = "ACGTTGCAAGCTACGTNNNN";
    for (int i = 0; i < 64; ++i) { memcpy(&ha[i], text + (i % 8), 8); memcpy(&hb[i], "TTGC", 4); }
    memcpy(&hb[1], "TT\0C", 4);   // masked byte in the reference (src1)?
    memcpy(&hb[2], "GTT\0", 4);
    memcpy(&hb[3], "\0\0\0\0", 4);
    uint64_t *da, *dout; uint32_t *db, *do2;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, 64 * 3 * 8); hipMalloc(&do2, 128 * 4);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(da, db, dout, do2);
    uint64_t ho[64 * 3]; uint32_t ho2[128];
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(ho2, do2, sizeof ho2, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) {
        printf("lane %d text=%.8s ref=%02x%02x%02x%02x  mqsad=%016llx qsad=%016llx mqsad+acc=%016llx\n", i, text + (i % 8),
               hb[i] & 255, (hb[i] >> 8) & 255, (hb[i] >> 16) & 255, hb[i] >> 24, (unsigned long long)ho[3 * i],
               (unsigned long long)ho[3 * i + 1], (unsigned long long)ho[3 * i + 2]);
    }
    printf("wave_shr:1 lanes 0,1,2,15,16,17,31,32,33,63: %u %u %u %u %u %u %u %u %u %u\n", ho2[0], ho2[1], ho2[2], ho2[15], ho2[16], ho2[17], ho2[31], ho2[32], ho2[33], ho2[63]);
    printf("wave_shl:1 lanes 0,1,14,15,16,31,32,62,63: %u %u %u %u %u %u %u %u %u\n", ho2[64], ho2[65], ho2[64 + 14], ho2[64 + 15], ho2[64 + 16], ho2[64 + 31], ho2[64 + 32], ho2[64 + 62], ho2[64 + 63]);
    rate<2>("simple valu x16 (xor+add)", 16);
    rate<0>("mqsad_pk_u16_u8 x4", 4);
    rate<1>("qsad_pk_u16_u8 x4", 4);
    rate<3>("alignbyte+xor x8", 8);
    rate<4>("mul_lo_u32+add x8(mad?)", 8);
    return 0;
}
