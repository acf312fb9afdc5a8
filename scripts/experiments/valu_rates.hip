// Probe (not product code): issue rate of single VALU instructions on gfx950, 8 independent chains per lane, inline asm so
// that exactly the named instruction is measured.  Output: wave-instructions per SIMD per microsecond and the ratio to
// v_xor_b32.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o bin/valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHAIN8(ASM)                                                                                             \
    for (int i = 0; i < iters; ++i) {                                                                           \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                     \
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
                     : "v"(p), "v"(q), "s"(sp));                                                               \
    }

#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define A_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, %9\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 9\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %9\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %9\n"
#define A_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_SAD(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CNDMASK64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define A_CNDMASKW(i) "v_cmp_lt_u32 vcc, %" #i ", %9\n v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_MADU64(i) "v_mad_u64_u32 v[40:41], s[22:23], %" #i ", %8, v[40:41]\n"
#define A_MQSAD(i) "v_mqsad_pk_u16_u8 v[42:43], v[44:45], %" #i ", v[42:43]\n"
#define A_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define A_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %" #i ", %8\n"
#define A_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_DPPW(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define A_CMP(i) "v_cmp_eq_u32 vcc, %" #i ", %8\n"
#define A_READLANE(i) "v_readlane_b32 s20, %" #i ", 63\n"

template <int M>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = seed * (j + 3) + threadIdx.x;
    uint32_t p = seed * 2654435761u + threadIdx.x, q = seed ^ 0x5bd1e995u, sp = seed | 7u;
    if (M == 0) CHAIN8(A_XOR) else if (M == 1) CHAIN8(A_ADD) else if (M == 2) CHAIN8(A_MULLO) else if (M == 3) CHAIN8(A_MULHI)
    else if (M == 4) CHAIN8(A_MUL24) else if (M == 5) CHAIN8(A_MAD24) else if (M == 6) CHAIN8(A_DOT4) else if (M == 7) CHAIN8(A_ANDOR)
    else if (M == 8) CHAIN8(A_OR3) else if (M == 9) CHAIN8(A_MIN3) else if (M == 10) CHAIN8(A_PERM) else if (M == 11) CHAIN8(A_ALIGNBIT)
    else if (M == 12) CHAIN8(A_ALIGNBYTE) else if (M == 13) CHAIN8(A_BFE) else if (M == 14) CHAIN8(A_BFI) else if (M == 15) CHAIN8(A_LSHLOR)
    else if (M == 16) CHAIN8(A_LSHLADD) else if (M == 17) CHAIN8(A_XAD) else if (M == 18) CHAIN8(A_ADD3) else if (M == 19) CHAIN8(A_SAD)
    else if (M == 20) CHAIN8(A_CNDMASK) else if (M == 21) CHAIN8(A_BCNT) else if (M == 22) CHAIN8(A_MBCNT) else if (M == 23) CHAIN8(A_DPP)
    else if (M == 24) CHAIN8(A_DPPW) else if (M == 25) CHAIN8(A_PKADD) else if (M == 26) CHAIN8(A_LSHR) else if (M == 27) CHAIN8(A_CMP)
    else if (M == 28) CHAIN8(A_READLANE) else if (M == 29) CHAIN8(A_CNDMASK64) else if (M == 30) CHAIN8(A_CNDMASKW)
    else if (M == 31) CHAIN8(A_MADU64) else if (M == 32) CHAIN8(A_MQSAD)
    uint32_t r = 0;
    for (int j = 0; j < 8; ++j) r ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static double base = 0;
template <int M>
static void run(const char* name, int waves_per_simd) {
    uint32_t* d;
    const int blocks = 256 * waves_per_simd, iters = 4000;  // 4 waves per block -> one block per CU per "waves_per_simd"
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<M><<<blocks, 256>>>(d, 1, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<M><<<blocks, 256>>>(d, 7, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double rate = (double)blocks * 4 * iters * 8 / 1024.0 / (ms * 1e3);
    if (M == 0) base = rate;
    printf("%-22s %7.1f wave-instr/SIMD/us   x%.2f of v_xor\n", name, rate, rate / base);
    hipFree(d);
}

int main() {
    for (int w : {8}) {
        printf("-- %d waves per SIMD\n", w);
        run<0>("v_xor_b32", w); run<1>("v_add_u32", w); run<2>("v_mul_lo_u32", w); run<3>("v_mul_hi_u32", w);
        run<4>("v_mul_u32_u24", w); run<5>("v_mad_u32_u24", w); run<6>("v_dot4_u32_u8", w); run<7>("v_and_or_b32", w);
        run<8>("v_or3_b32", w); run<9>("v_min3_u32", w); run<10>("v_perm_b32", w); run<11>("v_alignbit_b32", w);
        run<12>("v_alignbyte_b32", w); run<13>("v_bfe_u32", w); run<14>("v_bfi_b32", w); run<15>("v_lshl_or_b32", w);
        run<16>("v_lshl_add_u32", w); run<17>("v_xad_u32", w); run<18>("v_add3_u32", w); run<19>("v_sad_u8", w);
        run<20>("v_cndmask_b32", w); run<21>("v_bcnt_u32_b32", w); run<22>("v_mbcnt_lo_u32_b32", w); run<23>("dpp row_shr:1", w);
        run<24>("dpp wave_shr:1", w); run<25>("v_pk_add_u16", w); run<26>("v_lshrrev_b32", w); run<27>("v_cmp_eq_u32", w);
        run<28>("v_readlane_b32", w); run<29>("v_cndmask_e64 sgpr mask", w); run<30>("v_cmp+v_cndmask (2 instr)", w);
        run<31>("v_mad_u64_u32 (1 chain)", w); run<32>("v_mqsad_pk_u16_u8", w);
    }
    return 0;
}
