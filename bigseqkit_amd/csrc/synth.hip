// Synthetic FASTA/FASTQ generator kernel (bench + tests input).
// One thread produces 16 consecutive bytes and stores them with one 16-byte store.
#include <hip/hip_runtime.h>

#include "synth.hpp"

namespace bsk {

namespace {
__global__ __launch_bounds__(256) void k_synth(int kind, uint64_t seed, unsigned flags, uint64_t first_record,
                                               uint8_t* __restrict__ dst, uint64_t n) {
    const uint32_t rb = synth::record_bytes(kind);
    const uint64_t nchunks = (n + 15) / 16;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks;
         c += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t off = c * 16;
        uint64_t i;
        uint32_t k, cur;
        if (kind == synth::KIND_FASTA5K_VAR) {  // records of several sizes: the record of a file byte by search over the closed-form offsets
            const uint64_t x = synth::var_offset(first_record) + off;
            i = synth::var_record_at(x);
            k = (uint32_t)(x - synth::var_offset(i));
            cur = synth::var_record_bytes(i);
        } else {
            i = first_record + off / rb;
            k = (uint32_t)(off % rb);
            cur = rb;
        }
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            w[b >> 2] |= (uint32_t)synth::byte_at(kind, seed, flags, i, k) << ((b & 3) * 8);
            if (++k == cur) { k = 0; ++i; if (kind == synth::KIND_FASTA5K_VAR) cur = synth::var_record_bytes(i); }
        }
        if (off + 16 <= n && (((uintptr_t)dst) & 15) == 0) {
            *reinterpret_cast<uint4*>(dst + off) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int b = 0; b < 16 && off + b < n; ++b) dst[off + b] = (uint8_t)(w[b >> 2] >> ((b & 3) * 8));
        }
    }
}
}  // namespace

hipError_t launch_synth(int kind, uint64_t seed, unsigned flags, uint64_t first_record, uint8_t* dst, uint64_t n,
                        hipStream_t st) {
    const uint64_t nchunks = (n + 15) / 16;
    uint64_t blocks = (nchunks + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)blocks), dim3(256), 0, st, kind, seed, flags, first_record, dst, n);
    return hipGetLastError();
}

}  // namespace bsk
