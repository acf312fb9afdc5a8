// see ops_group.hpp
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "ops_group.hpp"

namespace bsk {
namespace {

// each block owns a contiguous chunk: count, ONE atomicAdd to reserve the slots, then write
__global__ __launch_bounds__(256) void k_group_compact(const uint64_t* __restrict__ group, uint64_t n,
                                                       uint64_t* __restrict__ list, unsigned long long* __restrict__ count) {
    __shared__ unsigned int s_cnt;
    __shared__ unsigned long long s_base;
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned int mine = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) mine += group[i] != i;
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) { s_base = s_cnt ? atomicAdd(count, (unsigned long long)s_cnt) : 0ull; s_cnt = 0; }
    __syncthreads();
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
        if (group[i] != i) list[s_base + atomicAdd(&s_cnt, 1u)] = (group[i] << 32) | i;
}

__global__ __launch_bounds__(256) void k_group_ordinals(const uint64_t* __restrict__ sorted, uint64_t m,
                                                        uint32_t* __restrict__ ord) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const uint64_t e = sorted[p], g0 = e & ~0xFFFFFFFFull;
    uint64_t lo = 0, hi = p;  // first entry of this group: lower bound of g0
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (sorted[mid] < g0) lo = mid + 1; else hi = mid;
    }
    ord[(uint32_t)e] = (uint32_t)(p - lo) + 1u;
}

}  // namespace

hipError_t launch_group_compact(const uint64_t* group, uint64_t n, uint64_t* list, uint64_t* count, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 4095) / 4096;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_group_compact, dim3((unsigned)blocks), dim3(256), 0, st, group, n, list, (unsigned long long*)count);
    return hipGetLastError();
}

hipError_t group_sort_temp_bytes(uint64_t m, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_keys(nullptr, *bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)m, 0, 64, (hipStream_t)0);
}

hipError_t launch_group_sort(void* tmp, size_t tmp_bytes, const uint64_t* in, uint64_t* out, uint64_t m, hipStream_t st) {
    if (m == 0) return hipSuccess;
    return rocprim::radix_sort_keys(tmp, tmp_bytes, in, out, (size_t)m, 0, 64, st);
}

hipError_t launch_group_ordinals(const uint64_t* sorted, uint64_t m, uint32_t* ord, hipStream_t st) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_group_ordinals, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, sorted, m, ord);
    return hipGetLastError();
}

}  // namespace bsk
