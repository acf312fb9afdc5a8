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

__global__ void k_group_all(const uint64_t* __restrict__ group, uint64_t n, uint64_t* __restrict__ list) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) list[i] = (group[i] << 32) | i;
}

__global__ __launch_bounds__(256) void k_count_below(const uint64_t* __restrict__ start, uint64_t n, uint64_t x,
                                                     unsigned long long* __restrict__ count) {
    __shared__ unsigned int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned int mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        mine += start[i] < x;
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(count, (unsigned long long)s_cnt);
}

__device__ __forceinline__ uint64_t lower_bound64(const uint64_t* __restrict__ a, uint64_t lo, uint64_t hi, uint64_t v) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One thread per sorted entry.  Groups are almost always two records (a read and its mate): the segment is found by a
// short walk, the binary searches are for the rare large groups.
__global__ __launch_bounds__(256) void k_pair_classify(const uint64_t* __restrict__ sorted, uint64_t n, uint32_t first2,
                                                       uint8_t* __restrict__ state, uint32_t* __restrict__ partner) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint64_t e = sorted[p], g = e >> 32;
    const uint32_t i = (uint32_t)e;
    uint64_t s = p, end = p + 1;
    int steps = 0;
    while (s > 0 && (sorted[s - 1] >> 32) == g && steps < 8) { --s; ++steps; }
    if (s > 0 && (sorted[s - 1] >> 32) == g) s = lower_bound64(sorted, 0, s, g << 32);
    steps = 0;
    while (end < n && (sorted[end] >> 32) == g && steps < 8) { ++end; ++steps; }
    if (end < n && (sorted[end] >> 32) == g) end = lower_bound64(sorted, end, n, (g + 1) << 32);
    uint64_t m1;  // members of file 1 (they sort first: their indices are below first2)
    if (end - s <= 16) {
        m1 = 0;
        for (uint64_t q = s; q < end; ++q) m1 += (uint32_t)sorted[q] < first2;
    } else {
        m1 = lower_bound64(sorted, s, end, (g << 32) | first2) - s;
    }
    const uint64_t m2 = end - s - m1;
    if (i < first2) {
        const uint64_t ord = p - s;
        if (ord < m2) { state[i] = 1; partner[i] = (uint32_t)sorted[s + m1 + ord]; }
        else state[i] = 3;
    } else {
        const uint64_t ord = p - s - m1;
        if (ord < m1) { state[i] = 2; partner[i] = (uint32_t)sorted[s + ord]; }
        else state[i] = 4;
    }
}

__global__ __launch_bounds__(256) void k_pair_totals(const uint8_t* __restrict__ state, const uint32_t* __restrict__ fmt_len,
                                                     uint64_t n, unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long s_b[4];
    __shared__ unsigned int s_c[4];
    if (threadIdx.x < 4) { s_b[threadIdx.x] = 0; s_c[threadIdx.x] = 0; }
    __syncthreads();
    unsigned long long b[4] = {0, 0, 0, 0};
    unsigned int c[4] = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = state[i];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q)
            if (k == q + 1u) { b[q] += fmt_len[i]; ++c[q]; }
    }
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q)
        if (c[q]) { atomicAdd(&s_b[q], b[q]); atomicAdd(&s_c[q], c[q]); }
    __syncthreads();
    if (threadIdx.x < 4 && s_c[threadIdx.x]) {
        atomicAdd(&totals[threadIdx.x], s_b[threadIdx.x]);
        atomicAdd(&totals[4 + threadIdx.x], (unsigned long long)s_c[threadIdx.x]);
    }
}

__global__ void k_pair_select(const uint8_t* __restrict__ state, const uint32_t* __restrict__ fmt_len, uint64_t n, uint8_t k,
                              uint32_t* __restrict__ len_k) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) len_k[i] = state[i] == k ? fmt_len[i] : 0u;
}

__global__ void k_pair_partner_len(const uint8_t* __restrict__ state, const uint32_t* __restrict__ partner,
                                   const uint32_t* __restrict__ fmt_len, uint64_t n, uint32_t* __restrict__ w) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = state[i] == 1 ? fmt_len[partner[i]] : 0u;
}

__global__ void k_pair_partner_off(const uint8_t* __restrict__ state, const uint32_t* __restrict__ partner,
                                   const uint64_t* __restrict__ offw, uint64_t n, uint64_t* __restrict__ off2) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && state[i] == 1) off2[partner[i]] = offw[i];
}

__device__ __forceinline__ uint32_t file_of(uint64_t pos, const uint64_t* __restrict__ ends, uint32_t k) {
    uint32_t f = 0;
    while (f + 1 < k && pos >= ends[f]) ++f;  // k <= 64 files, mostly 2
    return f;
}

__global__ void k_common_masks(const uint64_t* __restrict__ group, const uint64_t* __restrict__ start, uint64_t n,
                               const uint64_t* __restrict__ ends, uint32_t k, unsigned long long* __restrict__ masks) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicOr(&masks[group[i]], 1ull << file_of(start[i], ends, k));
}

__global__ void k_common_select(const uint64_t* __restrict__ group, const uint64_t* __restrict__ start, uint64_t n,
                                const uint64_t* __restrict__ ends, uint32_t k, const uint64_t* __restrict__ masks,
                                uint32_t* __restrict__ len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t full = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
    if (!(group[i] == i && start[i] < ends[0] && masks[i] == full)) len[i] = 0;
}

__global__ void k_mask_u32(uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !b[i]) a[i] = 0;
}

__global__ __launch_bounds__(256) void k_first_nonzero(const uint32_t* __restrict__ a, uint64_t n,
                                                       unsigned long long* __restrict__ first) {
    __shared__ unsigned long long s_min;
    if (threadIdx.x == 0) s_min = ~0ull;
    __syncthreads();
    unsigned long long mine = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (a[i]) { mine = i; break; }  // indices grow along the loop: the first one found is this thread's smallest
    if (mine != ~0ull) atomicMin(&s_min, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_min != ~0ull) atomicMin(first, s_min);
}

// grep --delete-matched with several patterns: masks[i] bit k = record i matches pattern k
__global__ void k_or_bit(uint32_t* __restrict__ masks, const uint32_t* __restrict__ hit, uint64_t n, uint32_t bit) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && hit[i]) masks[i] |= bit;
}
// first record at or after `from` that matches one of the remaining patterns (same reduction as k_first_nonzero)
__global__ __launch_bounds__(256) void k_first_masked(const uint32_t* __restrict__ masks, uint64_t n, uint32_t remaining,
                                                      uint64_t from, unsigned long long* __restrict__ first) {
    __shared__ unsigned long long s_min;
    if (threadIdx.x == 0) s_min = ~0ull;
    __syncthreads();
    unsigned long long mine = ~0ull;
    for (uint64_t i = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (masks[i] & remaining) { mine = i; break; }
    if (mine != ~0ull) atomicMin(&s_min, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_min != ~0ull) atomicMin(first, s_min);
}
// out_len[i] := 0 unless bit 31 of masks[i] marks the record as selected
__global__ void k_keep_selected(uint32_t* __restrict__ out_len, const uint32_t* __restrict__ masks, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(masks[i] & 0x80000000u)) out_len[i] = 0;
}

__global__ void k_keep_only(uint32_t* __restrict__ a, uint64_t n, const uint64_t* __restrict__ first) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && i != *first) a[i] = 0;
}

}  // namespace

#define BSK_GRID(n) dim3((unsigned)(((n) + 255) / 256))
hipError_t launch_group_all(const uint64_t* group, uint64_t n, uint64_t* list, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_group_all, BSK_GRID(n), dim3(256), 0, st, group, n, list);
    return hipGetLastError();
}
hipError_t launch_count_below(const uint64_t* start, uint64_t n, uint64_t x, uint64_t* count, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 4095) / 4096;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_count_below, dim3((unsigned)blocks), dim3(256), 0, st, start, n, x, (unsigned long long*)count);
    return hipGetLastError();
}
hipError_t launch_pair_classify(const uint64_t* sorted, uint64_t n, uint32_t first2, uint8_t* state, uint32_t* partner,
                                hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pair_classify, BSK_GRID(n), dim3(256), 0, st, sorted, n, first2, state, partner);
    return hipGetLastError();
}
hipError_t launch_pair_totals(const uint8_t* state, const uint32_t* fmt_len, uint64_t n, uint64_t* totals, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 4095) / 4096;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pair_totals, dim3((unsigned)blocks), dim3(256), 0, st, state, fmt_len, n, (unsigned long long*)totals);
    return hipGetLastError();
}
hipError_t launch_pair_select(const uint8_t* state, const uint32_t* fmt_len, uint64_t n, uint8_t k, uint32_t* len_k, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pair_select, BSK_GRID(n), dim3(256), 0, st, state, fmt_len, n, k, len_k);
    return hipGetLastError();
}
hipError_t launch_pair_partner_len(const uint8_t* state, const uint32_t* partner, const uint32_t* fmt_len, uint64_t n,
                                   uint32_t* w, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pair_partner_len, BSK_GRID(n), dim3(256), 0, st, state, partner, fmt_len, n, w);
    return hipGetLastError();
}
hipError_t launch_pair_partner_off(const uint8_t* state, const uint32_t* partner, const uint64_t* offw, uint64_t n,
                                   uint64_t* off2, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pair_partner_off, BSK_GRID(n), dim3(256), 0, st, state, partner, offw, n, off2);
    return hipGetLastError();
}
hipError_t launch_common_masks(const uint64_t* group, const uint64_t* start, uint64_t n, const uint64_t* file_ends, uint32_t k,
                               uint64_t* masks, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_common_masks, BSK_GRID(n), dim3(256), 0, st, group, start, n, file_ends, k, (unsigned long long*)masks);
    return hipGetLastError();
}
hipError_t launch_common_select(const uint64_t* group, const uint64_t* start, uint64_t n, const uint64_t* file_ends, uint32_t k,
                                const uint64_t* masks, uint32_t* len, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_common_select, BSK_GRID(n), dim3(256), 0, st, group, start, n, file_ends, k, masks, len);
    return hipGetLastError();
}
hipError_t launch_mask_u32(uint32_t* a, const uint32_t* b, uint64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mask_u32, BSK_GRID(n), dim3(256), 0, st, a, b, n);
    return hipGetLastError();
}
hipError_t launch_first_nonzero(const uint32_t* a, uint64_t n, uint64_t* first, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 4095) / 4096;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_first_nonzero, dim3((unsigned)blocks), dim3(256), 0, st, a, n, (unsigned long long*)first);
    return hipGetLastError();
}
hipError_t launch_or_bit(uint32_t* masks, const uint32_t* hit, uint64_t n, uint32_t bit, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_or_bit, BSK_GRID(n), dim3(256), 0, st, masks, hit, n, bit);
    return hipGetLastError();
}
hipError_t launch_first_masked(const uint32_t* masks, uint64_t n, uint32_t remaining, uint64_t from, uint64_t* first, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 4095) / 4096;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_first_masked, dim3((unsigned)blocks), dim3(256), 0, st, masks, n, remaining, from, (unsigned long long*)first);
    return hipGetLastError();
}
hipError_t launch_keep_selected(uint32_t* out_len, const uint32_t* masks, uint64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_keep_selected, BSK_GRID(n), dim3(256), 0, st, out_len, masks, n);
    return hipGetLastError();
}
hipError_t launch_keep_only(uint32_t* a, uint64_t n, const uint64_t* first, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_keep_only, BSK_GRID(n), dim3(256), 0, st, a, n, first);
    return hipGetLastError();
}
#undef BSK_GRID

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

hipError_t launch_group_sort(void* tmp, size_t tmp_bytes, const uint64_t* in, uint64_t* out, uint64_t m, hipStream_t st,
                             uint64_t records, bool index_ordered) {
    if (m == 0) return hipSuccess;
    // group < records: only the bits that can be set take part; and when the list is already in index order (the low
    // word) the stable sort by the group alone gives the (group, index) order: 4 radix passes instead of 8
    unsigned hi = 64;
    if (records) {
        unsigned b = 1;
        while (b < 32 && (records - 1) >> b) ++b;
        hi = 32 + b;
    }
    return rocprim::radix_sort_keys(tmp, tmp_bytes, in, out, (size_t)m, index_ordered ? 32u : 0u, hi, st);
}

hipError_t launch_group_ordinals(const uint64_t* sorted, uint64_t m, uint32_t* ord, hipStream_t st) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_group_ordinals, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, sorted, m, ord);
    return hipGetLastError();
}

}  // namespace bsk
