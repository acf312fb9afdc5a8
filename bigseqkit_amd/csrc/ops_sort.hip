// see ops_sort.hpp
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "ops_sort.hpp"
#include "text_dev.hpp"

namespace bsk {
namespace {

__device__ __forceinline__ bool in_set256(const uint32_t* s, uint8_t b) { return (s[b >> 5] >> (b & 31)) & 1u; }

__global__ void k_iota(uint32_t* __restrict__ perm, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = (uint32_t)i;
}

// where the string key of record i lives: head bytes [off, off + len) for modes 0 / 1, the sequence for mode 2
__device__ __forceinline__ uint32_t key_span(const uint8_t* __restrict__ buf, const RecordTable& t, const SortParams& P,
                                             uint64_t i, uint32_t* off) {
    *off = 0;
    if (P.mode == 2) {
        const uint32_t L = t.l_seq[i];
        return (P.prefix_len == 0 || L <= P.prefix_len) ? L : P.prefix_len;  // sort.go:74-87
    }
    const uint8_t* h = buf + t.start[i] + 1;
    const uint32_t lh = t.l_head[i];
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    if (P.mode == 1) return hl;                                  // record.Name
    return id_span_rec(t, i, h, hl, P.id_mode, off, P.buf_end);         // record.ID
}

// Natural order (natsort.Compare, PARITY.md SORT): the key is cut into runs of digits and runs of other bytes; digit runs
// compare as integers, other runs as strings, a key that runs out first comes first.  Rewritten so that plain byte order
// gives the same result:  digit run -> '0', number of significant digits, the significant digits;  other run -> its
// bytes (lower-cased with -i) and a 0 terminator.  One thread per record (keys are IDs / headers: short).
template <bool WRITE>
__device__ __forceinline__ uint32_t natural_key(const uint8_t* __restrict__ k, uint32_t len, bool fold, uint8_t* __restrict__ o) {
    uint32_t n = 0, i = 0;
    while (i < len) {
        if (k[i] >= '0' && k[i] <= '9') {
            uint32_t j = i;
            while (j < len && k[j] >= '0' && k[j] <= '9') ++j;
            uint32_t z = i;
            while (z + 1 < j && k[z] == '0') ++z;  // leading zeros do not count (an all-zero run keeps one '0')
            const uint32_t nd = j - z;
            if (WRITE) { o[n] = '0'; o[n + 1] = (uint8_t)(nd > 255u ? 255u : nd); for (uint32_t q = 0; q < nd; ++q) o[n + 2 + q] = k[z + q]; }
            n += 2 + nd;
            i = j;
        } else {
            while (i < len && !(k[i] >= '0' && k[i] <= '9')) {
                uint8_t c = k[i];
                if (fold && c >= 'A' && c <= 'Z') c += 32;
                if (WRITE) o[n] = c;
                ++n;
                ++i;
            }
            if (WRITE) o[n] = 0;
            ++n;
        }
    }
    return n;
}

__global__ __launch_bounds__(256) void k_sort_natlen(const uint8_t* __restrict__ buf, RecordTable t, SortParams P,
                                                     uint32_t* __restrict__ nat_len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    uint32_t off;
    const uint32_t l = key_span(buf, t, P, i, &off);
    nat_len[i] = natural_key<false>(buf + t.start[i] + 1 + off, l, P.ignore_case != 0, nullptr);
}

__global__ __launch_bounds__(256) void k_sort_natkeys(const uint8_t* __restrict__ buf, RecordTable t, SortParams P,
                                                      const uint64_t* __restrict__ nat_off, uint8_t* __restrict__ nat) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    uint32_t off;
    const uint32_t l = key_span(buf, t, P, i, &off);
    natural_key<true>(buf + t.start[i] + 1 + off, l, P.ignore_case != 0, nat + nat_off[i]);
}

__global__ __launch_bounds__(256) void k_sort_keylen(const uint8_t* __restrict__ buf, RecordTable t, SortParams P,
                                                     uint32_t* __restrict__ key_len, uint32_t* __restrict__ max_len) {
    __shared__ unsigned int s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t.n) {
        uint32_t off;
        const uint32_t l = P.nat ? (uint32_t)(P.nat_off[i + 1] - P.nat_off[i]) : key_span(buf, t, P, i, &off);
        key_len[i] = l;
        atomicMax(&s_max, l);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_max) atomicMax(max_len, s_max);
}

__global__ __launch_bounds__(256) void k_sort_chunk(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt,
                                                    SortParams P, const uint32_t* __restrict__ key_len,
                                                    const uint32_t* __restrict__ perm, uint32_t chunk,
                                                    uint64_t* __restrict__ keys, uint64_t count) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint64_t i = perm[j];
    const uint32_t len = key_len[i];
    const uint32_t b0 = chunk * 8u;
    uint64_t key = 0;
    if (b0 < len) {
        const uint32_t nb = len - b0 < 8u ? len - b0 : 8u;
        if (P.nat) {
            const uint8_t* h = P.nat + P.nat_off[i] + b0;  // already folded
            for (uint32_t k = 0; k < nb; ++k) key |= (uint64_t)h[k] << (56u - 8u * k);
        } else if (P.mode == 2) {
            const Text T = text_of(buf, t, tt, i);
            for (uint32_t k = 0; k < nb; ++k) {
                uint8_t c = T.at(b0 + k);
                if (P.ignore_case && c >= 'A' && c <= 'Z') c += 32;
                key |= (uint64_t)c << (56u - 8u * k);
            }
        } else {
            uint32_t off;
            key_span(buf, t, P, i, &off);
            const uint8_t* h = buf + t.start[i] + 1 + off + b0;
            for (uint32_t k = 0; k < nb; ++k) {
                uint8_t c = h[k];
                if (P.ignore_case && c >= 'A' && c <= 'Z') c += 32;
                key |= (uint64_t)c << (56u - 8u * k);
            }
        }
    }
    keys[j] = key;
}

// 16 lanes per record for the gap count of -b
__global__ __launch_bounds__(256) void k_sort_intkeys(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt,
                                                      SortParams P, uint64_t* __restrict__ keys) {
    constexpr int G = 16;
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const uint32_t gl = threadIdx.x % G;
    const bool live = i < t.n;
    const uint64_t ii = live ? i : 0;
    const uint32_t L = live ? t.l_seq[ii] : 0;
    uint32_t gaps = 0;
    if (P.mode == 4) {
        const Text T = text_of(buf, t, tt, ii);
        for (uint32_t k = gl; k < L; k += G) gaps += in_set256(P.gap_set, T.at(k)) ? 1u : 0u;
#pragma unroll
        for (int d = G / 2; d >= 1; d >>= 1) gaps += (uint32_t)__shfl_xor((int)gaps, d, G);
    }
    if (live && gl == 0) keys[i] = (uint64_t)(L - gaps);  // Seq.Bases(gapLetters) = len - gaps
}

__global__ void k_sort_gather(const uint32_t* __restrict__ out_len, const uint32_t* __restrict__ perm, uint64_t n,
                              uint32_t* __restrict__ len_perm) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) len_perm[j] = out_len[perm[j]];
}

__global__ void k_sort_scatter(const uint64_t* __restrict__ off_perm, const uint32_t* __restrict__ perm, uint64_t n,
                               uint64_t* __restrict__ out_off) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out_off[perm[j]] = off_perm[j];
    if (j == 0) out_off[n] = off_perm[n];
}

inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

hipError_t launch_sort_natlen(const uint8_t* buf, const RecordTable& t, const SortParams& P, uint32_t* nat_len, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sort_natlen, grid_for(t.n), dim3(256), 0, st, buf, t, P, nat_len);
    return hipGetLastError();
}

hipError_t launch_sort_natkeys(const uint8_t* buf, const RecordTable& t, const SortParams& P, const uint64_t* nat_off, uint8_t* nat,
                               hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sort_natkeys, grid_for(t.n), dim3(256), 0, st, buf, t, P, nat_off, nat);
    return hipGetLastError();
}

hipError_t launch_sort_iota(uint32_t* perm, uint64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_iota, grid_for(n), dim3(256), 0, st, perm, n);
    return hipGetLastError();
}

hipError_t launch_sort_keylen(const uint8_t* buf, const RecordTable& t, const SortParams& P, uint32_t* key_len,
                              uint32_t* max_len, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sort_keylen, grid_for(t.n), dim3(256), 0, st, buf, t, P, key_len, max_len);
    return hipGetLastError();
}

hipError_t launch_sort_chunk(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const SortParams& P,
                             const uint32_t* key_len, const uint32_t* perm, uint32_t chunk, uint64_t* keys, hipStream_t st,
                             uint64_t count) {
    if (count == ~0ull) count = t.n;
    if (count == 0) return hipSuccess;
    TextTable d{tt.text_w, tt.lin_off, tt.lin};
    hipLaunchKernelGGL(k_sort_chunk, grid_for(count), dim3(256), 0, st, buf, t, d, P, key_len, perm, chunk, keys, count);
    return hipGetLastError();
}

// ---- ties after the two leading chunks (sort_run_device) ----------------------------------------------------------
namespace {
// position j of the order by (chunk 0, chunk 1): tied[j] = 1 when a neighbour has the same 16 key bytes, start[j] = 1 for
// the first position of such a run
__global__ __launch_bounds__(256) void k_sort_tie_flags(const uint64_t* __restrict__ k0, const uint64_t* __restrict__ k1, uint64_t n,
                                                        uint32_t* __restrict__ tied, uint32_t* __restrict__ start) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const bool prev = j > 0 && k0[j - 1] == k0[j] && k1[j - 1] == k1[j];
    const bool next = j + 1 < n && k0[j + 1] == k0[j] && k1[j + 1] == k1[j];
    tied[j] = (prev || next) ? 1u : 0u;
    start[j] = (!prev && next) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_sort_tie_gather(const uint32_t* __restrict__ tied, const uint64_t* __restrict__ rank,
                                                         const uint64_t* __restrict__ run, const uint32_t* __restrict__ start,
                                                         const uint32_t* __restrict__ perm, uint64_t n, uint32_t* __restrict__ sub_pos,
                                                         uint32_t* __restrict__ sub_perm, uint64_t* __restrict__ run_of) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || !tied[j]) return;
    const uint64_t k = rank[j];
    sub_pos[k] = (uint32_t)j;
    sub_perm[k] = perm[j];
    run_of[perm[j]] = run[j] + start[j] - 1u;  // runs before this position (its own counts from its first element on)
}
__global__ __launch_bounds__(256) void k_sort_gather_keys(const uint64_t* __restrict__ by_record, const uint32_t* __restrict__ perm,
                                                          uint64_t m, uint64_t* __restrict__ keys) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < m) keys[k] = by_record[perm[k]];
}
__global__ __launch_bounds__(256) void k_sort_tie_scatter(const uint32_t* __restrict__ sub_pos, const uint32_t* __restrict__ sub_perm,
                                                          uint64_t m, uint32_t* __restrict__ perm) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < m) perm[sub_pos[k]] = sub_perm[k];
}
}  // namespace

hipError_t launch_sort_tie_flags(const uint64_t* k0, const uint64_t* k1, uint64_t n, uint32_t* tied, uint32_t* start, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sort_tie_flags, grid_for(n), dim3(256), 0, st, k0, k1, n, tied, start);
    return hipGetLastError();
}
hipError_t launch_sort_tie_gather(const uint32_t* tied, const uint64_t* rank, const uint64_t* run, const uint32_t* start,
                                  const uint32_t* perm, uint64_t n, uint32_t* sub_pos, uint32_t* sub_perm, uint64_t* run_of, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sort_tie_gather, grid_for(n), dim3(256), 0, st, tied, rank, run, start, perm, n, sub_pos, sub_perm, run_of);
    return hipGetLastError();
}
hipError_t launch_sort_gather_keys(const uint64_t* by_record, const uint32_t* perm, uint64_t m, uint64_t* keys, hipStream_t st) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sort_gather_keys, grid_for(m), dim3(256), 0, st, by_record, perm, m, keys);
    return hipGetLastError();
}
hipError_t launch_sort_tie_scatter(const uint32_t* sub_pos, const uint32_t* sub_perm, uint64_t m, uint32_t* perm, hipStream_t st) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sort_tie_scatter, grid_for(m), dim3(256), 0, st, sub_pos, sub_perm, m, perm);
    return hipGetLastError();
}

hipError_t launch_sort_intkeys(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const SortParams& P,
                               uint64_t* keys, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    TextTable d{tt.text_w, tt.lin_off, tt.lin};
    hipLaunchKernelGGL(k_sort_intkeys, grid_for(t.n * 16), dim3(256), 0, st, buf, t, d, P, keys);
    return hipGetLastError();
}

hipError_t sort_pairs_temp_bytes(uint64_t n, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
}

hipError_t launch_sort_pairs(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                             uint32_t* vout, uint64_t n, bool descending, int end_bit, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (descending) return rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0, end_bit, st);
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0, end_bit, st);
}

hipError_t sort_pairs_bits_temp_bytes(uint64_t n, int begin_bit, int end_bit, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (size_t)n, begin_bit, end_bit, (hipStream_t)0);
}

hipError_t launch_sort_pairs_bits(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                                  uint32_t* vout, uint64_t n, int begin_bit, int end_bit, hipStream_t st) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (size_t)n, begin_bit, end_bit, st);
}

// the same with the values 0, 1, 2, ... as input (a counting iterator: no iota pass, no array to read)
hipError_t sort_pairs_bits_iota_temp_bytes(uint64_t n, int begin_bit, int end_bit, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, rocprim::counting_iterator<uint32_t>(0),
                                     (uint32_t*)nullptr, (size_t)n, begin_bit, end_bit, (hipStream_t)0);
}

hipError_t launch_sort_pairs_bits_iota(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout, uint32_t* vout, uint64_t n,
                                       int begin_bit, int end_bit, hipStream_t st) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, rocprim::counting_iterator<uint32_t>(0), vout, (size_t)n, begin_bit,
                                     end_bit, st);
}

hipError_t launch_sort_gather(const uint32_t* out_len, const uint32_t* perm, uint64_t n, uint32_t* len_perm, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sort_gather, grid_for(n), dim3(256), 0, st, out_len, perm, n, len_perm);
    return hipGetLastError();
}

hipError_t launch_sort_scatter(const uint64_t* off_perm, const uint32_t* perm, uint64_t n, uint64_t* out_off, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sort_scatter, grid_for(n), dim3(256), 0, st, off_perm, perm, n, out_off);
    return hipGetLastError();
}

}  // namespace bsk
