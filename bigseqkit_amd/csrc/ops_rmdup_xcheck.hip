// ============================================================================
// ops_rmdup_xcheck.hip -- RmDupCheck's text comparison across ranks (round 6).
//
// The reference shuffles WHOLE records to the executor that owns their hash (GroupByKey, bigseqkit/rmdup.go:97) and there
// compares the subject text of every member of a hash group (bigseqkit-lib/rmdup.go:193-211).  Here only 24-byte tuples
// travel to the owner (ops_rmdup.hip); the owner's answer names the survivor of every tuple's group.  Until round 5 a
// duplicate whose survivor lived on another rank was dropped on the strength of its two 64-bit keys.  Now it sends its
// subject to the survivor's rank -- 24 bytes of request + the text, ~170 B per cross-rank duplicate -- which compares the
// bytes and answers one byte:
//   k_x_count / k_x_place : per destination rank, the requests of this shard and where every subject goes in the segment
//                           of its destination (block-aggregated cursors: a few thousand global atomics per launch)
//   k_x_copy              : the subjects into the send buffer, 4 lanes per request, 16 bytes per lane and step
//   k_x_compare           : survivor side, 4 lanes per request: received text against the local record's subject
//   k_x_apply             : sender side: a "differs" puts the record on the flagged list
// Flagged records (two subjects under one pair of keys) are settled exactly on the host: grouped by TEXT over all ranks,
// the lowest global index of every text survives -- what RmDupCheck's map keyed by the subject does.
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

#include "hash_dev.hpp"
#include "ops_rmdup_xcheck.hpp"
#include "rmdup_subject_dev.hpp"

namespace bsk {

namespace {

using hashdev::fold4;

__device__ __forceinline__ uint32_t rank_of(const XRanks& R, uint64_t g) {
    uint32_t r = 0;
    for (uint32_t k = 1; k < R.world; ++k) r += g >= R.base[k];
    return r;
}

// a duplicate of this shard whose survivor lives on another rank
__device__ __forceinline__ bool crosses(const XRanks& R, uint8_t reply, uint64_t surv) {
    return !reply && (surv < R.base[R.rank] || surv >= R.base[R.rank + 1]);
}

__device__ __forceinline__ bool contiguous(const Subject& s) { return s.seq ? s.T.W == 0u : true; }
__device__ __forceinline__ const uint8_t* base_ptr(const Subject& s) { return s.seq ? s.T.p : s.h; }

template <bool PLACE>
__global__ __launch_bounds__(256) void k_x_count_place(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt, RmDupParams P,
                                                       const uint64_t* __restrict__ send, const uint8_t* __restrict__ reply,
                                                       const uint64_t* __restrict__ surv, uint64_t n, XRanks R,
                                                       unsigned long long* __restrict__ g_cnt, unsigned long long* __restrict__ g_bytes,
                                                       uint64_t* __restrict__ req) {
    __shared__ unsigned int s_cnt[XCHECK_MAX_WORLD];
    __shared__ unsigned long long s_bytes[XCHECK_MAX_WORLD];
    __shared__ unsigned long long s_base_cnt[XCHECK_MAX_WORLD], s_base_bytes[XCHECK_MAX_WORLD];
    for (uint32_t o = threadIdx.x; o < R.world; o += blockDim.x) { s_cnt[o] = 0; s_bytes[o] = 0ull; }
    __syncthreads();
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    const uint64_t mybase = R.base[R.rank];
    for (uint64_t p = lo + threadIdx.x; p < hi; p += blockDim.x) {
        const uint64_t s = surv[p];
        if (!crosses(R, reply[p], s)) continue;
        const uint64_t i = send[3 * p + 2] - mybase;
        const Subject a = subject_of(buf, t, tt, P, i);
        const uint32_t d = rank_of(R, s);
        atomicAdd(&s_cnt[d], 1u);
        atomicAdd(&s_bytes[d], (unsigned long long)a.len);
    }
    __syncthreads();
    if (!PLACE) {
        for (uint32_t o = threadIdx.x; o < R.world; o += blockDim.x)
            if (s_cnt[o]) { atomicAdd(&g_cnt[o], (unsigned long long)s_cnt[o]); atomicAdd(&g_bytes[o], s_bytes[o]); }
        return;
    }
    for (uint32_t o = threadIdx.x; o < R.world; o += blockDim.x) {  // reserve the block's slots and bytes per destination
        s_base_cnt[o] = s_cnt[o] ? atomicAdd(&g_cnt[o], (unsigned long long)s_cnt[o]) : 0ull;
        s_base_bytes[o] = s_cnt[o] ? atomicAdd(&g_bytes[o], s_bytes[o]) : 0ull;
        s_cnt[o] = 0;
        s_bytes[o] = 0ull;
    }
    __syncthreads();
    for (uint64_t p = lo + threadIdx.x; p < hi; p += blockDim.x) {
        const uint64_t s = surv[p];
        if (!crosses(R, reply[p], s)) continue;
        const uint64_t i = send[3 * p + 2] - mybase;
        const Subject a = subject_of(buf, t, tt, P, i);
        const uint32_t d = rank_of(R, s);
        const uint64_t pos = s_base_cnt[d] + atomicAdd(&s_cnt[d], 1u);
        const uint64_t off = s_base_bytes[d] + atomicAdd(&s_bytes[d], (unsigned long long)a.len);
        req[3 * pos] = s;
        req[3 * pos + 1] = off;
        req[3 * pos + 2] = (uint64_t)a.len | (i << 32);
    }
}

__global__ __launch_bounds__(256) void k_x_copy(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt, RmDupParams P,
                                                const uint64_t* __restrict__ req, uint64_t m, XRanks R, XFrom seg,
                                                uint8_t* __restrict__ text) {
    const uint64_t j = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint32_t gl = threadIdx.x & 3u;
    if (j >= m) return;
    const uint64_t s = req[3 * j], off = req[3 * j + 1], w = req[3 * j + 2];
    const uint32_t len = (uint32_t)w;
    Subject a = subject_of(buf, t, tt, P, w >> 32);
    a.fold = false;  // the text travels as it is: the comparison folds both sides
    uint8_t* dst = text + seg.byte_start[rank_of(R, s)] + off;
    if (contiguous(a)) {
        const uint8_t* src = base_ptr(a);
        for (uint32_t q = 16u * gl; q + 16u <= len; q += 64u) {
            uint4 v;
            __builtin_memcpy(&v, src + q, 16);
            __builtin_memcpy(dst + q, &v, 16);
        }
        if (gl == 0u)
            for (uint32_t q = len & ~15u; q < len; ++q) dst[q] = src[q];
    } else {
        for (uint32_t q = gl; q < len; q += 4u) dst[q] = a.at(q);
    }
}

template <bool FOLD>
__device__ __forceinline__ uint32_t diff16(const uint8_t* pa, const uint8_t* pb) {
    uint4 x, y;
    __builtin_memcpy(&x, pa, 16);
    __builtin_memcpy(&y, pb, 16);
    if (FOLD) {
        x.x = fold4(x.x); x.y = fold4(x.y); x.z = fold4(x.z); x.w = fold4(x.w);
        y.x = fold4(y.x); y.y = fold4(y.y); y.z = fold4(y.z); y.w = fold4(y.w);
    }
    return (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
}

// text ta[0, la) against subject b, by the lanes gl = 0..3 of a quad; != 0: they differ (each lane holds a part of the answer)
template <bool FOLD>
__device__ __forceinline__ uint32_t quad_diff(const uint8_t* ta, uint32_t la, const Subject& b, uint32_t gl) {
    uint32_t diff = la ^ b.len;
    if (diff) return diff;
    if (contiguous(b)) {
        const uint8_t* pb = base_ptr(b);
        for (uint32_t q = 16u * gl; q + 16u <= la; q += 64u) diff |= diff16<FOLD>(ta + q, pb + q);  // (no early exit: the loads do not wait for each other)
        if (gl == 3u) {
            if (la >= 16u) { if (la & 15u) diff |= diff16<FOLD>(ta + la - 16u, pb + la - 16u); }  // the tail: the last 16 bytes once more
            else for (uint32_t q = 0; q < la; ++q) {
                uint8_t ca = ta[q], cb = pb[q];
                if (FOLD) { ca = lower8(ca); cb = lower8(cb); }
                diff |= (uint32_t)(ca ^ cb);
            }
        }
    } else {
        for (uint32_t q = gl; q < la; q += 4u) {
            uint8_t ca = ta[q];
            if (FOLD) ca = lower8(ca);
            diff |= (uint32_t)(ca ^ b.at(q));  // (b.at folds by itself)
        }
    }
    return diff;
}

template <bool FOLD>
__global__ __launch_bounds__(256) void k_x_compare(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt, RmDupParams P,
                                                   const uint64_t* __restrict__ req, uint64_t m, XFrom from,
                                                   const uint8_t* __restrict__ text, uint64_t base, uint8_t* __restrict__ verdict,
                                                   uint64_t* __restrict__ status) {
    const uint64_t j = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint32_t gl = threadIdx.x & 3u;
    if (j >= m) return;  // (a quad leaves together)
    const uint64_t s = req[3 * j], off = req[3 * j + 1];
    const uint32_t la = (uint32_t)req[3 * j + 2];
    uint32_t p = 0;
    for (uint32_t k = 1; k < from.world; ++k) p += j >= from.req_start[k];
    if (s < base || s - base >= t.n || from.byte_start[p] + off + la > from.byte_start[p + 1]) {
        if (gl == 0u) {
            verdict[j] = 2;
            atomicOr((unsigned long long*)&status[0], (unsigned long long)ERR_HASH_COLLISION);
        }
        return;
    }
    const Subject b = subject_of(buf, t, tt, P, s - base);
    uint32_t diff = quad_diff<FOLD>(text + from.byte_start[p] + off, la, b, gl);
    diff |= (uint32_t)__shfl_xor((int)diff, 1, 64);
    diff |= (uint32_t)__shfl_xor((int)diff, 2, 64);
    if (gl == 0u) verdict[j] = diff == 0u ? 1 : 0;
}

__device__ __forceinline__ void list_push(uint32_t* list, uint32_t cap, uint32_t i) {
    const uint32_t at = atomicAdd(&list[0], 1u);  // (rare: no aggregation)
    if (at < cap) list[1u + at] = i;
}

__global__ __launch_bounds__(256) void k_x_apply(const uint64_t* __restrict__ req, const uint8_t* __restrict__ verdict, uint64_t m,
                                                 uint32_t* __restrict__ list, uint32_t cap) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    if (verdict[j] != 1) list_push(list, cap, (uint32_t)(req[3 * j + 2] >> 32));
}

// two subjects of ONE shard, one lane: byte by byte through the accessors (every layout, every subject kind) -- the listing
// kernels run on the duplicates of a shard whose fast comparison has raised its flag, or on subjects it does not read
__device__ bool subjects_equal(const Subject& a, const Subject& b) {
    if (a.len != b.len) return false;
    if (contiguous(a) && contiguous(b) && !a.fold) {
        const uint8_t *pa = base_ptr(a), *pb = base_ptr(b);
        uint32_t q = 0, diff = 0;
        for (; q + 16u <= a.len; q += 16u) diff |= diff16<false>(pa + q, pb + q);
        for (; q < a.len; ++q) diff |= (uint32_t)(pa[q] ^ pb[q]);
        return diff == 0u;
    }
    for (uint32_t q = 0; q < a.len; ++q)
        if (a.at(q) != b.at(q)) return false;
    return true;
}

__global__ __launch_bounds__(256) void k_x_local_list(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt, RmDupParams P,
                                                      const uint64_t* __restrict__ send, const uint8_t* __restrict__ reply,
                                                      const uint64_t* __restrict__ surv, uint64_t n, uint64_t base,
                                                      uint32_t* __restrict__ list, uint32_t cap, unsigned long long* __restrict__ n_pairs) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool pair = false;
    if (p < n) {
        const uint64_t i = send[3 * p + 2] - base, s = surv[p];
        pair = !reply[p] && s >= base && s < base + n && s - base != i;
        if (pair && !subjects_equal(subject_of(buf, t, tt, P, i), subject_of(buf, t, tt, P, s - base))) list_push(list, cap, (uint32_t)i);
    }
    const uint64_t bm = __ballot(pair);
    if (n_pairs && bm && (threadIdx.x & 63u) == 0u) atomicAdd(n_pairs, (unsigned long long)__popcll(bm));
}

__global__ __launch_bounds__(256) void k_x_first_list(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt, RmDupParams P,
                                                      const uint32_t* __restrict__ first, uint32_t* __restrict__ list, uint32_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint64_t f = first[i];
    if (f != i && !subjects_equal(subject_of(buf, t, tt, P, i), subject_of(buf, t, tt, P, f))) list_push(list, cap, (uint32_t)i);
}

__global__ __launch_bounds__(256) void k_x_subject_len(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt, RmDupParams P,
                                                       const uint32_t* __restrict__ list, uint32_t m, uint32_t* __restrict__ len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) len[j] = subject_of(buf, t, tt, P, list[j]).len;
}

__global__ __launch_bounds__(256) void k_x_subject_copy(const uint8_t* __restrict__ buf, RecordTable t, TextTable tt, RmDupParams P,
                                                        const uint32_t* __restrict__ list, const uint64_t* __restrict__ off, uint32_t m,
                                                        uint8_t* __restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    Subject a = subject_of(buf, t, tt, P, list[j]);
    a.fold = false;  // (the host folds: it compares texts of several ranks)
    uint8_t* d = out + off[j];
    for (uint32_t q = 0; q < a.len; ++q) d[q] = a.at(q);
}

__global__ __launch_bounds__(256) void k_x_resurrect(RecordTable t, RmDupParams P, const uint32_t* __restrict__ list, uint32_t m,
                                                     uint32_t* __restrict__ out_len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t i = list[j];
    const uint32_t lh = t.l_head[i];
    out_len[i] = format_len(lh > 0 ? lh - 1 : 0, t.l_seq[i], P.fastq, P.line_width);
}

unsigned chunk_blocks(uint64_t n) {
    const uint64_t b = (n + 8191) / 8192;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
TextTable dev_tt(const TextTableH& tt) { return TextTable{tt.text_w, tt.lin_off, tt.lin, tt.lin_n}; }

}  // namespace

hipError_t launch_x_count(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint64_t* send,
                          const uint8_t* reply, const uint64_t* surv, uint64_t n, const XRanks& R, uint64_t* counts, uint64_t* bytes,
                          hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_x_count_place<false>, dim3(chunk_blocks(n)), dim3(256), 0, st, buf, t, dev_tt(tt), P, send, reply, surv, n, R,
                       (unsigned long long*)counts, (unsigned long long*)bytes, (uint64_t*)nullptr);
    return hipGetLastError();
}

hipError_t launch_x_place(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint64_t* send,
                          const uint8_t* reply, const uint64_t* surv, uint64_t n, const XRanks& R, uint64_t* cur_req, uint64_t* cur_bytes,
                          uint64_t* req, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_x_count_place<true>, dim3(chunk_blocks(n)), dim3(256), 0, st, buf, t, dev_tt(tt), P, send, reply, surv, n, R,
                       (unsigned long long*)cur_req, (unsigned long long*)cur_bytes, req);
    return hipGetLastError();
}

hipError_t launch_x_copy(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint64_t* req,
                         uint64_t m, const XRanks& R, const XFrom& seg, uint8_t* text, hipStream_t st) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_x_copy, dim3((unsigned)((4 * m + 255) / 256)), dim3(256), 0, st, buf, t, dev_tt(tt), P, req, m, R, seg, text);
    return hipGetLastError();
}

hipError_t launch_x_compare(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint64_t* req_in,
                            uint64_t m, const XFrom& from, const uint8_t* text_in, uint64_t base, uint8_t* verdict, uint64_t* status,
                            hipStream_t st) {
    if (m == 0) return hipSuccess;
    const dim3 g((unsigned)((4 * m + 255) / 256));
    if (P.ignore_case) hipLaunchKernelGGL(k_x_compare<true>, g, dim3(256), 0, st, buf, t, dev_tt(tt), P, req_in, m, from, text_in, base, verdict, status);
    else hipLaunchKernelGGL(k_x_compare<false>, g, dim3(256), 0, st, buf, t, dev_tt(tt), P, req_in, m, from, text_in, base, verdict, status);
    return hipGetLastError();
}

hipError_t launch_x_apply(const uint64_t* req, const uint8_t* verdict, uint64_t m, uint32_t* list, uint32_t cap, hipStream_t st) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_x_apply, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, req, verdict, m, list, cap);
    return hipGetLastError();
}

hipError_t launch_x_local_list(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint64_t* send,
                               const uint8_t* reply, const uint64_t* surv, uint64_t n, uint64_t base, uint32_t* list, uint32_t cap,
                               uint64_t* n_pairs, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_x_local_list, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, buf, t, dev_tt(tt), P, send, reply, surv, n, base,
                       list, cap, (unsigned long long*)n_pairs);
    return hipGetLastError();
}

hipError_t launch_x_first_list(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint32_t* first,
                               uint32_t* list, uint32_t cap, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_x_first_list, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, t, dev_tt(tt), P, first, list, cap);
    return hipGetLastError();
}

hipError_t launch_x_subject_len(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint32_t* list,
                                uint32_t m, uint32_t* len, hipStream_t st) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_x_subject_len, dim3((m + 255) / 256), dim3(256), 0, st, buf, t, dev_tt(tt), P, list, m, len);
    return hipGetLastError();
}

hipError_t launch_x_subject_copy(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const RmDupParams& P, const uint32_t* list,
                                 const uint64_t* off, uint32_t m, uint8_t* out, hipStream_t st) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_x_subject_copy, dim3((m + 255) / 256), dim3(256), 0, st, buf, t, dev_tt(tt), P, list, off, m, out);
    return hipGetLastError();
}

hipError_t launch_x_resurrect(const RecordTable& t, const RmDupParams& P, const uint32_t* list, uint32_t m, uint32_t* out_len, hipStream_t st) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_x_resurrect, dim3((m + 255) / 256), dim3(256), 0, st, t, P, list, m, out_len);
    return hipGetLastError();
}

}  // namespace bsk
