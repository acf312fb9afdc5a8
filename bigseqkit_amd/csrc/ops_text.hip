// ops_text.hip -- classify FASTA sequence regions (contiguous / uniformly wrapped /
// irregular) and linearise the irregular ones.  See text_dev.hpp.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ops_text.hpp"
#include "text_dev.hpp"

namespace bsk {

namespace {

constexpr int GROUP = 16;

__global__ __launch_bounds__(256) void k_text_classify(const uint8_t* __restrict__ buf, RecordTable t,
                                                       uint32_t* __restrict__ text_w, uint32_t* __restrict__ lin_len) {
    const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
    const uint32_t gl = threadIdx.x % GROUP;
    const uint32_t gshift = (threadIdx.x & 63) / GROUP * GROUP;
    const bool live = g < t.n;
    const uint64_t gi = live ? g : 0;
    const uint8_t* p = buf + t.start[gi] + t.l_head[gi] + 1;
    const uint32_t L = t.l_seq[gi], region = t.aux[gi];
    const uint32_t tail_nl = (region > 0 && p[region - 1] == '\n') ? 1u : 0u;
    const uint32_t nnl = region - L;
    uint32_t w = 0;
    bool ok = true;
    if (live && nnl > tail_nl) {
        uint32_t W = 0;
        while (W < region && p[W] != '\n') ++W;
        const uint32_t lines = W ? (L + W - 1) / W : 0;
        ok = W > 0 && nnl == lines - 1 + tail_nl;
        if (ok)
            for (uint32_t k = gl; k + 1 < lines; k += GROUP)
                if (p[(uint64_t)k * (W + 1) + W] != '\n') ok = false;
        w = W;
        // very narrow lines are cheaper to linearise than to index (and keep the raw-window
        // kernels' LDS budget bounded): treat them as irregular
        if (W < 16) ok = false;
    }
    const uint64_t bad = __ballot(!ok);
    if ((bad >> gshift) & 0xFFFFull) w = TEXT_IRREGULAR;
    if (live && gl == 0) {
        text_w[g] = w;
        lin_len[g] = w == TEXT_IRREGULAR ? L : 0u;
    }
}

__global__ __launch_bounds__(256) void k_text_linearise(const uint8_t* __restrict__ buf, RecordTable t,
                                                        const uint32_t* __restrict__ text_w,
                                                        const uint64_t* __restrict__ lin_off, uint8_t* __restrict__ lin) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n || text_w[i] != TEXT_IRREGULAR) return;
    const uint8_t* p = buf + t.start[i] + t.l_head[i] + 1;
    const uint32_t region = t.aux[i];
    uint8_t* o = lin + lin_off[i];
    uint32_t x = 0;
    for (uint32_t k = 0; k < region; ++k) {
        const uint8_t c = p[k];
        if (c != '\n') o[x++] = c;
    }
}

// flatten: EVERY record that is not one line already gets a linear copy (grep -s / locate / rmdup -s read the bases of a
// record many times at arbitrary offsets: one pass that removes the line ends costs less than a division per byte read)
__global__ __launch_bounds__(256) void k_lin_len_all(RecordTable t, uint32_t* __restrict__ lin_len, uint32_t* __restrict__ text_w_flat) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint32_t w = t.text_w[i];
    lin_len[i] = w ? t.l_seq[i] : 0u;
    text_w_flat[i] = w ? TEXT_IRREGULAR : 0u;  // "read the linear copy"
}

// 16 bases of the linear copy from position x0 (the last step of a record is taken from its end): 17 consecutive source
// bytes with at most one line end among them (W >= 16), squeezed out in registers
__device__ __forceinline__ void flatten_step(const uint8_t* __restrict__ p, uint8_t* __restrict__ o, uint32_t L, uint32_t W, uint32_t x0,
                                             const uint8_t* __restrict__ buf_end) {
    const uint32_t text_len = L + (L - 1u) / W;
    const uint32_t x = x0 + 16u > L ? L - 16u : x0;
    const uint32_t sl = x / W, s0 = x + sl, k_in = W - (x - sl * W);
    uint32_t w[5];
    if (buf_end ? p + s0 + 20 <= buf_end : s0 + 20u <= text_len) {
        uint4 v0;
        uint32_t v1;
        __builtin_memcpy(&v0, p + s0, 16);
        __builtin_memcpy(&v1, p + s0 + 16, 4);
        w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1;
    } else {
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            uint32_t z = 0;
            for (int e = 0; e < 4; ++e)
                if (s0 + (uint32_t)(4 * d + e) < text_len) z |= (uint32_t)p[s0 + (uint32_t)(4 * d + e)] << (8 * e);
            w[d] = z;
        }
    }
    uint32_t c[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int tt = (int)k_in - 4 * d;
        const uint32_t keep = tt >= 4 ? 0xFFFFFFFFu : (tt <= 0 ? 0u : (1u << (8 * tt)) - 1u);
        c[d] = (w[d] & keep) | (__builtin_amdgcn_alignbyte(w[d + 1], w[d], 1) & ~keep);
    }
    const uint4 ov = make_uint4(c[0], c[1], c[2], c[3]);
    __builtin_memcpy(o + x, &ov, 16);
}

// 16 lanes per record: a uniformly wrapped record line by line, 16 bytes per step (the last step of a line from the line's
// end); an irregular one byte by byte on lane 0
__global__ __launch_bounds__(256) void k_text_flatten(const uint8_t* __restrict__ buf, RecordTable t, const uint64_t* __restrict__ lin_off,
                                                      uint8_t* __restrict__ lin, const uint8_t* __restrict__ buf_end, uint32_t long_thresh) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
    const uint32_t gl = threadIdx.x % GROUP;
    if (i >= t.n) return;
    const uint32_t W = t.text_w[i];
    if (W == 0) return;
    const uint8_t* p = buf + t.start[i] + t.l_head[i] + 1;
    const uint32_t L = t.l_seq[i];
    uint8_t* o = lin + lin_off[i];
    if (W == TEXT_IRREGULAR) {
        if (gl) return;
        const uint32_t region = t.aux[i];
        uint32_t x = 0;
        for (uint32_t k = 0; k < region; ++k) {
            const uint8_t c = p[k];
            if (c != '\n') o[x++] = c;
        }
        return;
    }
    if (W >= 16u && L >= 16u) {
        if (long_thresh && L >= long_thresh) return;  // k_text_flatten_long: whole blocks per 64 KiB of the copy
        // output-driven: a lane writes 16 consecutive bases per step (the group's stores are contiguous); they come from 17
        // consecutive source bytes with at most one newline, squeezed out in registers (ops_seq.hip, same scheme)
        for (uint32_t x0 = gl * 16u; x0 < L; x0 += GROUP * 16u) flatten_step(p, o, L, W, x0, buf_end);
        return;
    }
    const uint32_t lines = (L + W - 1) / W;
    for (uint32_t l = gl; l < lines; l += GROUP) {
        const uint32_t nb = (l + 1) * W <= L ? W : L - l * W;
        const uint8_t* src = p + (uint64_t)l * (W + 1);
        uint8_t* dst = o + (uint64_t)l * W;
        if (nb >= 16u) {
            for (uint32_t x0 = 0; x0 < nb; x0 += 16u) {
                const uint32_t x = x0 + 16u > nb ? nb - 16u : x0;
                uint4 v;
                __builtin_memcpy(&v, src + x, 16);
                __builtin_memcpy(dst + x, &v, 16);
            }
        } else {
            for (uint32_t k = 0; k < nb; ++k) dst[k] = src[k];
        }
    }
}

// chromosome-sized records (uniformly wrapped, lines of >= 16 bases): one block per FLAT_CH bytes of the copy, grid = chunks x
// long records.  (16 lanes walking a 250 MB record alone: 590 ms for grep -s / locate on 2 GB of chromosomes.)
constexpr uint32_t FLAT_CH = 64u * 1024u;
__global__ __launch_bounds__(256) void k_text_flatten_long(const uint8_t* __restrict__ buf, RecordTable t, const uint64_t* __restrict__ lin_off,
                                                           uint8_t* __restrict__ lin, const uint8_t* __restrict__ buf_end,
                                                           const uint32_t* __restrict__ long_list) {
    const uint64_t i = long_list[blockIdx.y];
    const uint32_t W = t.text_w[i];
    const uint32_t L = t.l_seq[i];
    const uint64_t lo = (uint64_t)blockIdx.x * FLAT_CH;
    if (W == 0 || W == TEXT_IRREGULAR || W < 16u || lo >= L) return;
    const uint8_t* p = buf + t.start[i] + t.l_head[i] + 1;
    uint8_t* o = lin + lin_off[i];
    const uint64_t hi = lo + FLAT_CH < L ? lo + FLAT_CH : L;
    for (uint64_t x0 = lo + threadIdx.x * 16u; x0 < hi; x0 += 256u * 16u) flatten_step(p, o, L, W, (uint32_t)x0, buf_end);
}

// bytes each record needs in the linear side buffer, from the layout the index pass recorded (RecordTable::text_w)
__global__ __launch_bounds__(256) void k_lin_len(RecordTable t, uint32_t* __restrict__ lin_len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t.n) lin_len[i] = t.text_w[i] == TEXT_IRREGULAR ? t.l_seq[i] : 0u;
}

}  // namespace

hipError_t launch_lin_len(const RecordTable& t, uint32_t* lin_len, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_lin_len, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, t, lin_len);
    return hipGetLastError();
}

hipError_t launch_text_classify(const uint8_t* buf, const RecordTable& t, uint32_t* text_w, uint32_t* lin_len,
                                hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    const uint64_t blocks = (t.n * GROUP + 255) / 256;
    hipLaunchKernelGGL(k_text_classify, dim3((unsigned)blocks), dim3(256), 0, st, buf, t, text_w, lin_len);
    return hipGetLastError();
}

hipError_t launch_text_linearise(const uint8_t* buf, const RecordTable& t, const uint32_t* text_w,
                                 const uint64_t* lin_off, uint8_t* lin, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    const uint64_t blocks = (t.n + 255) / 256;
    hipLaunchKernelGGL(k_text_linearise, dim3((unsigned)blocks), dim3(256), 0, st, buf, t, text_w, lin_off, lin);
    return hipGetLastError();
}

hipError_t launch_lin_len_all(const RecordTable& t, uint32_t* lin_len, uint32_t* text_w_flat, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_lin_len_all, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, t, lin_len, text_w_flat);
    return hipGetLastError();
}

hipError_t launch_text_flatten(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const uint64_t* lin_off, uint8_t* lin, hipStream_t st,
                               const uint32_t* long_list, uint64_t long_count, uint64_t long_max, uint32_t long_thresh) {
    if (t.n == 0) return hipSuccess;
    const uint8_t* buf_end = buf_n ? buf + buf_n : nullptr;
    hipLaunchKernelGGL(k_text_flatten, dim3((unsigned)((t.n * GROUP + 255) / 256)), dim3(256), 0, st, buf, t, lin_off, lin,
                       buf_end, long_count ? long_thresh : 0u);
    if (long_count)
        hipLaunchKernelGGL(k_text_flatten_long, dim3((unsigned)((long_max + FLAT_CH - 1) / FLAT_CH), (unsigned)long_count), dim3(256), 0, st,
                           buf, t, lin_off, lin, buf_end, long_list);
    return hipGetLastError();
}

}  // namespace bsk
