// ============================================================================
// ops_mlfq.hip -- multi-line FASTQ -> strict 4-line FASTQ, on the device.
//
// SeqParser.Read accepts FASTQ whose sequence and quality are wrapped over several lines
// (/root/reference/bigseqkit-lib/helper.go:252-269: lines up to the first non-empty line that starts with '+' are
// sequence, everything after it is quality).  Where a record ends is decided by the FASTQ grammar (PARITY.md SPLIT-FQ):
// quality lines are taken until len(qual) >= len(seq).  The streaming kernels read strict 4-line records, so a shard
// whose head shows wrapped records is rewritten once -- same heads, the sequence lines joined, the '+' line as it is,
// the quality lines joined -- and every operator then runs on the rewritten text.  What an operator prints depends on
// (head, seq, qual) only, so the result is that of the reference on the original file.
//
// Which '@'-led line begins a record is a sequential question (a quality line may begin with '@').  It is answered in
// parallel: every '@'-led line computes where the record that would begin there ends (next[]), the records of the file
// are the orbit of line 0 under next[], and the orbit is marked chunk-wise (exit of every candidate from its chunk of
// 1024 lines -> one walk over the chunks -> one walk inside every chunk).
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ops_mlfq.hpp"

namespace bsk {
namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t CHUNK = 1024;  // lines per chunk of the orbit marking

__global__ __launch_bounds__(256) void k_nl_count(const uint8_t* __restrict__ buf, uint64_t n, uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s_c;
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    const uint64_t b0 = (uint64_t)blockIdx.x * 4096u + threadIdx.x * 16u;
    uint32_t c = 0;
    for (uint32_t k = 0; k < 16u; ++k)
        if (b0 + k < n && buf[b0 + k] == '\n') ++c;
    if (c) atomicAdd(&s_c, c);
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = s_c;
}

// ls[k] = byte after the (k - 1)-th newline (ls[0] = 0 is written by the host side)
__global__ __launch_bounds__(256) void k_nl_write(const uint8_t* __restrict__ buf, uint64_t n, const uint64_t* __restrict__ base,
                                                  uint64_t* __restrict__ ls) {
    __shared__ uint32_t s_pre[256];
    const uint64_t b0 = (uint64_t)blockIdx.x * 4096u + threadIdx.x * 16u;
    uint32_t c = 0;
    for (uint32_t k = 0; k < 16u; ++k)
        if (b0 + k < n && buf[b0 + k] == '\n') ++c;
    s_pre[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {  // inclusive scan
        const uint32_t v = threadIdx.x >= d ? s_pre[threadIdx.x - d] : 0u;
        __syncthreads();
        s_pre[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t at = base[blockIdx.x] + (s_pre[threadIdx.x] - c) + 1;  // index in ls of this thread's first newline
    for (uint32_t k = 0; k < 16u; ++k)
        if (b0 + k < n && buf[b0 + k] == '\n') ls[at++] = b0 + k + 1;
}

// next[i] for every '@'-led line: the line after the record that would begin at line i (SPLIT-FQ); plus[i] = its '+' line
// (NONE: none), slen[i] = bases (NONE: the record is not well-formed -- no '+' line, or len(qual) != len(seq))
__global__ __launch_bounds__(256) void k_mlfq_next(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ ls, uint32_t L,
                                                   uint32_t* __restrict__ next, uint32_t* __restrict__ plus, uint32_t* __restrict__ slen) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const uint64_t a = ls[i];
    const uint64_t len0 = ls[i + 1] - a - 1;
    if (len0 == 0 || buf[a] != '@') { next[i] = NONE; return; }
    uint64_t seqlen = 0, quallen = 0;
    bool isq = false, anyq = false;
    uint32_t pl = NONE, cur = i + 1, end = L;
    while (cur < L) {
        const uint64_t s = ls[cur];
        const uint64_t k = ls[cur + 1] - s - 1;
        if (!isq) {
            if (k > 0 && buf[s] == '+') { isq = true; pl = cur; }   // helper.go:255
            else seqlen += k;
        } else {
            quallen += k;
            anyq = true;
        }
        ++cur;
        if (isq && anyq && quallen >= seqlen) { end = cur; break; }
        // quality shorter than the sequence and the next line looks like a header: the record ends here and the parser
        // reports the mismatch (helper.go:308-311)
        if (isq && anyq && cur < L && ls[cur + 1] - ls[cur] - 1 > 0 && buf[ls[cur]] == '@') { end = cur; break; }
    }
    next[i] = end;
    plus[i] = pl;
    slen[i] = (isq && quallen == seqlen && seqlen < 0xFFFFFFF0ull) ? (uint32_t)seqlen : NONE;
}

// exit[i] = the first line of the orbit of candidate i that lies beyond i's chunk
__global__ __launch_bounds__(64) void k_mlfq_exit(const uint32_t* __restrict__ next, uint32_t L, uint32_t* __restrict__ exitp) {
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo64 = (uint64_t)ch * CHUNK;
    if (lo64 >= L) return;
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = L - lo > CHUNK ? lo + CHUNK : L;
    for (uint32_t i = hi; i-- > lo;) {
        const uint32_t nx = next[i];
        if (nx == NONE) { exitp[i] = NONE; continue; }
        exitp[i] = (nx >= hi) ? nx : (next[nx] == NONE ? nx : exitp[nx]);  // a non-candidate successor ends the orbit there
    }
}

// the orbit of line 0, chunk by chunk: entry[ch] = its first line inside chunk ch (NONE: it skips the chunk).
// tb = first line of the run of empty lines at the end of the text (L if none): the orbit ends there.
__global__ void k_mlfq_top(const uint32_t* __restrict__ next, const uint32_t* __restrict__ exitp, uint32_t L, uint32_t tb,
                           uint32_t* __restrict__ entry, unsigned long long* __restrict__ status) {
    uint32_t cur = 0;
    while (cur < tb) {
        entry[cur / CHUNK] = cur;
        if (next[cur] == NONE) { atomicOr(status, 1ull); return; }  // a record that does not begin with '@'
        const uint32_t e = exitp[cur];
        if (e <= cur) { atomicOr(status, 1ull); return; }
        cur = e;
    }
}

__global__ __launch_bounds__(64) void k_mlfq_mark(const uint32_t* __restrict__ next, const uint32_t* __restrict__ slen,
                                                  const uint32_t* __restrict__ entry, uint32_t L, uint32_t tb,
                                                  uint32_t* __restrict__ is_start, unsigned long long* __restrict__ status) {
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo64 = (uint64_t)ch * CHUNK;
    if (lo64 >= L) return;
    const uint32_t hi = L - (uint32_t)lo64 > CHUNK ? (uint32_t)lo64 + CHUNK : L;
    uint32_t cur = entry[ch];
    while (cur != NONE && cur < hi && cur < tb) {
        const uint32_t nx = next[cur];
        if (nx == NONE) { atomicOr(status, 1ull); return; }
        if (slen[cur] == NONE) { atomicOr(status, 2ull); return; }  // no '+' line / unmatched lengths
        is_start[cur] = 1u;
        cur = nx;
    }
}

// record r begins at line rec_line[r]; out_len[r] = bytes of its 4-line form
__global__ __launch_bounds__(256) void k_mlfq_list(const uint64_t* __restrict__ ls, const uint32_t* __restrict__ is_start,
                                                   const uint64_t* __restrict__ rank, const uint32_t* __restrict__ plus,
                                                   const uint32_t* __restrict__ slen, uint32_t L, uint32_t* __restrict__ rec_line,
                                                   uint32_t* __restrict__ out_len, unsigned long long* __restrict__ status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L || !is_start[i]) return;
    const uint64_t r = rank[i];
    rec_line[r] = i;
    const uint64_t lh = ls[i + 1] - ls[i] - 1;
    const uint32_t p = plus[i];
    const uint64_t lp = ls[p + 1] - ls[p] - 1;
    const uint64_t total = lh + 1 + (uint64_t)slen[i] + 1 + lp + 1 + (uint64_t)slen[i] + 1;
    if (total > 0xFFFFFFF0ull) { atomicOr(status, 4ull); out_len[r] = 0; return; }
    out_len[r] = (uint32_t)total;
}

// one wave per record: head, joined sequence lines, '+' line, joined quality lines
__global__ __launch_bounds__(256) void k_mlfq_emit(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ ls,
                                                   const uint32_t* __restrict__ rec_line, const uint32_t* __restrict__ next,
                                                   const uint32_t* __restrict__ plus, const uint64_t* __restrict__ out_off,
                                                   uint64_t nrec, uint8_t* __restrict__ out) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (r >= nrec) return;
    const uint32_t l0 = rec_line[r], le = next[l0], pl = plus[l0];
    uint8_t* o = out + out_off[r];
    uint64_t at = 0;
    for (uint32_t l = l0; l < le; ++l) {
        const uint64_t s = ls[l];
        const uint64_t k = ls[l + 1] - s - 1;
        for (uint64_t x = lane; x < k; x += 64u) o[at + x] = buf[s + x];
        at += k;
        // newlines: after the head, after the sequence (also when it has no line at all), after the '+' line, after the
        // quality
        uint32_t cnt = (l == l0 ? 1u : 0u) + (l + 1 == pl ? 1u : 0u) + (l == pl ? 1u : 0u) + ((l + 1 == le && l != pl) ? 1u : 0u);
        for (; cnt; --cnt) {
            if (lane == 0) o[at] = (uint8_t)'\n';
            ++at;
        }
    }
    if (pl + 1 == le) {  // no quality line at all (empty sequence at the end of the text): an empty one
        if (lane == 0) o[at] = (uint8_t)'\n';
    }
}

__global__ void k_trailing_blank(const uint64_t* __restrict__ ls, uint32_t L, uint32_t* __restrict__ tb) {
    uint32_t t = L;
    while (t > 0 && ls[t] - ls[t - 1] - 1 == 0) --t;
    *tb = t;
}

}  // namespace

uint64_t mlfq_blocks(uint64_t n) { return (n + 4095) / 4096; }

hipError_t launch_nl_count(const uint8_t* buf, uint64_t n, uint32_t* cnt, hipStream_t st) {
    hipLaunchKernelGGL(k_nl_count, dim3((unsigned)mlfq_blocks(n)), dim3(256), 0, st, buf, n, cnt);
    return hipGetLastError();
}
hipError_t launch_nl_write(const uint8_t* buf, uint64_t n, const uint64_t* base, uint64_t* ls, hipStream_t st) {
    hipLaunchKernelGGL(k_nl_write, dim3((unsigned)mlfq_blocks(n)), dim3(256), 0, st, buf, n, base, ls);
    return hipGetLastError();
}
hipError_t launch_trailing_blank(const uint64_t* ls, uint32_t L, uint32_t* tb, hipStream_t st) {
    hipLaunchKernelGGL(k_trailing_blank, dim3(1), dim3(1), 0, st, ls, L, tb);
    return hipGetLastError();
}
hipError_t launch_mlfq_resolve(const uint8_t* buf, const uint64_t* ls, uint32_t L, uint32_t tb, const MlfqScratch& S, hipStream_t st) {
    const uint32_t nch = (L + CHUNK - 1) / CHUNK;
    hipLaunchKernelGGL(k_mlfq_next, dim3((L + 255) / 256), dim3(256), 0, st, buf, ls, L, S.next, S.plus, S.slen);
    hipLaunchKernelGGL(k_mlfq_exit, dim3((nch + 63) / 64), dim3(64), 0, st, S.next, L, S.exitp);
    hipLaunchKernelGGL(k_mlfq_top, dim3(1), dim3(1), 0, st, S.next, S.exitp, L, tb, S.entry, (unsigned long long*)S.status);
    hipLaunchKernelGGL(k_mlfq_mark, dim3((nch + 63) / 64), dim3(64), 0, st, S.next, S.slen, S.entry, L, tb, S.is_start,
                       (unsigned long long*)S.status);
    return hipGetLastError();
}
uint32_t mlfq_chunks(uint32_t L) { return (L + CHUNK - 1) / CHUNK; }
hipError_t launch_mlfq_list(const uint64_t* ls, uint32_t L, const MlfqScratch& S, const uint64_t* rank, uint32_t* rec_line,
                            uint32_t* out_len, hipStream_t st) {
    hipLaunchKernelGGL(k_mlfq_list, dim3((L + 255) / 256), dim3(256), 0, st, ls, S.is_start, rank, S.plus, S.slen, L, rec_line, out_len,
                       (unsigned long long*)S.status);
    return hipGetLastError();
}
hipError_t launch_mlfq_emit(const uint8_t* buf, const uint64_t* ls, const MlfqScratch& S, const uint32_t* rec_line,
                            const uint64_t* out_off, uint64_t nrec, uint8_t* out, hipStream_t st) {
    if (nrec == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mlfq_emit, dim3((unsigned)((nrec * 64 + 255) / 256)), dim3(256), 0, st, buf, ls, rec_line, S.next, S.plus, out_off,
                       nrec, out);
    return hipGetLastError();
}

namespace {
// start[r] = first byte of record r; start[nrec] = first byte behind the last record's text (its final newline included)
__global__ __launch_bounds__(256) void k_mlfq_starts(const uint64_t* __restrict__ ls, const uint32_t* __restrict__ rec_line,
                                                     const uint32_t* __restrict__ next, uint64_t nrec, uint64_t n,
                                                     uint64_t* __restrict__ start) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const uint32_t l0 = rec_line[r];
    start[r] = ls[l0];
    if (r + 1 == nrec) {
        const uint64_t e = ls[next[l0]];  // (n + 1 for a last line without a newline)
        start[nrec] = e < n ? e : n;
    }
}
}  // namespace

hipError_t launch_mlfq_starts(const uint64_t* ls, const MlfqScratch& S, const uint32_t* rec_line, uint64_t nrec, uint64_t n,
                              uint64_t* start, hipStream_t st) {
    if (nrec == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mlfq_starts, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st, ls, rec_line, S.next, nrec, n, start);
    return hipGetLastError();
}

}  // namespace bsk
