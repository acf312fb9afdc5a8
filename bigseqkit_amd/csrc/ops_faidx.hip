// see ops_faidx.hpp
#include <hip/hip_runtime.h>

#include "ops_faidx.hpp"
#include "text_dev.hpp"

namespace bsk {
namespace {

__device__ __forceinline__ uint32_t dlen(uint64_t v) {
    uint32_t n = 1;
    while (v >= 10) { v /= 10; ++n; }
    return n;
}
__device__ __forceinline__ uint32_t dput(uint8_t* o, uint64_t v) {
    const uint32_t n = dlen(v);
    for (uint32_t k = n; k-- > 0;) { o[k] = (uint8_t)('0' + v % 10); v /= 10; }
    return n;
}

struct Row {
    const uint8_t* name;
    uint32_t name_len;
    uint64_t length, offset, linebases, linewidth, qual;  // qual: FASTQ only
};

__device__ __forceinline__ Row row_of(const uint8_t* __restrict__ buf, const RecordTable& t, const FaidxParams& P,
                                      uint64_t i, uint32_t lb) {
    Row r;
    const uint8_t* h = buf + t.start[i] + 1;
    const uint32_t lh = t.l_head[i];
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    uint32_t off = 0;
    r.name_len = P.full_head ? hl : id_span_rec(t, i, h, hl, P.id_mode, &off, P.buf_end);  // parseHeadID, faidx.go:434-450
    r.name = h + off;
    r.length = t.l_seq[i];
    r.offset = P.base_offset + t.start[i] + lh + 1;
    const bool has_lines = P.fastq || t.aux[i] > 0;  // FASTA record without a sequence line: both widths 0
    r.linebases = has_lines ? lb : 0;
    r.linewidth = has_lines ? (uint64_t)lb + 1 : 0;
    r.qual = P.fastq ? r.offset + t.l_seq[i] + 1 + t.aux[i] + 1 : 0;
    return r;
}

__device__ __forceinline__ uint32_t row_len(const Row& r, bool fastq) {
    return r.name_len + 1 + dlen(r.length) + 1 + dlen(r.offset) + 1 + dlen(r.linebases) + 1 + dlen(r.linewidth) +
           (fastq ? 1 + dlen(r.qual) : 0) + 1;
}

// the reference's check of one record's sequence lines (faidx.go:117-137): walking back from the last line, the
// width may change once, and only upwards.  Returns the first line's bases; *bad on a violation.
__device__ uint32_t scan_lines(const uint8_t* __restrict__ seq, uint32_t region, bool* bad) {
    *bad = false;
    uint32_t first = 0, nlines = 0;
    // forward: widths must be of the form W.. W w.. w with W > w  <=>  never increasing, at most one decrease
    uint32_t prev = 0, changes = 0, cur = 0;
    for (uint32_t k = 0; k <= region; ++k) {
        const bool end = k == region;
        if (end && cur == 0 && (region == 0 || seq[region - 1] == '\n')) break;  // no partial last line
        if (end || seq[k] == '\n') {
            if (nlines == 0) first = cur;
            else if (cur != prev) {
                ++changes;
                if (cur > prev || changes >= 2) *bad = true;
            }
            prev = cur;
            cur = 0;
            ++nlines;
        } else {
            ++cur;
        }
    }
    return first;
}

__global__ __launch_bounds__(256) void k_faidx_size(const uint8_t* __restrict__ buf, RecordTable t, FaidxParams P,
                                                    uint32_t* __restrict__ out_len, uint32_t* __restrict__ linebases,
                                                    uint64_t* __restrict__ status) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    uint32_t lb;
    if (P.fastq) lb = t.l_seq[i];
    else {
        const uint32_t w = t.text_w[i];
        if (w == 0) lb = t.l_seq[i];                 // one line
        else if (w != 0xFFFFFFFFu) lb = w;           // uniform lines of w >= 16 bases, the last one shorter or equal
        else {
            bool bad;
            lb = scan_lines(buf + t.start[i] + t.l_head[i] + 1, t.aux[i], &bad);
            if (bad) {
                atomicOr((unsigned long long*)&status[0], (unsigned long long)ERR_LINE_LENGTHS);
                atomicMin((unsigned long long*)&status[1], (unsigned long long)i);
            }
        }
    }
    linebases[i] = lb;
    out_len[i] = row_len(row_of(buf, t, P, i, lb), P.fastq != 0);
}

__global__ __launch_bounds__(256) void k_faidx_rows(const uint8_t* __restrict__ buf, RecordTable t, FaidxParams P,
                                                    const uint32_t* __restrict__ linebases,
                                                    const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const Row r = row_of(buf, t, P, i, linebases[i]);
    uint8_t* o = out + out_off[i];
    uint32_t x = 0;
    for (uint32_t k = 0; k < r.name_len; ++k) o[x++] = r.name[k];
    o[x++] = '\t'; x += dput(o + x, r.length);
    o[x++] = '\t'; x += dput(o + x, r.offset);
    o[x++] = '\t'; x += dput(o + x, r.linebases);
    o[x++] = '\t'; x += dput(o + x, r.linewidth);
    if (P.fastq) { o[x++] = '\t'; x += dput(o + x, r.qual); }
    o[x++] = '\n';
}

}  // namespace

hipError_t launch_faidx_size(const uint8_t* buf, const RecordTable& t, const FaidxParams& P, uint32_t* out_len,
                             uint32_t* linebases, uint64_t* status, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_faidx_size, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, t, P, out_len, linebases, status);
    return hipGetLastError();
}

hipError_t launch_faidx_rows(const uint8_t* buf, const RecordTable& t, const FaidxParams& P, const uint32_t* linebases,
                             const uint64_t* out_off, uint8_t* out, hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_faidx_rows, dim3((unsigned)((t.n + 255) / 256)), dim3(256), 0, st, buf, t, P, linebases, out_off, out);
    return hipGetLastError();
}

}  // namespace bsk
