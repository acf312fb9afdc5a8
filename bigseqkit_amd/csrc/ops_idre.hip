// ============================================================================
// ops_idre.hip -- custom --id-regexp on the device: ID of a record = FindSubmatch(head)[1]
// (/root/reference/bigseqkit-lib/helper.go:362-368): the leftmost-first match of the user's expression in the header and
// the bounds of its first capture group; no match -> the whole header; a match whose group 1 took no part -> empty ID.
// One lane per record runs the Pike VM of regex_vm.hpp (thread lists in private memory): a rare path, run once per
// shard right after the record table is built; every later kernel reads the spans from the table (id_span_rec, text_dev.hpp).
// ============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>

#include "index.hpp"
#include "regex_vm.hpp"

namespace bsk {

namespace {

__global__ __launch_bounds__(64) void k_id_spans(const uint8_t* __restrict__ buf, RecordTable t, const VmProgram* __restrict__ prog,
                                                 uint32_t* __restrict__ id_off, uint32_t* __restrict__ id_len) {
    __shared__ uint32_t s_words[sizeof(VmProgram) / 4];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(prog);
        for (uint32_t k = threadIdx.x; k < sizeof(VmProgram) / 4; k += blockDim.x) s_words[k] = src[k];
    }
    __syncthreads();
    const VmProgram& s_prog = *reinterpret_cast<const VmProgram*>(s_words);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n) return;
    const uint32_t lh = t.l_head[i];
    const uint32_t hl = lh > 0 ? lh - 1 : 0;
    const uint8_t* h = buf + t.start[i] + 1;
    uint32_t caps[4];
    uint32_t off = 0, len = hl;  // not match -> the whole header (helper.go:365-367)
    if (vm_search(s_prog, h, hl, 0u, caps)) {
        if (caps[2] == 0xFFFFFFFFu || caps[3] == 0xFFFFFFFFu) { off = 0; len = 0; }  // found[1] == nil
        else { off = caps[2]; len = caps[3] - caps[2]; }
    }
    id_off[i] = off;
    id_len[i] = len;
}

}  // namespace

hipError_t launch_id_spans(const uint8_t* buf, const RecordTable& t, const VmProgram* d_prog, uint32_t* id_off, uint32_t* id_len,
                           hipStream_t st) {
    if (t.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_id_spans, dim3((unsigned)((t.n + 63) / 64)), dim3(64), 0, st, buf, t, d_prog, id_off, id_len);
    return hipGetLastError();
}

}  // namespace bsk
