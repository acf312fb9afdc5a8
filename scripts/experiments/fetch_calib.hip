// Probe (not product code): what rocprofv3's FETCH_SIZE reports on gfx950 for the GATHER shapes of this engine.
// MI355X_MICROARCH.md (HBM): FETCH_SIZE is exactly half the bytes of a 16 B/lane coalesced streaming read; "other access
// widths ... are uncalibrated: calibrate on a known byte count in your own access pattern" -- VERDICT r04 item 5: the traffic
// ratios of `seq -n` (k_names' header loads), `rmdup` (the 4-lane 16-byte gathers of the byte comparison) and of the
// segmented copy were quoted with the streaming factor, i.e. as upper bounds.  Each kernel below reads a KNOWN set of bytes
// of a buffer laid out like FASTQ-150 (317-byte records); the host prints, per kernel, the bytes requested, the bytes of
// the distinct 64-byte sectors and 128-byte lines touched, and the run under `rocprofv3 --pmc FETCH_SIZE` gives the counter.
//   calib_stream16      every lane 16 B, coalesced, the whole buffer                     (the calibrated case: factor 2)
//   calib_header16      one lane per record: 16 B at the record's first byte            (k_names: the header line of a record)
//   calib_quad_gather   record i % 5 == 4: four lanes x 16 B per step over its 150 bases AND over those of a record up to
//                       1 001 places earlier                                             (k_rmdup_verify_fastq / k_rmdup_place)
//   calib_seg_read      the records with i % 5 != 4, 16 B per lane at the byte order of the OUTPUT (unaligned, contiguous
//                       inside a record)                                                 (k_seg_copy)
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o bin/fetch_calib ; run: rocprofv3 --pmc FETCH_SIZE -- bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr uint64_t REC = 317;

__device__ __forceinline__ uint32_t fold(const uint4& v) { return v.x ^ v.y ^ v.z ^ v.w; }
__device__ __forceinline__ uint64_t source_of(uint64_t i) { const uint64_t back = 1 + (i * 2654435761ull >> 7) % 1000; return i >= back ? i - back : 0; }

__global__ __launch_bounds__(256) void calib_stream16(const uint8_t* __restrict__ buf, uint64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = (blockIdx.x * 256ull + threadIdx.x) * 16; i + 16 <= n; i += (uint64_t)gridDim.x * 256ull * 16) acc ^= fold(*reinterpret_cast<const uint4*>(buf + i));
    if (acc == 0x9E3779B9u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_header16(const uint8_t* __restrict__ buf, uint64_t nrec, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nrec; i += (uint64_t)gridDim.x * 256ull) {
        uint4 v;
        __builtin_memcpy(&v, buf + i * REC, 16);
        acc ^= fold(v);
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_quad_gather(const uint8_t* __restrict__ buf, uint64_t nrec, uint32_t* sink) {
    uint32_t acc = 0;
    const uint32_t gl = threadIdx.x & 3u;
    for (uint64_t d = blockIdx.x * 64ull + (threadIdx.x >> 2); d * 5 + 4 < nrec; d += (uint64_t)gridDim.x * 64ull) {
        const uint64_t i = d * 5 + 4, f = source_of(i);
        const uint8_t *pa = buf + i * REC + 13, *pb = buf + f * REC + 13;
        for (uint32_t q = 16u * gl; q + 16u <= 150u; q += 64u) {
            uint4 x, y;
            __builtin_memcpy(&x, pa + q, 16);
            __builtin_memcpy(&y, pb + q, 16);
            acc ^= fold(x) ^ fold(y);
        }
        if (gl == 3u) { uint4 x, y; __builtin_memcpy(&x, pa + 134, 16); __builtin_memcpy(&y, pb + 134, 16); acc ^= fold(x) ^ fold(y); }
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_seg_read(const uint8_t* __restrict__ buf, uint64_t nrec, uint32_t* sink) {
    // output byte x of the kept records (4 of 5): record = (x / REC) / 4 * 5 + (x / REC) % 4, offset x % REC
    uint32_t acc = 0;
    const uint64_t kept = nrec / 5 * 4, total = kept * REC;
    for (uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 16; x + 16 <= total; x += (uint64_t)gridDim.x * 256ull * 16) {
        const uint64_t k = x / REC, o = x % REC, r = k / 4 * 5 + k % 4;
        if (o + 16 <= REC) { uint4 v; __builtin_memcpy(&v, buf + r * REC + o, 16); acc ^= fold(v); }
        else {  // a chunk across a record end: two loads, as the copy does
            const uint64_t k2 = k + 1, r2 = k2 / 4 * 5 + k2 % 4;
            uint4 a, b;
            __builtin_memcpy(&a, buf + r * REC + o, 16);          // (reads into the next record of the file: the copy's first load)
            __builtin_memcpy(&b, buf + r2 * REC + o - REC, 16);   // (the bytes in front of r2: the copy's second load)
            acc ^= fold(a) ^ fold(b);
        }
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

// Round 6: the round-5 run could not tell "a request moves the touched 64-byte sector" from "a request
// moves the whole 128-byte line" -- calib_header16 counted one 64-byte unit per distinct LINE, which fits both.  These probes
// do: ONE sector of every line is asked for (16 bytes at offset `off` of every 128-byte line, lane i takes line i), and the
// kernel is TIMED next to the streaming read of the same buffer.  If a request moved only its sector the probe would move
// half the bytes of the stream and -- both are bound by the memory system, not by request issue -- take about half its
// time; if it moves the line it takes the stream's time.  calib_two_sectors_apart asks for the OTHER sector of every line in
// a second sweep of the same launch, 8 GB later (nothing of the first sweep is in a cache any more).
__global__ __launch_bounds__(256) void calib_one_sector(const uint8_t* __restrict__ buf, uint64_t n, uint32_t off, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t l = blockIdx.x * 256ull + threadIdx.x; l * 128 + 128 <= n; l += (uint64_t)gridDim.x * 256ull) acc ^= fold(*reinterpret_cast<const uint4*>(buf + l * 128 + off));
    if (acc == 0x9E3779B9u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_two_sectors_apart(const uint8_t* __restrict__ buf, uint64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint32_t off = 0; off < 128u; off += 64u)
        for (uint64_t l = blockIdx.x * 256ull + threadIdx.x; l * 128 + 128 <= n; l += (uint64_t)gridDim.x * 256ull) acc ^= fold(*reinterpret_cast<const uint4*>(buf + l * 128 + off));
    if (acc == 0x9E3779B9u) sink[0] = acc;
}
// both sectors of a line by the same lane, back to back (two requests per line, or one?)
__global__ __launch_bounds__(256) void calib_two_sectors_together(const uint8_t* __restrict__ buf, uint64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t l = blockIdx.x * 256ull + threadIdx.x; l * 128 + 128 <= n; l += (uint64_t)gridDim.x * 256ull)
        acc ^= fold(*reinterpret_cast<const uint4*>(buf + l * 128)) ^ fold(*reinterpret_cast<const uint4*>(buf + l * 128 + 64));
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

// distinct units of `unit` bytes touched by the ranges (start, len), merged on a bitmap
struct Touch {
    std::vector<uint64_t> bits;
    uint64_t unit;
    Touch(uint64_t n, uint64_t u) : bits((n / u + 64) / 64 + 1, 0), unit(u) {}
    void add(uint64_t s, uint64_t len) { for (uint64_t u = s / unit; u <= (s + len - 1) / unit; ++u) bits[u >> 6] |= 1ull << (u & 63); }
    uint64_t bytes() const { uint64_t c = 0; for (uint64_t w : bits) c += (uint64_t)__builtin_popcountll(w); return c * unit; }
};

int main(int argc, char** argv) {
    const uint64_t nrec = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 8000000000ull) / REC;  // 8 GB: far past the 256 MiB infinity cache
    const uint64_t n = nrec * REC;
    uint8_t* d = nullptr;
    uint32_t* sink = nullptr;
    CK(hipMalloc((void**)&d, n + 4096));
    CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(d, 0x41, n + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms[8] = {0};
    auto timed = [&](int k, auto launch) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[k], e0, e1));
    };
    for (int rep = 0; rep < 2; ++rep) {  // (the second launch of every kernel is the one reported)
        timed(0, [&] { hipLaunchKernelGGL(calib_stream16, dim3(8192), dim3(256), 0, 0, d, n, sink); });
        timed(1, [&] { hipLaunchKernelGGL(calib_header16, dim3(8192), dim3(256), 0, 0, d, nrec, sink); });
        timed(2, [&] { hipLaunchKernelGGL(calib_quad_gather, dim3(8192), dim3(256), 0, 0, d, nrec, sink); });
        timed(3, [&] { hipLaunchKernelGGL(calib_seg_read, dim3(8192), dim3(256), 0, 0, d, nrec, sink); });
        timed(4, [&] { hipLaunchKernelGGL(calib_one_sector, dim3(8192), dim3(256), 0, 0, d, n, 0u, sink); });
        timed(5, [&] { hipLaunchKernelGGL(calib_two_sectors_apart, dim3(8192), dim3(256), 0, 0, d, n, sink); });
        timed(6, [&] { hipLaunchKernelGGL(calib_two_sectors_together, dim3(8192), dim3(256), 0, 0, d, n, sink); });
        CK(hipDeviceSynchronize());
    }
    {
        const char* names[7] = {"calib_stream16", "calib_header16", "calib_quad_gather", "calib_seg_read", "calib_one_sector", "calib_two_sectors_apart", "calib_two_sectors_together"};
        for (int k = 0; k < 7; ++k) printf("{\"timing\": \"%s\", \"ms\": %.4f}\n", names[k], ms[k]);
    }
    // what the kernels asked for, counted on the host
    auto report = [&](const char* name, uint64_t requested, const Touch& t64, const Touch& t128) {
        printf("{\"kernel\": \"%s\", \"requested_bytes\": %llu, \"sectors64_bytes\": %llu, \"lines128_bytes\": %llu}\n", name,
               (unsigned long long)requested, (unsigned long long)t64.bytes(), (unsigned long long)t128.bytes());
    };
    { Touch a(n, 64), b(n, 128); a.add(0, n / 16 * 16); b.add(0, n / 16 * 16); report("calib_stream16", n / 16 * 16, a, b); }
    { Touch a(n, 64), b(n, 128); for (uint64_t i = 0; i < nrec; ++i) { a.add(i * REC, 16); b.add(i * REC, 16); } report("calib_header16", nrec * 16, a, b); }
    {
        Touch a(n, 64), b(n, 128);
        uint64_t req = 0;
        for (uint64_t i = 4; i < nrec; i += 5) {
            const uint64_t f = i >= 1 + (i * 2654435761ull >> 7) % 1000 ? i - (1 + (i * 2654435761ull >> 7) % 1000) : 0;
            a.add(i * REC + 13, 150); b.add(i * REC + 13, 150); a.add(f * REC + 13, 150); b.add(f * REC + 13, 150);
            req += 2 * 160;   // nine whole 16-byte steps + the last 16 bytes once more, from either text
        }
        report("calib_quad_gather", req, a, b);
    }
    {
        const uint64_t lines = n / 128;
        printf("{\"kernel\": \"calib_one_sector\", \"requested_bytes\": %llu, \"sectors64_bytes\": %llu, \"lines128_bytes\": %llu}\n",
               (unsigned long long)(lines * 16), (unsigned long long)(lines * 64), (unsigned long long)(lines * 128));
        printf("{\"kernel\": \"calib_two_sectors_apart\", \"requested_bytes\": %llu, \"sectors64_bytes\": %llu, \"lines128_bytes\": %llu, \"line_visits_bytes\": %llu}\n",
               (unsigned long long)(lines * 32), (unsigned long long)(lines * 128), (unsigned long long)(lines * 128), (unsigned long long)(lines * 256));
        printf("{\"kernel\": \"calib_two_sectors_together\", \"requested_bytes\": %llu, \"sectors64_bytes\": %llu, \"lines128_bytes\": %llu}\n",
               (unsigned long long)(lines * 32), (unsigned long long)(lines * 128), (unsigned long long)(lines * 128));
    }
    {
        Touch a(n + 4096, 64), b(n + 4096, 128);
        const uint64_t kept = nrec / 5 * 4;
        for (uint64_t k = 0; k < kept; ++k) { const uint64_t r = k / 4 * 5 + k % 4; a.add(r * REC, REC); b.add(r * REC, REC); }
        report("calib_seg_read", kept * REC, a, b);
    }
    hipFree(d);
    hipFree(sink);
    return 0;
}
