// ============================================================================
// shard_io.cpp -- a byte range of a FILE into device memory (the read side of the file -> result path).
//
// Reference: ReadFASTA[N] / ReadFASTQ[N] (/root/reference/bigseqkit/helper.go:148-178) call worker.PlainFile[N], which
// hands every executor its byte ranges of the file; the executor reads them and ReadFixer (bigseqkit-lib/helper.go) mends
// the cut records.  Here the caller cuts at record starts (bsk_find_record_start) and every worker asks for its range.
//
// What the first native driver did -- one pread() of the whole shard into ONE pinned buffer, then one copy -- spent, for
// 8 GB out of the page cache (scripts/history/r05_cli_timing.sh): 1.17 s pinning 8 GB, 0.85 s reading (one thread: 9.4
// GB/s), 0.15 s copying, 0.78 s unpinning.  Now `threads` readers take pieces of 16 MiB from a shared counter; each has
// two pinned buffers and a stream of its own, so the pread() of a piece runs while the piece before it crosses PCIe and
// while the other readers do the same -- the read of the file and the copy overlap, and only 2 x 16 MiB per reader is
// ever pinned.  The readers run on the CPUs of the GPU's NUMA node when the platform names it (BSK_NUMA=off: wherever
// the scheduler puts them), their pinned buffers are first touched there.
// ============================================================================
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bsk.h"
#include "ctx.hpp"

namespace {

constexpr size_t PIECE_DEFAULT = 16u << 20;

struct LoadJob {
    int fd;
    uint64_t offset;
    size_t n, piece;
    int device;
    char* d;
    bool pin = false;
    cpu_set_t cpus;
    std::atomic<uint64_t> next{0};
    std::atomic<int> bad{0};
    std::mutex m;
    std::string error;
    void fail(const std::string& what) {
        std::lock_guard<std::mutex> g(m);
        if (error.empty()) error = what;
        bad.store(1);
    }
};

// the CPUs of the NUMA node the GPU hangs on ("pin reader threads per GPU" -- eight workers reading through
// one socket's memory controllers is what bends the scaling curve of the file -> result path); empty when the platform does
// not say (a container without /sys/bus/pci, numa_node == -1) or BSK_NUMA=off
bool numa_cpus_of_device(int device, cpu_set_t* set) {
    const char* off = getenv("BSK_NUMA");
    if (off && !strcmp(off, "off")) return false;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess || !bdf[0]) return false;
    for (char* c = bdf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    int node = -1;
    const int got = fscanf(f, "%d", &node);
    fclose(f);
    if (got != 1 || node < 0) return false;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return false;
    char list[4096] = {0};
    const size_t k = fread(list, 1, sizeof list - 1, f);
    fclose(f);
    if (k == 0) return false;
    CPU_ZERO(set);
    int any = 0;
    for (char* p = list; *p;) {  // "0-31,64-95"
        char* e = nullptr;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { if (c >= 0) { CPU_SET((int)c, set); any = 1; } }
        p = e;
        while (*p == ',' || *p == '\n' || *p == ' ') ++p;
    }
    return any != 0;
}

void reader(LoadJob* J) {
    if (J->pin) sched_setaffinity(0, sizeof J->cpus, &J->cpus);  // (this thread only; refused outside the cgroup's set: then it stays where it is)
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    hipStream_t st = nullptr;
    if (hipSetDevice(J->device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { J->fail("libbsk: no stream on the device of the shard"); return; }
    for (int b = 0; b < 2; ++b)
        if (hipHostMalloc(&buf[b], J->piece, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) != hipSuccess) {
            J->fail("libbsk: pinned allocation failed");  // (what was allocated is released below)
        }
    const uint64_t pieces = (J->n + J->piece - 1) / J->piece;
    for (int turn = 0; !J->bad.load(); ++turn) {
        const uint64_t k = J->next.fetch_add(1);
        if (k >= pieces) break;
        const int b = turn & 1;
        if (used[b] && hipEventSynchronize(ev[b]) != hipSuccess) { J->fail("libbsk: the copy of a piece of the shard failed"); break; }
        const size_t at = (size_t)(k * J->piece), len = std::min(J->piece, J->n - at);
        size_t done = 0;
        errno = 0;
        while (done < len) {
            const ssize_t got = pread(J->fd, (char*)buf[b] + done, len - done, (off_t)(J->offset + at + done));
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) break;
            done += (size_t)got;
        }
        if (done < len) { J->fail(std::string("libbsk: short read of the shard's file") + (errno ? std::string(": ") + strerror(errno) : std::string())); break; }
        if (hipMemcpyAsync(J->d + at, buf[b], len, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[b], st) != hipSuccess) { J->fail("libbsk: the copy of a piece of the shard failed"); break; }
        used[b] = true;
    }
    if (st && hipStreamSynchronize(st) != hipSuccess) J->fail("libbsk: the copy of a piece of the shard failed");
    for (int b = 0; b < 2; ++b) {
        if (ev[b]) hipEventDestroy(ev[b]);
        if (buf[b]) hipHostFree(buf[b]);
    }
    if (st) hipStreamDestroy(st);
}

}  // namespace

extern "C" int bsk_shard_load(int fd, uint64_t offset, size_t n, int device, int threads, void** d_shard) {
    if (!d_shard || fd < 0) return bsk::global_error_set(BSK_ERR_INVALID_ARG, "libbsk: bad shard arguments");
    *d_shard = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return bsk::global_error_set(BSK_ERR_NO_DEVICE, "libbsk: no HIP device visible");
    if (device < 0 || device >= have) return bsk::global_error_set(BSK_ERR_NO_DEVICE, "libbsk: device " + std::to_string(device) + " is not visible");
    if (hipSetDevice(device) != hipSuccess) return bsk::global_error_set(BSK_ERR_NO_DEVICE, "libbsk: no such HIP device");
    void* d = nullptr;
    const bool timing = getenv("BSK_SHARD_TIMING") != nullptr;  // (stderr: allocation / readers, scripts/r06_load_sweep.py)
    const auto t_begin = std::chrono::steady_clock::now();
    // (BSK_SHARD_FAIL_ALLOC: the tests' stand-in for a shard larger than the free HBM -- the callers' fall-back and messages)
    if (getenv("BSK_SHARD_FAIL_ALLOC") || hipMalloc(&d, n ? n : 1) != hipSuccess) return bsk::global_error_set(BSK_ERR_HIP, "libbsk: device allocation of the shard (" + std::to_string(n) + " bytes) failed");
    const auto t_alloc = std::chrono::steady_clock::now();
    if (n) {
        LoadJob J;
        J.fd = fd; J.offset = offset; J.n = n; J.device = device; J.d = (char*)d;
        J.piece = PIECE_DEFAULT;
        if (const char* e = getenv("BSK_SHARD_PIECE_BYTES")) { const long long v = atoll(e); if (v >= 4096) J.piece = (size_t)v; }
        const uint64_t pieces = (n + J.piece - 1) / J.piece;
        // (8 readers: the whole 100 GB file of C2 is on the device 1.87 - 2.0 s after the allocation = 50 - 53 GB/s, what PCIe
        // Gen5 x16 gives; 4: 2.0 - 2.2, 12: 2.0, 16: 2.0 - 2.15, 24: 2.5 -- more readers only get in each other's way in the
        // page cache (scripts/r06_load_sweep.py with BSK_SHARD_TIMING=1).  What made round 6's first measurements spread
        // from 2.3 to 8.9 s per call was not the readers: every SECOND fresh process pays 3.5 - 5.8 s for its hipMalloc of
        // 100 GB -- the memory the process before it released is cleared first -- and the others 0.001 s.)
        int T = threads > 0 ? threads : 8;
        if (threads <= 0) { if (const char* e = getenv("BSK_SHARD_READERS")) { const int v = atoi(e); if (v > 0) T = v; } }  // (scripts/r06_load_sweep.py)
        T = (int)std::min<uint64_t>((uint64_t)std::min(T, 64), pieces);
        J.pin = numa_cpus_of_device(device, &J.cpus);
        // (every reader is a thread of its own: the caller's affinity is not touched)
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back(reader, &J);
        for (auto& t : pool) t.join();
        if (J.bad.load()) {
            hipFree(d);
            return bsk::global_error_set(BSK_ERR_HIP, J.error);
        }
    }
    if (timing) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "[shard] %zu bytes: allocation %.3f s, readers %.3f s\n", n, std::chrono::duration<double>(t_alloc - t_begin).count(),
                std::chrono::duration<double>(t_end - t_alloc).count());
    }
    *d_shard = d;
    return BSK_OK;
}
