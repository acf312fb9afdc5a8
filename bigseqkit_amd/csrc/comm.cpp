// ============================================================================
// comm.cpp -- the collectives of the hot path behind the C ABI (round 5).
//
// In the reference the driver gets its reductions and its shuffle from the framework, in the same binary:
//     StatsReduce      IgnisHPC Reduce      /root/reference/bigseqkit/stats.go:91
//     GrepReduceCount  IgnisHPC Reduce      bigseqkit/grep.go:175
//     rmdup            GroupByKey           bigseqkit/rmdup.go:97
//     executors        ignisDriver          bigseqkit-cli/helper.go:87-132
// Until round 4 only Python + torch.distributed could drive more than one GPU here (bigseqkit_amd/dist.py); a Go or C++
// host had no tested multi-GPU path (VERDICT r04 missing 2).  This file puts the same three collectives behind plain C
// entry points over librccl itself -- no torch, no Python:
//     bsk_comm_*                 communicators: one rank per process (ncclCommInitRank with an id the host hands round) or
//                                all ranks in one process, one thread each (ncclCommInitAll)
//     bsk_stats_collect_reduced  ONE ncclAllReduce(sum) of the dense stats vector, the collect, and -- only when the reduced
//                                vector counts lengths beyond the dense histogram -- the exchange of the overflow lists
//     bsk_count_allreduce        one u64
//     bsk_rmdup_dist_run         keys -> counts all-gather -> tuples to their owners (grouped ncclSend / ncclRecv) -> owner
//                                side -> keep bytes back -> emit, in one call
// RCCL is resolved at FIRST USE (dlopen of librccl.so.1): a process that never forms a communicator -- every single-GPU
// caller, and the Python tests, whose torch carries its own copy of the library under the same SONAME -- does not load
// 570 MB of collectives.  A second backend, "local", serves the ranks of ONE process that cannot have a GPU each (RCCL
// refuses two ranks on one device): the same calls staged through host memory between the threads -- how a one-GPU box
// runs N > 1 ranks of this code at all (tests), as gloo does for the Python harness.
// ============================================================================
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bsk.h"
#include "ctx.hpp"

namespace {

// ---- librccl, resolved at first use ----------------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl* rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            R.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (R.handle) break;
        }
        if (!R.handle) { R.error = std::string("libbsk: librccl.so.1 cannot be loaded: ") + dlerror(); return; }
        bool ok = true;
        auto sym = [&](const char* s) { void* p = dlsym(R.handle, s); if (!p) { ok = false; R.error = std::string("libbsk: librccl lacks ") + s; } return p; };
        R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
        R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
        R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
        R.AllReduce = (decltype(R.AllReduce))sym("ncclAllReduce");
        R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
        R.Send = (decltype(R.Send))sym("ncclSend");
        R.Recv = (decltype(R.Recv))sym("ncclRecv");
        R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
        R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(R.handle); R.handle = nullptr; }
    });
    return &R;
}

// ---- the "local" backend: the ranks are threads of this process ---------------------------------------------------------
struct LocalGroup {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::vector<const void*> ptr;            // what every rank published for the running collective
    std::vector<std::vector<uint64_t>> cnt;  // ... and its per-peer counts
    std::vector<int> ok;                     // ... and whether its own part of the collective worked so far (a rank
                                             // whose copy fails still reaches BOTH barriers; then every rank fails together)
    bool all_ok() const { for (int v : ok) if (!v) return false; return true; }
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = gen;
        if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

}  // namespace

struct bsk_comm {
    int world = 1, rank = 0, device = 0;
    ncclComm_t nccl = nullptr;
    std::shared_ptr<LocalGroup> local;
    uint64_t* d_word = nullptr;  // world + 2 device words of this communicator: all-gather of one value, barrier (no hipMalloc on the collective path)
    std::string err;
    int fail(int code, const std::string& m) { err = m; return code; }
};

namespace {

thread_local std::string g_comm_error;

#define COMM_HIP(c, expr)                                                                                          \
    do {                                                                                                            \
        hipError_t e__ = (expr);                                                                                    \
        if (e__ != hipSuccess) return (c)->fail(BSK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));   \
    } while (0)
#define COMM_NCCL(c, expr)                                                                                                     \
    do {                                                                                                                        \
        ncclResult_t r__ = (expr);                                                                                              \
        if (r__ != ncclSuccess) return (c)->fail(BSK_ERR_HIP, std::string(#expr) + ": " + rccl()->GetErrorString(r__));         \
    } while (0)

// in-place reduction of count u64 words on the device
int allreduce_u64(bsk_comm* c, uint64_t* d_buf, size_t count, int op, hipStream_t st) {
    if (count == 0) return BSK_OK;
    if (c->nccl) {
        COMM_HIP(c, hipSetDevice(c->device));
        const ncclRedOp_t o = op == 1 ? ncclMax : (op == 2 ? ncclMin : ncclSum);
        COMM_NCCL(c, rccl()->AllReduce(d_buf, d_buf, count, ncclUint64, o, c->nccl, st));
        return BSK_OK;
    }
    // local: a rank whose copy fails publishes "not ok" and still reaches both barriers (its peers wait there)
    LocalGroup& G = *c->local;
    std::vector<uint64_t> mine(count);
    const bool mine_ok = hipSetDevice(c->device) == hipSuccess && hipMemcpyAsync(mine.data(), d_buf, count * 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
                         hipStreamSynchronize(st) == hipSuccess;
    G.ptr[c->rank] = mine.data();
    G.ok[c->rank] = mine_ok ? 1 : 0;
    G.barrier();
    const bool all = G.all_ok();
    std::vector<uint64_t> acc(count, op == 2 ? ~0ull : 0ull);
    if (all)
        for (int r = 0; r < G.world; ++r) {
            const uint64_t* p = (const uint64_t*)G.ptr[r];
            for (size_t i = 0; i < count; ++i) acc[i] = op == 1 ? std::max(acc[i], p[i]) : (op == 2 ? std::min(acc[i], p[i]) : acc[i] + p[i]);
        }
    G.barrier();  // (every rank has read every contribution: `mine` may go)
    if (!all) return c->fail(BSK_ERR_HIP, mine_ok ? "libbsk: all-reduce: another rank of this process could not read its buffer" : "libbsk: all-reduce: reading the device buffer failed");
    COMM_HIP(c, hipMemcpyAsync(d_buf, acc.data(), count * 8, hipMemcpyHostToDevice, st));
    COMM_HIP(c, hipStreamSynchronize(st));
    return BSK_OK;
}

// one u64 of every rank -> out[world] on the host (synchronises)
int allgather_value(bsk_comm* c, uint64_t value, uint64_t* out, hipStream_t st) {
    if (c->world == 1 && !c->nccl) { out[0] = value; return BSK_OK; }
    if (c->nccl) {
        COMM_HIP(c, hipSetDevice(c->device));
        uint64_t* d = c->d_word;  // [0, world): the gathered values, [world]: this rank's
        // (a copy that cannot even be queued still lets the all-gather run: the peers are inside it)
        const bool staged = hipMemcpyAsync(d + c->world, &value, 8, hipMemcpyHostToDevice, st) == hipSuccess;
        const ncclResult_t r = rccl()->AllGather(d + c->world, d, 1, ncclUint64, c->nccl, st);
        if (r != ncclSuccess) return c->fail(BSK_ERR_HIP, std::string("ncclAllGather: ") + rccl()->GetErrorString(r));
        if (hipMemcpyAsync(out, d, (size_t)c->world * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return c->fail(BSK_ERR_HIP, "libbsk: read-back of the gathered values failed");
        if (!staged) return c->fail(BSK_ERR_HIP, "libbsk: copy of the gathered value failed");
        return BSK_OK;
    }
    LocalGroup& G = *c->local;
    G.ptr[c->rank] = &value;
    G.barrier();
    for (int r = 0; r < G.world; ++r) out[r] = *(const uint64_t*)G.ptr[r];
    G.barrier();
    return BSK_OK;
}

// No single message may exceed this many bytes.  Measured on this image in round 4 (RCCL 2.26.6, one rank sending to itself:
// scripts/history/r04_a2a_probe2.py): a message of up to 1 GiB arrives whole, of a larger one only the FIRST HALF -- silently.
// A C5 rank sends 1.9 GB of tuples (236 MB per message with 8 ranks, 946 MB with 2, 1.9 GB with one): larger exchanges go in
// rounds.  BSK_A2A_MAX_BYTES overrides (the tests force the rounds).
uint64_t a2a_max_bytes() {
    static const uint64_t v = [] {
        const char* e = getenv("BSK_A2A_MAX_BYTES");
        const long long x = e ? atoll(e) : 0;
        return x > 0 ? (uint64_t)x : (512ull << 20);
    }();
    return v;
}

// d_send holds, peer after peer, send_cnt[p] elements of elem bytes for peer p; d_recv receives recv_cnt[p] from peer p,
// peer after peer (what a rank sends to itself travels like the rest).  global_max = the largest count of any message of
// any rank (every rank must take the same number of rounds)
int alltoallv(bsk_comm* c, const uint8_t* d_send, const uint64_t* send_cnt, uint8_t* d_recv, const uint64_t* recv_cnt, size_t elem,
              uint64_t global_max, hipStream_t st) {
    if (c->nccl) {
        COMM_HIP(c, hipSetDevice(c->device));
        const uint64_t lim = std::max<uint64_t>(1, a2a_max_bytes() / elem);  // elements per message and round
        const uint64_t rounds = std::max<uint64_t>(1, (global_max + lim - 1) / lim);
        ncclResult_t bad = ncclSuccess;  // (remembered, not returned on the spot: every rank takes every round, and nothing leaves a group open)
        for (uint64_t r = 0; r < rounds; ++r) {
            const ncclResult_t gs = rccl()->GroupStart();
            if (gs != ncclSuccess) { bad = gs; continue; }
            uint64_t so = 0, ro = 0;
            for (int p = 0; p < c->world; ++p) {
                const uint64_t s0 = std::min(send_cnt[p], r * lim), s1 = std::min(send_cnt[p], (r + 1) * lim);
                const uint64_t r0 = std::min(recv_cnt[p], r * lim), r1 = std::min(recv_cnt[p], (r + 1) * lim);
                if (s1 > s0) { const ncclResult_t x = rccl()->Send(d_send + (so + s0) * elem, (s1 - s0) * elem, ncclUint8, p, c->nccl, st); if (x != ncclSuccess) bad = x; }
                if (r1 > r0) { const ncclResult_t x = rccl()->Recv(d_recv + (ro + r0) * elem, (r1 - r0) * elem, ncclUint8, p, c->nccl, st); if (x != ncclSuccess) bad = x; }
                so += send_cnt[p];
                ro += recv_cnt[p];
            }
            const ncclResult_t ge = rccl()->GroupEnd();
            if (ge != ncclSuccess) bad = ge;
        }
        if (bad != ncclSuccess) return c->fail(BSK_ERR_HIP, std::string("ncclSend / ncclRecv: ") + rccl()->GetErrorString(bad));
        return BSK_OK;
    }
    LocalGroup& G = *c->local;
    // (the peers read this rank's send buffer directly: what was queued on the stream must have landed)
    const bool mine_ok = hipSetDevice(c->device) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    G.ptr[c->rank] = d_send;
    G.cnt[c->rank].assign(send_cnt, send_cnt + G.world);
    G.ok[c->rank] = mine_ok ? 1 : 0;
    G.barrier();
    bool agree = G.all_ok();
    uint64_t ro = 0;
    for (int p = 0; p < G.world && agree; ++p) {
        uint64_t so = 0;
        for (int q = 0; q < c->rank; ++q) so += G.cnt[p][q];
        if (recv_cnt[p] != G.cnt[p][c->rank]) agree = false;
        else if (recv_cnt[p] && hipMemcpyAsync(d_recv + ro * elem, (const uint8_t*)G.ptr[p] + so * elem, recv_cnt[p] * elem, hipMemcpyDefault, st) != hipSuccess) agree = false;
        ro += recv_cnt[p];
    }
    const hipError_t e = hipStreamSynchronize(st);
    G.barrier();  // (every rank has read what it was sent: the send buffers may go)
    if (!agree || e != hipSuccess) return c->fail(BSK_ERR_HIP, "libbsk: all-to-all between the ranks of this process failed");
    return BSK_OK;
}

int barrier_(bsk_comm* c, hipStream_t st) {
    if (c->nccl) {
        COMM_HIP(c, hipSetDevice(c->device));
        uint64_t* d = c->d_word + c->world + 1;
        hipMemsetAsync(d, 0, 8, st);
        const ncclResult_t r = rccl()->AllReduce(d, d, 1, ncclUint64, ncclSum, c->nccl, st);
        const hipError_t e = hipStreamSynchronize(st);
        if (r != ncclSuccess) return c->fail(BSK_ERR_HIP, std::string("ncclAllReduce (barrier): ") + rccl()->GetErrorString(r));
        if (e != hipSuccess) return c->fail(BSK_ERR_HIP, "libbsk: barrier: the stream failed");
        return BSK_OK;
    }
    if (c->local) c->local->barrier();
    return BSK_OK;
}

// the communicator's own device words (allocated with it: nothing is allocated on the path of a collective)
bool comm_words(bsk_comm* c) {
    return hipSetDevice(c->device) == hipSuccess && hipMalloc((void**)&c->d_word, (size_t)(c->world + 2) * 8) == hipSuccess;
}

int global_fail(int code, const std::string& m) { g_comm_error = m; return code; }

}  // namespace

extern "C" {

const char* bsk_comm_error(const bsk_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

int bsk_comm_unique_id(void* id128) {
    if (!id128) return global_fail(BSK_ERR_INVALID_ARG, "libbsk: null id");
    if (!rccl()->handle) return global_fail(BSK_ERR_UNSUPPORTED, rccl()->error);
    static_assert(sizeof(ncclUniqueId) == BSK_COMM_ID_BYTES, "the id the host hands round is RCCL's");
    ncclUniqueId id;
    const ncclResult_t r = rccl()->GetUniqueId(&id);
    if (r != ncclSuccess) return global_fail(BSK_ERR_HIP, std::string("ncclGetUniqueId: ") + rccl()->GetErrorString(r));
    memcpy(id128, &id, sizeof id);
    return BSK_OK;
}

int bsk_comm_init_rank(int world, int rank, const void* id128, int device, bsk_comm** out) {
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return global_fail(BSK_ERR_INVALID_ARG, "libbsk: bad communicator arguments");
    if (!rccl()->handle) return global_fail(BSK_ERR_UNSUPPORTED, rccl()->error);
    if (hipSetDevice(device) != hipSuccess) return global_fail(BSK_ERR_NO_DEVICE, "libbsk: no such HIP device");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = rccl()->CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return global_fail(BSK_ERR_HIP, std::string("ncclCommInitRank: ") + rccl()->GetErrorString(r));
    bsk_comm* c = new bsk_comm();
    c->world = world; c->rank = rank; c->device = device; c->nccl = comm;
    if (!comm_words(c)) { rccl()->CommDestroy(comm); delete c; return global_fail(BSK_ERR_HIP, "libbsk: no device memory for the communicator"); }
    *out = c;
    return BSK_OK;
}

int bsk_comm_init_all(int ndev, const int* devices, bsk_comm** out) {
    if (!out || !devices || ndev < 1 || ndev > 64) return global_fail(BSK_ERR_INVALID_ARG, "libbsk: bad communicator arguments");
    bool distinct = true;
    for (int a = 0; a < ndev; ++a)
        for (int b = a + 1; b < ndev; ++b) distinct = distinct && devices[a] != devices[b];
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return global_fail(BSK_ERR_NO_DEVICE, "libbsk: no HIP device visible");
    for (int a = 0; a < ndev; ++a)
        if (devices[a] < 0 || devices[a] >= have) return global_fail(BSK_ERR_NO_DEVICE, "libbsk: device " + std::to_string(devices[a]) + " is not visible");
    // ONE worker has nobody to talk to: its collectives are copies, and loading librccl + ncclCommInitAll costs 1.7 s of a
    // command that runs for 0.7 s (scripts/history/r05_cli_timing.sh).  BSK_COMM=rccl keeps the one-rank RCCL communicator
    // (the tests: every collective of the N-rank path through librccl on the one GPU of the test box).
    const char* want = getenv("BSK_COMM");
    if (ndev == 1 && !(want && !strcmp(want, "rccl"))) distinct = false;
    if (distinct) {  // RCCL over xGMI: one communicator per device, all in this process
        if (!rccl()->handle) return global_fail(BSK_ERR_UNSUPPORTED, rccl()->error);
        std::vector<ncclComm_t> comms((size_t)ndev, nullptr);
        const ncclResult_t r = rccl()->CommInitAll(comms.data(), ndev, devices);
        if (r != ncclSuccess) return global_fail(BSK_ERR_HIP, std::string("ncclCommInitAll: ") + rccl()->GetErrorString(r));
        bool ok = true;
        for (int a = 0; a < ndev; ++a) {
            bsk_comm* c = new bsk_comm();
            c->world = ndev; c->rank = a; c->device = devices[a]; c->nccl = comms[(size_t)a];
            ok = comm_words(c) && ok;
            out[a] = c;
        }
        if (!ok) {
            for (int a = 0; a < ndev; ++a) { bsk_comm_destroy(out[a]); out[a] = nullptr; }
            return global_fail(BSK_ERR_HIP, "libbsk: no device memory for the communicators");
        }
        return BSK_OK;
    }
    // ranks that share a device: RCCL refuses them -- the same calls through host memory between the threads
    auto G = std::make_shared<LocalGroup>();
    G->world = ndev;
    G->ptr.assign((size_t)ndev, nullptr);
    G->cnt.assign((size_t)ndev, {});
    G->ok.assign((size_t)ndev, 1);
    for (int a = 0; a < ndev; ++a) {
        bsk_comm* c = new bsk_comm();
        c->world = ndev; c->rank = a; c->device = devices[a]; c->local = G;
        out[a] = c;
    }
    return BSK_OK;
}

int bsk_comm_destroy(bsk_comm* c) {
    if (!c) return BSK_OK;
    if (c->nccl && rccl()->handle) {
        hipSetDevice(c->device);
        rccl()->CommDestroy(c->nccl);
    }
    if (c->d_word) { hipSetDevice(c->device); hipFree(c->d_word); }
    delete c;
    return BSK_OK;
}

int bsk_comm_info(const bsk_comm* c, int* world, int* rank, int* device, int* over_rccl) {
    if (!c) return BSK_ERR_INVALID_ARG;
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (device) *device = c->device;
    if (over_rccl) *over_rccl = c->nccl ? 1 : 0;
    return BSK_OK;
}

int bsk_comm_barrier(bsk_comm* c, void* stream) {
    if (!c) return global_fail(BSK_ERR_INVALID_ARG, "libbsk: null communicator");
    return barrier_(c, (hipStream_t)stream);
}

int bsk_comm_allreduce_u64(bsk_comm* c, void* d_buf, size_t count, int op, void* stream) {
    if (!c || (!d_buf && count)) return global_fail(BSK_ERR_INVALID_ARG, "libbsk: null communicator / buffer");
    return allreduce_u64(c, (uint64_t*)d_buf, count, op, (hipStream_t)stream);
}

int bsk_comm_allgather_u64(bsk_comm* c, uint64_t value, uint64_t* out, void* stream) {
    if (!c || !out) return global_fail(BSK_ERR_INVALID_ARG, "libbsk: null communicator / out");
    return allgather_value(c, value, out, (hipStream_t)stream);
}

// GrepReduceCount (bigseqkit-lib/grep.go:598-611) across ranks: one u64
int bsk_count_allreduce(bsk_comm* c, uint64_t* inout, void* stream) {
    if (!c || !inout) return global_fail(BSK_ERR_INVALID_ARG, "libbsk: null communicator / count");
    std::vector<uint64_t> all((size_t)c->world);
    const int rc = allgather_value(c, *inout, all.data(), (hipStream_t)stream);
    if (rc != BSK_OK) return rc;
    uint64_t s = 0;
    for (uint64_t v : all) s += v;
    *inout = s;
    return BSK_OK;
}

// StatsReduce (bigseqkit-lib/stats.go:128-137 through IgnisHPC Reduce, bigseqkit/stats.go:91) + the driver's collect:
// one sum all-reduce of the stats vector (d_vec, or the context's own when NULL), bsk_stats_collect, and only when the
// reduced vector counts lengths >= hist_cap somewhere: every rank hands its overflow list to every other (one all-gather of
// the counts, one grouped exchange of the lists) and collects again.  Every rank gets the whole map.
// The vector is reduced IN PLACE: a context's own vector takes part in ONE reduction between two bsk_stats_reset calls (a
// second one would add the peers' counts again) -- refused here, on every rank alike.  One caller per context, as for every
// bsk_stats_* call (the phases below take the context's call scope one after the other; `reducing` covers the whole call).
int bsk_stats_collect_reduced(bsk_ctx* ctx, bsk_comm* c, void* d_vec, void* stream, int64_t* keys, int64_t* vals, size_t cap, size_t* n_out) {
    if (!ctx || !c) return global_fail(BSK_ERR_INVALID_ARG, "libbsk: null context / communicator");
    struct Guard {
        std::atomic<bool>& f; bool own;
        explicit Guard(std::atomic<bool>& f_) : f(f_), own(false) { bool e = false; own = f.compare_exchange_strong(e, true); }
        ~Guard() { if (own) f.store(false); }
    } guard(ctx->reducing);
    if (!guard.own) { ctx->set_error(BSK_BUSY_TEXT); return BSK_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    uint64_t* dv = d_vec ? (uint64_t*)d_vec : ctx->d_vec;
    if (!dv) { ctx->set_error("libbsk: the context holds no stats vector (bsk_stats_run first)"); return BSK_ERR_INVALID_ARG; }
    if (!d_vec && ctx->vec_reduced) {
        ctx->set_error("libbsk: the context's stats vector has been reduced already (bsk_stats_reset before the next reduction)");
        return BSK_ERR_INVALID_ARG;
    }
    int rc = allreduce_u64(c, dv, bsk_stats_vector_len(ctx), 0, st);
    if (rc != BSK_OK) { ctx->set_error(c->err); return rc; }
    if (!d_vec) ctx->vec_reduced = true;
    // slot [5] of the REDUCED vector -- the lengths beyond the dense histogram over all ranks -- decides whether the lists are
    // exchanged: read here, so that every rank decides alike, also one whose own shard was malformed (its collect fails on
    // its own error flags before it looks at the vector; it still takes part in the exchange the others enter)
    uint64_t total = 0;
    if (hipMemcpyAsync(&total, dv + 5, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        ctx->set_error("libbsk: read-back of the reduced stats vector failed");
        return BSK_ERR_HIP;  // (the copy of 8 bytes right after a collective that worked: if THIS fails the device is gone, on every rank's next call too)
    }
    rc = bsk_stats_collect(ctx, d_vec, keys, vals, cap, n_out);
    if (total == 0 || c->world == 1) return rc;
    const int rc_first = rc;
    const std::string first_error = rc_first != BSK_OK ? std::string(bsk_last_error(ctx)) : std::string();
    // From here to the end every rank enters the SAME two collectives whatever happens to it (a rank-local
    // return between them left the peers waiting): its own failure travels as the count ~0 with the all-gather, and a rank
    // that cannot stage its list still takes part in the exchange (with a list of its true length and undefined content --
    // all ranks then fail together on the sentinel the NEXT time they meet: the second all-gather below).
    size_t mine_n = 0;
    int r_own = bsk_stats_overflow_get(ctx, nullptr, 0, &mine_n);
    std::vector<uint64_t> mine(std::max<size_t>(1, mine_n));
    if (r_own == BSK_OK) r_own = bsk_stats_overflow_get(ctx, mine.data(), mine_n, &mine_n);
    std::string own_error = r_own != BSK_OK ? std::string(bsk_last_error(ctx)) : std::string();
    std::vector<uint64_t> counts((size_t)c->world);
    int r2 = allgather_value(c, r_own == BSK_OK ? (uint64_t)mine_n : ~0ull, counts.data(), st);
    if (r2 != BSK_OK) { ctx->set_error(c->err); return r2; }
    if (r_own != BSK_OK) { ctx->set_error(own_error); return r_own; }
    for (uint64_t v : counts)
        if (v == ~0ull) { ctx->set_error("libbsk: stats: another rank could not read its overflow list (its worker reports why)"); return BSK_ERR_UNSUPPORTED; }
    uint64_t all_n = 0;
    for (uint64_t v : counts) all_n += v;
    std::vector<uint64_t> send_cnt((size_t)c->world, (uint64_t)mine_n);  // the whole list to every peer ...
    send_cnt[(size_t)c->rank] = 0;                                        // ... but not to itself
    std::vector<uint64_t> recv_cnt = counts;
    recv_cnt[(size_t)c->rank] = 0;
    const uint64_t recv_n = all_n - mine_n;
    uint64_t *d_s = nullptr, *d_r = nullptr;
    bool staged = hipSetDevice(c->device) == hipSuccess && hipMalloc((void**)&d_s, std::max<size_t>(8, mine_n * 8 * (size_t)c->world)) == hipSuccess &&
                  hipMalloc((void**)&d_r, std::max<uint64_t>(8, recv_n * 8)) == hipSuccess;
    {
        // (the send buffer holds one copy of the list per peer, peer after peer: alltoallv's layout)
        uint64_t at = 0;
        for (int p = 0; p < c->world && staged; ++p) {
            if (send_cnt[(size_t)p]) staged = hipMemcpyAsync(d_s + at, mine.data(), mine_n * 8, hipMemcpyHostToDevice, st) == hipSuccess;
            at += send_cnt[(size_t)p];
        }
    }
    // whether every rank could stage: one more gathered word, so that nobody enters the exchange with a buffer it does not have
    r2 = allgather_value(c, staged ? 0 : 1, counts.data(), st);
    bool others_ok = true;
    for (uint64_t v : counts) others_ok = others_ok && v == 0;
    std::vector<uint64_t> got(std::max<uint64_t>(1, recv_n));
    if (r2 == BSK_OK && staged && others_ok) {
        r2 = alltoallv(c, (const uint8_t*)d_s, send_cnt.data(), (uint8_t*)d_r, recv_cnt.data(), 8, *std::max_element(recv_cnt.begin(), recv_cnt.end()) > mine_n
                                                                                                    ? *std::max_element(recv_cnt.begin(), recv_cnt.end()) : (uint64_t)mine_n, st);
        if (r2 != BSK_OK) ctx->set_error(c->err);
        else if (recv_n && (hipMemcpyAsync(got.data(), d_r, recv_n * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
            r2 = BSK_ERR_HIP; ctx->set_error("libbsk: read-back of the overflow lists failed");
        }
    } else if (r2 != BSK_OK) ctx->set_error(c->err);
    else { r2 = BSK_ERR_HIP; ctx->set_error(staged ? "libbsk: stats: another rank had no device memory for the overflow-list exchange" : "libbsk: no device memory for the overflow-list exchange"); }
    if (d_s) hipFree(d_s);
    if (d_r) hipFree(d_r);
    if (r2 != BSK_OK) return r2;
    if (rc_first != BSK_OK && rc_first != BSK_ERR_OVERFLOW_EXCHANGE) { ctx->set_error(first_error); return rc_first; }  // (this rank's own shard)
    if (recv_n) { r2 = bsk_stats_overflow_add(ctx, got.data(), recv_n); if (r2 != BSK_OK) return r2; }
    return bsk_stats_collect(ctx, d_vec, keys, vals, cap, n_out);
}

// RmDup over the shards of all ranks (GroupByKey, bigseqkit/rmdup.go:97) in ONE call: the device phases of bsk_rmdup_dist_*
// with the collectives between them run here -- all-gather of the record counts, tuples to their owners (rank = key % world)
// by grouped ncclSend / ncclRecv, keep byte and survivor index back the same way, and (round 6) the text of every duplicate
// whose survivor lives on another rank to that rank and one verdict byte back: RmDupCheck's comparison
// (bigseqkit-lib/rmdup.go:193-211) for EVERY pair.  Returns the survivors of THIS rank's shard in file order (the
// concatenation over the ranks equals the single-GPU output).
int bsk_rmdup_dist_run(bsk_ctx* ctx, bsk_comm* c, const void* d_shard, size_t n, int format, void* stream, bsk_out* out) {
    if (!ctx || !c || !out) return global_fail(BSK_ERR_INVALID_ARG, "libbsk: null context / communicator / out");
    hipStream_t st = (hipStream_t)stream;
    const int W = c->world;
    if (W > 64) { ctx->set_error("libbsk: rmdup across ranks takes up to 64 ranks"); return BSK_ERR_INVALID_ARG; }
    // A rank whose phase fails must not leave the others waiting in the next collective: every phase's outcome travels WITH
    // that collective (a sentinel count, an extra word of the reduced matrix, one gathered word in front of the replies), and
    // all ranks leave together.
    const std::string other = "libbsk: rmdup: another rank's shard failed (its worker reports why)";
    // `agree`: one gathered word -- did the phase work everywhere?  BSK_OK: yes; else this rank's own code (its error text is
    // set), or BSK_ERR_UNSUPPORTED with `other`
    std::vector<uint64_t> okv((size_t)W);
    auto agree = [&](int rc_own) -> int {
        const std::string own_text = rc_own != BSK_OK ? std::string(bsk_last_error(ctx)) : std::string();
        const int rc = allgather_value(c, rc_own == BSK_OK ? 0 : 1, okv.data(), st);
        if (rc != BSK_OK) { ctx->set_error(c->err); return rc; }
        if (rc_own != BSK_OK) { ctx->set_error(own_text); return rc_own; }
        for (uint64_t v : okv) if (v) { ctx->set_error(other); return BSK_ERR_UNSUPPORTED; }
        return BSK_OK;
    };
    uint64_t nrec = 0;
    const int rc_keys = bsk_rmdup_dist_keys(ctx, d_shard, n, format, stream, &nrec);
    std::vector<uint64_t> counts((size_t)W);
    int rc = allgather_value(c, rc_keys == BSK_OK ? nrec : ~0ull, counts.data(), st);
    if (rc != BSK_OK) { ctx->set_error(c->err); return rc; }
    if (rc_keys != BSK_OK) return rc_keys;
    for (uint64_t v : counts)
        if (v == ~0ull) { ctx->set_error(other); return BSK_ERR_UNSUPPORTED; }
    std::vector<uint64_t> rank_base((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r) rank_base[(size_t)r + 1] = rank_base[(size_t)r] + counts[(size_t)r];
    const uint64_t base = rank_base[(size_t)c->rank];
    if (hipSetDevice(c->device) != hipSuccess) { ctx->set_error("libbsk: hipSetDevice failed"); return BSK_ERR_HIP; }
    uint64_t *d_send = nullptr, *d_recv = nullptr;
    uint8_t *d_keep = nullptr, *d_reply = nullptr;
    uint64_t *d_surv = nullptr, *d_surv_reply = nullptr;  // the survivor's global index per tuple, and routed back per record
    uint64_t* d_xreq_in = nullptr;                         // round 6: the requests / texts this rank must compare, its verdicts, the verdicts on its own requests
    uint8_t *d_xtext_in = nullptr, *d_verdict = nullptr, *d_verdict_back = nullptr, *d_flag_s = nullptr, *d_flag_r = nullptr;
    auto cleanup = [&] {
        for (void* p : {(void*)d_send, (void*)d_recv, (void*)d_keep, (void*)d_reply, (void*)d_surv, (void*)d_surv_reply, (void*)d_xreq_in, (void*)d_xtext_in,
                        (void*)d_verdict, (void*)d_verdict_back, (void*)d_flag_s, (void*)d_flag_r})
            if (p) hipFree(p);
    };
    // how many <things> every rank sends to every other: W all-gathers of one word each would be W round trips -- the rows
    // travel as one reduction of `rows` W x W matrices in which every rank fills its own row (+ one word: "a rank failed")
    auto exchange_rows = [&](int rc_own, int rows, const std::vector<const uint64_t*>& mine, std::vector<std::vector<uint64_t>>* from_peer,
                             std::vector<uint64_t>* biggest) -> int {
        const size_t WW = (size_t)W * (size_t)W;
        std::vector<uint64_t> matrix((size_t)rows * WW + 1, 0);
        const std::string own_text = rc_own != BSK_OK ? std::string(bsk_last_error(ctx)) : std::string();
        uint64_t* d_m = nullptr;
        if (hipMalloc((void**)&d_m, matrix.size() * 8) != hipSuccess) { ctx->set_error("libbsk: no device memory"); return BSK_ERR_HIP; }
        if (rc_own == BSK_OK)
            for (int k = 0; k < rows; ++k)
                for (int p = 0; p < W; ++p) matrix[(size_t)k * WW + (size_t)c->rank * (size_t)W + (size_t)p] = mine[(size_t)k][p];
        matrix[(size_t)rows * WW] = rc_own == BSK_OK ? 0 : 1;
        bool ok = hipMemcpyAsync(d_m, matrix.data(), matrix.size() * 8, hipMemcpyHostToDevice, st) == hipSuccess;
        int rcx = BSK_OK;
        if (ok) { rcx = allreduce_u64(c, d_m, matrix.size(), 0, st); ok = rcx == BSK_OK; }
        ok = ok && hipMemcpyAsync(matrix.data(), d_m, matrix.size() * 8, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        hipFree(d_m);
        if (!ok) { ctx->set_error(c->err.empty() ? "libbsk: exchange of the split sizes failed" : c->err); return BSK_ERR_HIP; }
        if (rc_own != BSK_OK) { ctx->set_error(own_text); return rc_own; }
        if (matrix[(size_t)rows * WW] != 0) { ctx->set_error(other); return BSK_ERR_UNSUPPORTED; }
        from_peer->assign((size_t)rows, std::vector<uint64_t>((size_t)W));
        biggest->assign((size_t)rows, 0);
        for (int k = 0; k < rows; ++k) {
            for (int p = 0; p < W; ++p) (*from_peer)[(size_t)k][(size_t)p] = matrix[(size_t)k * WW + (size_t)p * (size_t)W + (size_t)c->rank];
            (*biggest)[(size_t)k] = *std::max_element(matrix.begin() + (size_t)k * WW, matrix.begin() + (size_t)(k + 1) * WW);
        }
        return BSK_OK;
    };
    std::vector<uint64_t> send_cnt((size_t)W, 0);
    int rc_own = BSK_OK;
    if (hipMalloc((void**)&d_send, std::max<uint64_t>(24, nrec * 24)) != hipSuccess) { ctx->set_error("libbsk: no device memory for the tuples"); rc_own = BSK_ERR_HIP; }
    else rc_own = bsk_rmdup_dist_pack(ctx, base, W, d_send, send_cnt.data(), stream);
    std::vector<std::vector<uint64_t>> from;
    std::vector<uint64_t> big;
    rc = exchange_rows(rc_own, 1, {send_cnt.data()}, &from, &big);
    if (rc != BSK_OK) { cleanup(); return rc; }
    const std::vector<uint64_t> recv_cnt = from[0];
    const uint64_t biggest = big[0];
    uint64_t m = 0;
    for (uint64_t v : recv_cnt) m += v;
    rc_own = BSK_OK;
    if (hipMalloc((void**)&d_recv, std::max<uint64_t>(24, m * 24)) != hipSuccess || hipMalloc((void**)&d_keep, std::max<uint64_t>(1, m)) != hipSuccess ||
        hipMalloc((void**)&d_reply, std::max<uint64_t>(1, nrec)) != hipSuccess || hipMalloc((void**)&d_surv, std::max<uint64_t>(8, m * 8)) != hipSuccess ||
        hipMalloc((void**)&d_surv_reply, std::max<uint64_t>(8, nrec * 8)) != hipSuccess) { ctx->set_error("libbsk: no device memory for the exchange"); rc_own = BSK_ERR_HIP; }
    rc = agree(rc_own);
    if (rc != BSK_OK) { cleanup(); return rc; }
    rc = alltoallv(c, (const uint8_t*)d_send, send_cnt.data(), (uint8_t*)d_recv, recv_cnt.data(), 24, biggest, st);
    if (rc != BSK_OK) { cleanup(); ctx->set_error(c->err); return rc; }
    rc = agree(bsk_rmdup_dist_resolve_ex(ctx, d_recv, m, d_keep, d_surv, stream));
    if (rc != BSK_OK) { cleanup(); return rc; }
    rc = alltoallv(c, d_keep, recv_cnt.data(), d_reply, send_cnt.data(), 1, biggest, st);  // the same routes backwards
    if (rc != BSK_OK) { cleanup(); ctx->set_error(c->err); return rc; }
    // ... and who survives
    rc = alltoallv(c, (const uint8_t*)d_surv, recv_cnt.data(), (uint8_t*)d_surv_reply, send_cnt.data(), 8, biggest, st);
    if (rc != BSK_OK) { cleanup(); ctx->set_error(c->err); return rc; }
    if (!ctx->tune.is("rmdup_xcheck", "off")) {
        // ---- round 6: the text of every duplicate to the rank of its survivor; one verdict byte back
        std::vector<uint64_t> req_cnt((size_t)W, 0), byte_cnt((size_t)W, 0);
        void *d_xreq = nullptr, *d_xtext = nullptr;
        rc_own = bsk_rmdup_dist_xpack(ctx, d_send, d_reply, d_surv_reply, base, rank_base.data(), W, req_cnt.data(), byte_cnt.data(), &d_xreq, &d_xtext, stream);
        rc = exchange_rows(rc_own, 2, {req_cnt.data(), byte_cnt.data()}, &from, &big);
        if (rc != BSK_OK) { cleanup(); return rc; }
        const std::vector<uint64_t> req_from = from[0], bytes_from = from[1];
        uint64_t m_in = 0, b_in = 0, m_out = 0;
        for (int p = 0; p < W; ++p) { m_in += req_from[(size_t)p]; b_in += bytes_from[(size_t)p]; m_out += req_cnt[(size_t)p]; }
        rc_own = BSK_OK;
        if (hipMalloc((void**)&d_xreq_in, std::max<uint64_t>(24, m_in * 24)) != hipSuccess || hipMalloc((void**)&d_xtext_in, b_in + 16) != hipSuccess ||
            hipMalloc((void**)&d_verdict, std::max<uint64_t>(1, m_in)) != hipSuccess || hipMalloc((void**)&d_verdict_back, std::max<uint64_t>(1, m_out)) != hipSuccess) {
            ctx->set_error("libbsk: no device memory for the cross-rank comparison");
            rc_own = BSK_ERR_HIP;
        }
        rc = agree(rc_own);
        if (rc != BSK_OK) { cleanup(); return rc; }
        if (big[0]) {  // (no duplicate crosses a rank anywhere: nothing travels)
            rc = alltoallv(c, (const uint8_t*)d_xreq, req_cnt.data(), (uint8_t*)d_xreq_in, req_from.data(), 24, big[0], st);
            if (rc != BSK_OK) { cleanup(); ctx->set_error(c->err); return rc; }
            rc = alltoallv(c, (const uint8_t*)d_xtext, byte_cnt.data(), d_xtext_in, bytes_from.data(), 1, big[1], st);
            if (rc != BSK_OK) { cleanup(); ctx->set_error(c->err); return rc; }
            rc = agree(bsk_rmdup_dist_xcompare(ctx, d_xreq_in, req_from.data(), d_xtext_in, bytes_from.data(), W, d_verdict, stream));
            if (rc != BSK_OK) { cleanup(); return rc; }
            rc = alltoallv(c, d_verdict, req_from.data(), d_verdict_back, req_cnt.data(), 1, big[0], st);
            if (rc != BSK_OK) { cleanup(); ctx->set_error(c->err); return rc; }
        }
        uint64_t n_flagged = 0, pairs = 0;
        rc_own = bsk_rmdup_dist_xapply(ctx, d_verdict_back, &n_flagged, &pairs, stream);
        size_t blob_n = 0;
        if (rc_own == BSK_OK && n_flagged) rc_own = bsk_rmdup_dist_flagged_get(ctx, nullptr, 0, &blob_n);
        // the sizes of the flagged lists (0 almost always: then this all-gather is the last collective of the check)
        const std::string own_text = rc_own != BSK_OK ? std::string(bsk_last_error(ctx)) : std::string();
        rc = allgather_value(c, rc_own == BSK_OK ? (uint64_t)blob_n : ~0ull, counts.data(), st);
        if (rc != BSK_OK) { cleanup(); ctx->set_error(c->err); return rc; }
        if (rc_own != BSK_OK) { cleanup(); ctx->set_error(own_text); return rc_own; }
        uint64_t all_b = 0, max_b = 0;
        for (uint64_t v : counts) {
            if (v == ~0ull) { cleanup(); ctx->set_error(other); return BSK_ERR_UNSUPPORTED; }
            all_b += v;
            max_b = std::max(max_b, v);
        }
        if (all_b) {
            // two subjects under one pair of keys somewhere: every rank gets every rank's flagged records (with their text) and
            // settles them the same way -- grouped by text, the lowest global index of every text survives
            std::vector<uint8_t> mine_blob(std::max<size_t>(1, blob_n)), all_blob((size_t)all_b);
            rc_own = blob_n ? bsk_rmdup_dist_flagged_get(ctx, mine_blob.data(), blob_n, &blob_n) : BSK_OK;
            std::vector<uint64_t> s_cnt((size_t)W, (uint64_t)blob_n), r_cnt = counts;
            if (rc_own == BSK_OK && (hipMalloc((void**)&d_flag_s, std::max<uint64_t>(8, (uint64_t)blob_n * (uint64_t)W)) != hipSuccess ||
                                     hipMalloc((void**)&d_flag_r, std::max<uint64_t>(8, all_b)) != hipSuccess)) {
                ctx->set_error("libbsk: no device memory for the flagged records");
                rc_own = BSK_ERR_HIP;
            }
            for (int p = 0; p < W && rc_own == BSK_OK && blob_n; ++p)
                if (hipMemcpyAsync(d_flag_s + (size_t)p * blob_n, mine_blob.data(), blob_n, hipMemcpyHostToDevice, st) != hipSuccess) { ctx->set_error("libbsk: staging the flagged records failed"); rc_own = BSK_ERR_HIP; }
            rc = agree(rc_own);
            if (rc != BSK_OK) { cleanup(); return rc; }
            rc = alltoallv(c, d_flag_s, s_cnt.data(), d_flag_r, r_cnt.data(), 1, max_b, st);  // (every list to every rank, itself included)
            if (rc != BSK_OK) { cleanup(); ctx->set_error(c->err); return rc; }
            rc_own = BSK_OK;
            if (hipMemcpyAsync(all_blob.data(), d_flag_r, all_b, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                ctx->set_error("libbsk: read-back of the flagged records failed");
                rc_own = BSK_ERR_HIP;
            } else rc_own = bsk_rmdup_dist_flagged_settle(ctx, all_blob.data(), all_blob.size());
            rc = agree(rc_own);
            if (rc != BSK_OK) { cleanup(); return rc; }
        }
    }
    rc = bsk_rmdup_dist_emit_ex(ctx, d_send, d_reply, d_surv_reply, base, stream, out, nullptr);
    if (rc == BSK_OK && hipStreamSynchronize(st) != hipSuccess) { rc = BSK_ERR_HIP; ctx->set_error("libbsk: the emit of the survivors failed"); }
    cleanup();
    return rc;
}

}  // extern "C"
