// ============================================================================
// store.cpp -- FileStore / StoreFASTXN behind the C ABI, and the pipelined host path of the record operators.
//
// Reference: NewFileStore (/root/reference/bigseqkit-lib/helper.go:378-460) writes the elements of every partition,
// joined by '\n', into ONE file in partition order -- an MPI token travels from executor to executor so that partition
// k + 1 starts where k ended; StoreFASTXN (bigseqkit/helper.go:193-195) is SaveAsTextFile: one file per partition.
// Here the order comes from a scan of sizes instead of a token: a part that arrives before its predecessors is kept
// in host memory until they are written (single file), or simply owns its file (directory of parts).
//
// bsk_run_to_store is the file -> file path for a partition that sits in HOST memory (a file read into a pinned
// buffer): record-aligned chunks of 256 MiB go through two device buffers,
//     H2D(chunk i+1)  ||  kernels(chunk i)  ||  D2H(output of chunk i-1) + write,
// on three streams, so that the PCIe copies in both directions hide behind each other and behind the kernels
// (north_star: "async hipMemcpy double-buffers host file reads against kernel execution").  Operators whose result
// depends on the whole partition (rmdup, sort, rename, range, grep -C / --delete-matched) run on the whole shard and
// only their output is drained in chunks.
//
// The drain (round 3).  What one file takes was measured before anything was built (scripts/experiments/write_rates.cpp,
// profiles/r03_write_rates.txt, the GPU box's 256-thread host): ONE file accepts 6.0 GB/s in /dev/shm and 10.6 GB/s under
// /tmp whatever the number of writer threads -- a buffered write holds the inode lock for the whole copy (2 .. 16 threads on
// disjoint pieces of one file: 5.1 .. 3.6 GB/s) -- a shared mapping filled by 1 .. 16 threads 3.5 .. 3.9 GB/s (page faults
// of one address space), a preallocated file 7.6 / 12.2 GB/s; a file PER THREAD scales: 35 GB/s with 8, 108 GB/s with 32.
// So the drain of one part stays one stream of pwrite()s (three pinned buffers: two D2H copies in flight behind the piece
// being written, no lock held while writing), and the way to more than ~10 GB/s is the reference's own default layout, a
// directory of part files (StoreFASTXN) written by several contexts at once: scripts/bench_file_to_file.py --parts.
// ============================================================================
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <future>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bsk.h"
#include "anchor.hpp"
#include "ctx.hpp"
#include "ops_host.hpp"
#include "ops_host_internal.hpp"
#include "ops_segcopy.hpp"

using namespace bsk;

struct bsk_store {
    std::string path;
    bool merge = true;
    int fd = -1;               // single file
    std::mutex mu;
    uint64_t next = 0;         // the part whose bytes go to the file directly
    uint64_t offset = 0;       // end of what has been written
    struct Pending { std::vector<uint8_t> bytes; bool done = false; int fd = -1; uint64_t off = 0; };
    std::map<uint64_t, Pending> parts;  // parts that are not `next` yet (single file) / open part files (directory)
    uint64_t total = 0;
    std::string err;
};

namespace {

int store_fail(bsk_store* s, const std::string& m) {
    if (s) s->err = m;
    return BSK_ERR_INVALID_ARG;
}

bool write_all(int fd, const uint8_t* p, size_t n, uint64_t at) {
    while (n) {
        const ssize_t w = pwrite(fd, p, n, (off_t)at);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; n -= (size_t)w; at += (uint64_t)w;
    }
    return true;
}

// bytes of part `part`; last: the part is complete
int store_append(bsk_store* s, uint64_t part, const uint8_t* data, size_t n, bool last) {
    std::lock_guard<std::mutex> g(s->mu);
    if (!s->merge) {  // a file per partition (SaveAsTextFile)
        auto& P = s->parts[part];
        if (P.fd < 0) {
            char nm[64];
            snprintf(nm, sizeof nm, "/part%05llu", (unsigned long long)part);
            P.fd = open((s->path + nm).c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
            if (P.fd < 0) return store_fail(s, "libbsk: cannot create " + s->path + nm + ": " + strerror(errno));
            P.bytes.clear();
            P.off = 0;
        }
        if (!write_all(P.fd, data, n, P.off)) return store_fail(s, "libbsk: write failed: " + std::string(strerror(errno)));
        P.off += n;
        s->total += n;
        if (last) { close(P.fd); s->parts.erase(part); }
        return BSK_OK;
    }
    if (part < s->next) return store_fail(s, "libbsk: part " + std::to_string(part) + " was already completed");
    if (part != s->next) {  // not its turn: keep the bytes
        auto& P = s->parts[part];
        P.bytes.insert(P.bytes.end(), data, data + n);
        if (last) P.done = true;
        return BSK_OK;
    }
    if (!write_all(s->fd, data, n, s->offset)) return store_fail(s, "libbsk: write failed: " + std::string(strerror(errno)));
    s->offset += n;
    s->total += n;
    if (!last) return BSK_OK;
    // the next parts that are already here
    for (;;) {
        ++s->next;
        auto it = s->parts.find(s->next);
        if (it == s->parts.end()) break;
        auto& P = it->second;
        if (!P.bytes.empty()) {
            if (!write_all(s->fd, P.bytes.data(), P.bytes.size(), s->offset)) return store_fail(s, "libbsk: write failed: " + std::string(strerror(errno)));
            s->offset += P.bytes.size();
            s->total += P.bytes.size();
        }
        const bool done = P.done;
        s->parts.erase(it);
        if (!done) break;  // that part is still being produced: it now writes directly
    }
    return BSK_OK;
}

int hip_fail(bsk_ctx* c, hipError_t e, const char* what) {
    c->set_error(std::string(what) + ": " + hipGetErrorString(e));
    return BSK_ERR_HIP;
}
#define ST_TRY(c, expr)                                        \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) return hip_fail(c, e__, #expr); \
    } while (0)

typedef int (*dev_fn)(bsk_ctx*, const uint8_t*, size_t, int, hipStream_t, bsk_out*);
dev_fn fn_of(const bsk_ctx* c, bool* chunkable) {
    *chunkable = false;
    switch (c->op) {
        case Op::Seq: *chunkable = true; return seq_run_device;
        case Op::Grep: *chunkable = !c->opts.b("Count") && !(c->opts.b("DeleteMatched") && !c->opts.b("InvertMatch")); return grep_run_device;
        case Op::Locate: *chunkable = true; return locate_run_device;
        case Op::Subseq: *chunkable = true; return subseq_run_device;
        case Op::Translate: *chunkable = true; return translate_run_device;
        case Op::Fq2Fa: *chunkable = true; return fq2fa_run_device;
        case Op::RmDup: return rmdup_run_device;
        case Op::Rename: return rename_run_device;
        case Op::Sort: return sort_run_device;
        case Op::Duplicate: *chunkable = true; return records_run_device;
        default: return nullptr;
    }
}

// Pinned staging of the drain, per context (allocated by the thread that owns the context, before any writer thread runs)
constexpr size_t DRAIN_PIECE = (size_t)32 << 20;
constexpr int DRAIN_NBUF = 3;
struct Drainer {
    uint8_t* pin[DRAIN_NBUF] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[DRAIN_NBUF] = {nullptr, nullptr, nullptr};
    uint8_t* dev[DRAIN_NBUF] = {nullptr, nullptr, nullptr};  // round 6: a result that is a list of slices is gathered piece by piece into these
};

int drainer_prepare(bsk_ctx* c) {
    if (c->drainer) return BSK_OK;
    Drainer* d = new Drainer;
    for (int b = 0; b < DRAIN_NBUF; ++b) {
        if (hipHostMalloc((void**)&d->pin[b], DRAIN_PIECE, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&d->ev[b], hipEventDisableTiming) != hipSuccess) {
            for (int k = 0; k <= b; ++k) { if (d->pin[k]) hipHostFree(d->pin[k]); if (d->ev[k]) hipEventDestroy(d->ev[k]); }
            delete d;
            c->set_error("libbsk: pinned staging buffers of the output drain could not be allocated");
            return BSK_ERR_HIP;
        }
    }
    c->drainer = d;
    return BSK_OK;
}

// where the next n bytes of `part` go.  direct: fd + file offset reserved (the part's turn, or its own file);
// else the bytes are kept in host memory until the part's turn (store_append)
struct Target { bool direct = false; int fd = -1; uint64_t base = 0; };
int reserve_target(bsk_store* s, uint64_t part, size_t n, Target* t) {
    std::lock_guard<std::mutex> g(s->mu);
    if (!s->merge) {
        auto& P = s->parts[part];
        if (P.fd < 0) {
            char nm[64];
            snprintf(nm, sizeof nm, "/part%05llu", (unsigned long long)part);
            P.fd = open((s->path + nm).c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
            if (P.fd < 0) return store_fail(s, "libbsk: cannot create " + s->path + nm + ": " + strerror(errno));
            P.off = 0;
        }
        t->direct = true; t->fd = P.fd; t->base = P.off;
        P.off += n;
        s->total += n;
        return BSK_OK;
    }
    if (part < s->next) return store_fail(s, "libbsk: part " + std::to_string(part) + " was already completed");
    if (part != s->next) { t->direct = false; return BSK_OK; }
    t->direct = true; t->fd = s->fd; t->base = s->offset;
    s->offset += n;
    s->total += n;
    return BSK_OK;
}

// the part is complete: the next parts that are already here follow (single file) / the part file is closed (directory)
int finish_part(bsk_store* s, uint64_t part) {
    std::lock_guard<std::mutex> g(s->mu);
    if (!s->merge) {
        auto it = s->parts.find(part);
        if (it != s->parts.end()) { if (it->second.fd >= 0) close(it->second.fd); s->parts.erase(it); }
        return BSK_OK;
    }
    if (part != s->next) return BSK_OK;  // (kept in memory: store_append marked it done)
    for (;;) {
        ++s->next;
        auto it = s->parts.find(s->next);
        if (it == s->parts.end()) break;
        auto& P = it->second;
        if (!P.bytes.empty()) {
            if (!write_all(s->fd, P.bytes.data(), P.bytes.size(), s->offset)) return store_fail(s, "libbsk: write failed: " + std::string(strerror(errno)));
            s->offset += P.bytes.size();
            s->total += P.bytes.size();
        }
        const bool done = P.done;
        s->parts.erase(it);
        if (!done) break;  // that part is still being produced: it now writes directly
    }
    return BSK_OK;
}

// device bytes -> store.  Returns a status and, on failure, the message in *err (the caller owns the context's error
// text: this function also runs on the writer thread of bsk_run_to_store, next to the thread that computes).
int drain_to_store(bsk_ctx* c, bsk_store* s, uint64_t part, const uint8_t* d, size_t n, bool last, hipStream_t d2h, std::string* err,
                   const bsk_ctx::PendingOut* seg = nullptr /* the n bytes are this list of slices (d is null) */) {
    auto hipfail = [&](hipError_t e, const char* what) { *err = std::string(what) + ": " + hipGetErrorString(e); return BSK_ERR_HIP; };
#define DR_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return hipfail(e__, #expr); } while (0)
    DR_TRY(hipSetDevice(c->device));
    Drainer* D = (Drainer*)c->drainer;
    if (!D) { *err = "libbsk: drain without staging buffers"; return BSK_ERR_INVALID_ARG; }
    int rc = BSK_OK;
    if (n) {
        Target t;
        rc = reserve_target(s, part, n, &t);
        if (rc != BSK_OK) { *err = s->err; return rc; }
        const size_t npieces = (n + DRAIN_PIECE - 1) / DRAIN_PIECE;
        auto piece_len = [&](size_t k) { return std::min(DRAIN_PIECE, n - k * DRAIN_PIECE); };
        if (seg)
            for (int b = 0; b < DRAIN_NBUF; ++b)
                if (!D->dev[b]) DR_TRY(hipMalloc((void**)&D->dev[b], DRAIN_PIECE));
        auto issue = [&](size_t k) -> hipError_t {
            const int b = (int)(k % DRAIN_NBUF);
            const uint8_t* from = d ? d + k * DRAIN_PIECE : D->dev[b];
            hipError_t e = hipSuccess;
            if (seg)  // the piece of the text out of its slices (the buffer is free: the D2H copy that read it was waited for before its piece was written)
                e = launch_seg_copy_range(seg->seg_src, seg->seg_off, seg->nseg, seg->first4k, D->dev[b], k * DRAIN_PIECE,
                                          k * DRAIN_PIECE + piece_len(k), seg->total, seg->lo, seg->hi, d2h);
            if (e == hipSuccess) e = hipMemcpyAsync(D->pin[b], from, piece_len(k), hipMemcpyDeviceToHost, d2h);
            return e == hipSuccess ? hipEventRecord(D->ev[b], d2h) : e;
        };
        // two copies in flight ahead of the piece being written (a buffer is free again once its piece has been written)
        for (size_t k = 0; k < std::min<size_t>(npieces, DRAIN_NBUF - 1); ++k) {
            const hipError_t e = issue(k);
            if (e != hipSuccess) return hipfail(e, "hipMemcpyAsync (drain)");
        }
        for (size_t k = 0; k < npieces && rc == BSK_OK; ++k) {
            const int b = (int)(k % DRAIN_NBUF);
            if (k + DRAIN_NBUF - 1 < npieces) {
                const hipError_t e = issue(k + DRAIN_NBUF - 1);
                if (e != hipSuccess) { rc = hipfail(e, "hipMemcpyAsync (drain)"); break; }
            }
            const hipError_t e = hipEventSynchronize(D->ev[b]);
            if (e != hipSuccess) { rc = hipfail(e, "hipEventSynchronize (drain)"); break; }
            const size_t len = piece_len(k);
            if (t.direct) {  // (the range is reserved: no lock is held while the bytes go out)
                if (!write_all(t.fd, D->pin[b], len, t.base + k * DRAIN_PIECE)) { *err = "libbsk: write failed: " + std::string(strerror(errno)); rc = BSK_ERR_INVALID_ARG; }
            } else {
                rc = store_append(s, part, D->pin[b], len, false);
                if (rc != BSK_OK) *err = s->err;
            }
        }
        if (rc != BSK_OK) { hipStreamSynchronize(d2h); return rc; }
    }
    if (last) {
        // a part that waits in memory is marked complete by an empty append; one whose turn it is hands over to its successors
        bool in_turn;
        { std::lock_guard<std::mutex> g(s->mu); in_turn = !s->merge || part == s->next; }
        rc = in_turn ? finish_part(s, part) : store_append(s, part, nullptr, 0, true);
        if (rc != BSK_OK) *err = s->err;
    }
    return rc;
#undef DR_TRY
}

}  // namespace

namespace bsk {
void store_drainer_free(bsk_ctx* c) {
    Drainer* d = (Drainer*)c->drainer;
    if (!d) return;
    for (int b = 0; b < DRAIN_NBUF; ++b) { if (d->pin[b]) hipHostFree(d->pin[b]); if (d->ev[b]) hipEventDestroy(d->ev[b]); if (d->dev[b]) hipFree(d->dev[b]); }
    delete d;
    c->drainer = nullptr;
}
}  // namespace bsk

extern "C" {

int bsk_store_open(const char* path, int merge, bsk_store** out) {
    if (!path || !out) return BSK_ERR_INVALID_ARG;
    bsk_store* s = new bsk_store;
    s->path = path;
    s->merge = merge != 0;
    if (s->merge) {
        s->fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (s->fd < 0) { delete s; return BSK_ERR_INVALID_ARG; }
    } else {
        struct stat sb;
        if (!(stat(path, &sb) == 0 && S_ISDIR(sb.st_mode)) && mkdir(path, 0755) != 0) { delete s; return BSK_ERR_INVALID_ARG; }
    }
    *out = s;
    return BSK_OK;
}

const char* bsk_store_error(const bsk_store* s) { return s ? s->err.c_str() : ""; }

int bsk_store_put_host(bsk_store* s, uint64_t part, const void* data, size_t n) {
    if (!s || (n && !data)) return BSK_ERR_INVALID_ARG;
    return store_append(s, part, (const uint8_t*)data, n, true);
}

int bsk_store_put(bsk_store* s, bsk_ctx* c, uint64_t part, const bsk_out* o) {
    if (!s || !c || !o) return BSK_ERR_INVALID_ARG;
    if (c->device < 0) { c->set_error("libbsk: context was created without a device"); return BSK_ERR_NO_DEVICE; }
    bsk_call_scope scope(c);
    if (!scope.owns) { c->set_error(BSK_BUSY_TEXT); return BSK_ERR_INVALID_ARG; }
    ST_TRY(c, hipSetDevice(c->device));
    ST_TRY(c, hipDeviceSynchronize());
    if (!c->copy_stream[1]) ST_TRY(c, hipStreamCreateWithFlags(&c->copy_stream[1], hipStreamNonBlocking));
    int rc = drainer_prepare(c);
    if (rc != BSK_OK) return rc;
    std::string err;
    if (o->n_segments && o->len) {
        // round 6: the result is a list of slices (include/bsk.h): every 32 MiB piece of the text is gathered out of them into
        // a staging buffer of the device and leaves from there -- the one block is never made
        const bsk_ctx::PendingOut& P = c->pend_out;
        if (P.kind == 0 || o->d_seg_src != P.seg_src || o->d_seg_off != P.seg_off || o->n_segments != P.nseg || o->len != P.total) {
            c->set_error("libbsk: this result is not the one the context holds as slices (a later run replaced it)");
            return BSK_ERR_INVALID_ARG;
        }
        rc = pending_first4k(c, nullptr);
        if (rc != BSK_OK) return rc;
        ST_TRY(c, hipDeviceSynchronize());
        rc = drain_to_store(c, s, part, nullptr, o->len, true, c->copy_stream[1], &err, &P);
    } else {
        rc = drain_to_store(c, s, part, (const uint8_t*)o->d_data, o->len, true, c->copy_stream[1], &err);
    }
    if (rc != BSK_OK) c->set_error(err);
    return rc;
}

int bsk_store_close(bsk_store* s, uint64_t* total_bytes) {
    if (!s) return BSK_ERR_INVALID_ARG;
    int rc = BSK_OK;
    {
        std::lock_guard<std::mutex> g(s->mu);
        if (s->merge) {
            // parts that never became `next` (a gap in the part numbers): written in part order all the same
            for (auto& kv : s->parts) {
                if (!kv.second.bytes.empty()) {
                    if (!write_all(s->fd, kv.second.bytes.data(), kv.second.bytes.size(), s->offset)) rc = BSK_ERR_INVALID_ARG;
                    s->offset += kv.second.bytes.size();
                    s->total += kv.second.bytes.size();
                }
            }
            if (s->fd >= 0) close(s->fd);
        } else {
            for (auto& kv : s->parts) if (kv.second.fd >= 0) close(kv.second.fd);
        }
        if (total_bytes) *total_bytes = s->total;
    }
    delete s;
    return rc;
}

int bsk_run_to_store(bsk_ctx* c, const void* host_shard, size_t n, int format, int64_t pid, bsk_store* s, uint64_t part,
                     uint64_t* out_bytes, uint64_t* out_records) {
    if (!c || !s) return BSK_ERR_INVALID_ARG;
    if (c->device < 0) { c->set_error("libbsk: context was created without a device"); return BSK_ERR_NO_DEVICE; }
    if (format != BSK_FORMAT_FASTA && format != BSK_FORMAT_FASTQ) { c->set_error("libbsk: bad format"); return BSK_ERR_INVALID_ARG; }
    if (n && !host_shard) { c->set_error("libbsk: null shard"); return BSK_ERR_INVALID_ARG; }
    bool chunkable = false;
    dev_fn fn = fn_of(c, &chunkable);
    if (!fn) { c->set_error("libbsk: bsk_run_to_store: operator without a single-shard record output"); return BSK_ERR_INVALID_ARG; }
    if (out_bytes) *out_bytes = 0;      // (also on the early ways out: ADVICE r03)
    if (out_records) *out_records = 0;
    bsk_call_scope scope(c);
    if (!scope.owns) { c->set_error(BSK_BUSY_TEXT); return BSK_ERR_INVALID_ARG; }
    // (the chunks' outputs are drained from the two output buffers while the next chunk computes: one block each, whatever
    // the switch "out" says)
    struct Contig { bsk_ctx* c; bool was; ~Contig() { c->force_contiguous = was; } } contig{c, c->force_contiguous};
    c->force_contiguous = true;
    c->pend_out.kind = 0;
    ST_TRY(c, hipSetDevice(c->device));
    for (int b = 0; b < 2; ++b)
        if (!c->copy_stream[b]) ST_TRY(c, hipStreamCreateWithFlags(&c->copy_stream[b], hipStreamNonBlocking));
    if (!c->own_stream) ST_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    hipStream_t st = c->own_stream, h2d = c->copy_stream[0], d2h = c->copy_stream[1];
    int rc = drainer_prepare(c);  // (pinned buffers and events exist before the writer thread does)
    if (rc != BSK_OK) return rc;
    // everything the call borrows from the context or creates goes back on EVERY way out: the spare output buffer (the
    // two are swapped per chunk -- left swapped, the next call would drain one chunk while computing the next in the
    // same memory), the events, the alphabet pinned for the chunks, the writer thread
    struct Scope {
        bsk_ctx* c;
        uint8_t* alt_out;
        uint64_t alt_cap;
        bool swapped = false;
        hipEvent_t in_done[2] = {nullptr, nullptr}, in_free[2] = {nullptr, nullptr};
        std::future<std::pair<int, std::string>> writer;
        Alphabet saved_alphabet;
        bool alphabet_pinned = false;
        ~Scope() {
            if (writer.valid()) writer.get();
            if (swapped) { std::swap(c->d_out, alt_out); std::swap(c->out_cap, alt_cap); }
            c->d_out_alt = alt_out;
            c->out_alt_cap = alt_cap;
            if (alphabet_pinned) { c->alphabet = saved_alphabet; c->alphabet_guessed = false; }
            for (int b = 0; b < 2; ++b) { if (in_done[b]) hipEventDestroy(in_done[b]); if (in_free[b]) hipEventDestroy(in_free[b]); }
        }
    } S{c, c->d_out_alt, c->out_alt_cap};
    S.saved_alphabet = c->alphabet;
    for (int b = 0; b < 2; ++b) {
        ST_TRY(c, hipEventCreateWithFlags(&S.in_done[b], hipEventDisableTiming));
        ST_TRY(c, hipEventCreateWithFlags(&S.in_free[b], hipEventDisableTiming));
    }
    const char* env = c->tune.get("stage_bytes");
    const size_t want = env && strtoull(env, nullptr, 10) ? (size_t)strtoull(env, nullptr, 10) : ((size_t)256 << 20);
    // records wrapped over several lines: the cut points below assume 4-line records, so the shard goes as one piece
    if (format == BSK_FORMAT_FASTQ && fastq_head_multiline((const uint8_t*)host_shard, std::min<size_t>(n, 256 * 1024))) chunkable = false;
    const size_t chunk = chunkable ? want : n;
    const uint8_t* h = (const uint8_t*)host_shard;
    // cut points: record starts (a chunk holds whole records)
    std::vector<size_t> cuts{0};
    while (cuts.back() < n) {
        size_t lo = cuts.back(), hi = n;
        if (n - lo > chunk) {
            hi = format == BSK_FORMAT_FASTQ ? (size_t)find_fastq_cut(h, n, lo + chunk) : (size_t)find_fasta_start(h, n, lo + chunk);
            if (hi <= lo || hi > n) hi = n;
        }
        cuts.push_back(hi);
    }
    const size_t nchunks = cuts.size() - 1;
    uint64_t bytes = 0, records = 0;
    auto stage_in = [&](size_t i) -> int {  // H2D of chunk i into input buffer i & 1
        const int b = (int)(i & 1);
        const size_t len = cuts[i + 1] - cuts[i];
        if (len > c->stage_cap_b[b] || !c->d_stage[b]) {
            ST_TRY(c, hipStreamSynchronize(st));
            if (c->d_stage[b]) ST_TRY(c, hipFree(c->d_stage[b]));
            c->d_stage[b] = nullptr;
            ST_TRY(c, hipMalloc((void**)&c->d_stage[b], len + len / 16 + 4096));
            c->stage_cap_b[b] = len + len / 16;
        }
        if (i >= 2) ST_TRY(c, hipStreamWaitEvent(h2d, S.in_free[b], 0));
        ST_TRY(c, hipMemcpyAsync(c->d_stage[b], h + cuts[i], len, hipMemcpyHostToDevice, h2d));
        ST_TRY(c, hipEventRecord(S.in_done[b], h2d));
        return BSK_OK;
    };
    if (nchunks == 0) return store_append(s, part, nullptr, 0, true);
    rc = stage_in(0);
    // the output of chunk i is drained (D2H + write, on a writer thread) while chunk i+1 computes: two output buffers,
    // swapped with the context's; the drain of chunk i must be over before chunk i+2 reuses its buffer, and before the
    // drain of chunk i+1 starts (the bytes of a part are appended in order)
    for (size_t i = 0; i < nchunks && rc == BSK_OK; ++i) {
        const int b = (int)(i & 1);
        if (i + 1 < nchunks) { rc = stage_in(i + 1); if (rc != BSK_OK) break; }
        ST_TRY(c, hipStreamWaitEvent(st, S.in_done[b], 0));
        ST_TRY(c, hipMemsetAsync(c->d_status, 0, 8 * sizeof(uint64_t), st));
        c->cur_pid = (pid == 0 && i == 0) ? 0 : (pid == 0 ? 1 : pid);  // locate: the header row belongs to the first chunk of partition 0
        // SeqType auto: the reference guesses the alphabet ONCE per partition, from its first record (helper.go:286-291);
        // a chunk that guessed from its own first record could search other strands or validate other letters than a
        // run over the whole shard.  The guess of chunk 0 is pinned for the rest of the call.
        // (switch "pin_alphabet": the caller feeds ONE partition through several calls -- the command line's pieces of a shard
        // that does not fit the GPU -- and the guess of the first call's first record stays for all of them)
        const bool pin_for_good = c->tune.is("pin_alphabet", "1");
        if (i == 0 && (nchunks > 1 || pin_for_good) && c->alphabet == AB_NONE && c->op != Op::Duplicate) {
            ST_TRY(c, hipStreamSynchronize(st));  // (chunk 0 is on the device: its head is read back)
            int arc = BSK_OK;
            const Alphabet ab = partition_alphabet(c, c->d_stage[b], cuts[1] - cuts[0], format, st, &arc);
            if (arc != BSK_OK) { rc = arc; break; }
            c->alphabet = ab;
            c->alphabet_guessed = true;
            S.alphabet_pinned = !pin_for_good;
        }
        std::swap(c->d_out, S.alt_out);
        std::swap(c->out_cap, S.alt_cap);
        S.swapped = !S.swapped;
        bsk_out o;
        memset(&o, 0, sizeof o);
        ++c->call_gen;  // (the staging buffers come round again with other text: nothing sampled from chunk i - 2 may survive)
        rc = fn(c, c->d_stage[b], cuts[i + 1] - cuts[i], format, st, &o);
        if (rc == BSK_ERR_MULTILINE_FASTQ) {  // (range / head / duplicate deal with wrapped records themselves: records_run_device)
            const uint8_t* d2 = nullptr;
            size_t n2 = 0;
            rc = normalize_multiline_fastq(c, c->d_stage[b], cuts[i + 1] - cuts[i], st, &d2, &n2);
            if (rc != BSK_OK) break;
            ST_TRY(c, hipMemsetAsync(c->d_status, 0, 8 * sizeof(uint64_t), st));
            c->norm_active = true;
            rc = fn(c, d2, n2, format, st, &o);
            c->norm_active = false;
        }
        if (rc != BSK_OK) break;
        ST_TRY(c, hipStreamSynchronize(st));  // the output is complete (the run functions end with launches in flight)
        ST_TRY(c, hipEventRecord(S.in_free[b], st));
        if (S.writer.valid()) {
            auto w = S.writer.get();
            if (w.first != BSK_OK) { c->set_error(w.second); rc = w.first; break; }
        }
        const uint8_t* od = (const uint8_t*)o.d_data;
        const size_t ol = o.len;
        const bool last = i + 1 == nchunks;
        S.writer = std::async(std::launch::async, [=]() {
            std::string err;
            const int wrc = drain_to_store(c, s, part, od, ol, last, d2h, &err);
            return std::make_pair(wrc, err);
        });
        bytes += o.len;
        records += o.records;
    }
    if (S.writer.valid()) {
        auto w = S.writer.get();
        if (rc == BSK_OK && w.first != BSK_OK) { c->set_error(w.second); rc = w.first; }
    }
    if (out_bytes) *out_bytes = bytes;
    if (out_records) *out_records = records;
    return rc;
}

}  // extern "C"
