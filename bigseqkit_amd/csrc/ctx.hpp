// Operator context == one instance of a reference plugin operator between
// Before() and After() (see include/bsk.h).
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cstdlib>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "index.hpp"
#include "opts.hpp"
#include "regex_nfa.hpp"
#include "regex_vm.hpp"

// Run-time switches of a context (which kernel variant runs, thresholds used by the tests).  Round 2 read them with
// getenv() on the call path; now a context takes the environment's values ONCE, when it is created (BSK_<NAME>), and
// bsk_ctx_set(ctx, "<name>", "<value>") changes them for that context only -- a cgo caller sets them per operator instead
// of per process.  Names (lower case, without the BSK_ prefix) and meanings: INTEGRATION.md "Switches".
struct bsk_tuning {
    static const char* const* names() {
        static const char* const N[] = {"filter", "grep_shiftand", "index", "locate_nopre", "long_bytes", "min_range_bytes", "names",
                                        "names_scale", "out", "pin_alphabet", "ranges_per_wave", "rmdup", "rmdup_buckets", "rmdup_hash", "rmdup_k1_bits", "rmdup_k2_bits", "rmdup_keys", "rmdup_place", "rmdup_xcheck", "rmdup_xlocal", "scan", "segcopy",
                                        "sort", "stage_bytes", "stats_a", "stats_fasta", "subseq", "subseq_scale", "text", "translate", "translate_index", "translate_probe", "translate_stream", "tr_lanes", nullptr};
        return N;
    }
    std::map<std::string, std::string> v;
    static bool known(const std::string& k) {
        for (const char* const* n = names(); *n; ++n) if (k == *n) return true;
        return false;
    }
    void load_env() {
        for (const char* const* n = names(); *n; ++n) {
            std::string e = "BSK_";
            for (const char* q = *n; *q; ++q) e.push_back((char)toupper((unsigned char)*q));
            if (const char* val = getenv(e.c_str())) v[*n] = val;
        }
    }
    // the value, or null when unset (the shape of the getenv() calls it replaces)
    const char* get(const char* k) const {
        auto it = v.find(k);
        return it == v.end() ? nullptr : it->second.c_str();
    }
    bool is(const char* k, const char* val) const { const char* e = get(k); return e && strcmp(e, val) == 0; }
    long long num(const char* k, long long dflt = 0) const { const char* e = get(k); return e && atoll(e) > 0 ? atoll(e) : dflt; }
};

struct bsk_ctx {
    bsk_tuning tune;
    bsk::Op op;
    bsk::Options opts;
    std::string opts_json;
    int device = -1;  // < 0: options only
    mutable std::mutex mu;
    mutable std::string last_error;
    // One caller at a time (include/bsk.h "Threads"): a context owns its output buffer, record table and control block, so
    // the reference's "Call() from Threads() goroutines on one struct" (bigseqkit-lib/helper.go:413-416) maps to one
    // context per caller thread.  A second call that arrives while one is running is refused (BSK_ERR_INVALID_ARG,
    // "context busy"), not raced.
    std::atomic<bool> busy{false};
    std::atomic<bool> reducing{false};  // a bsk_stats_collect_reduced is running on this context (its phases take `busy` one after the other)
    bool vec_reduced = false;           // the context's own stats vector went through an all-reduce since the last bsk_stats_reset

    // ---- device state shared by the ops -----------------------------------
    int num_cus = 0;
    uint64_t min_range_bytes = 64 * 1024;
    bool use_dpp = true;          // BSK_SCAN=shfl selects the ds_bpermute scan
    bool stats_a_dense = false;   // BSK_STATS_A=dense: FASTQ `stats -a` by running counters (dense path) instead of line roles
    uint64_t* d_anchors = nullptr;  // [cap_ranges + 1] + queue word
    uint32_t cap_ranges = 0;
    hipStream_t own_stream = nullptr;
    uint64_t* d_rng = nullptr;       // FASTA stats: r_head[cap] ++ r_tail[cap] ++ (u32) r_flags[cap]
    uint32_t rng_cap = 0;

    // ---- Stats ----------------------------------------------------------------
    bsk::Alphabet alphabet = bsk::AB_NONE;  // forced by -t, else AB_NONE
    // ... or the GUESS of a partition's first record, held for the calls that feed that partition piece by piece
    // (bsk_run_to_store, switch pin_alphabet): that is not a -t -- SeqTransform.Before switches validation on for a GIVEN
    // type only (seq.go:66-72).  Round 6: `seq` on a streamed shard whose first record read as DNA refused the RNA behind it.
    bool alphabet_guessed = false;
    bool alphabet_given() const { return !alphabet_guessed && !(alphabet == bsk::AB_NONE || alphabet == bsk::AB_UNLIMIT); }
    uint32_t hist_cap = 1u << 16;
    uint64_t* d_vec = nullptr;       // ctx-owned stats vector
    uint64_t* d_status = nullptr;    // [0] err flags [1] overflow count            (= d_ctl)
    // ---- the control block: every scalar a call reads back lives in ONE allocation and comes to the host with ONE copy
    // into pinned memory (ctl_readback).  Round 3 fetched total / kept / long count / status with a copy each: ~17 us of
    // stream time per copy, twenty of them in a `grep` call -- 0.6 ms of a 4.4 ms call.
    //   d_ctl[0..7]  = d_status ([0] flags [1] stats overflow count [2] collect scratch [3] rmdup overflow-list count;
    //                  zeroed at the entry of every run)    d_ctl[8..15] = d_counter    d_ctl[16..31] = d_fin (FIN_* below)
    uint64_t* d_ctl = nullptr;
    uint64_t* h_ctl = nullptr;       // pinned mirror [CTL_WORDS]
    uint64_t* d_fin = nullptr;
    static constexpr int CTL_WORDS = 32;
    enum { FIN_TOTAL = 0, FIN_KEPT = 1, FIN_LONG_COUNT = 2, FIN_LONG_MAX = 3, FIN_OTHER = 4, FIN_TABLE_N = 5, FIN_AUX0 = 6, FIN_AUX1 = 7 };
    uint64_t fin(int k) const { return h_ctl[16 + k]; }
    uint64_t status_word() const { return h_ctl[0]; }
    // ---- head (and tail) of the shard of the running call, in pinned memory: ONE copy per call serves the alphabet
    // guess, the record density, the multi-line check and the light FASTA table (each took its own copy before)
    uint8_t* h_head = nullptr;       // [HEAD_BYTES + TAIL_BYTES]
    static constexpr size_t HEAD_BYTES = 256 * 1024, TAIL_BYTES = 4096;
    const uint8_t* head_of = nullptr;  // the shard the sample belongs to (null: none)
    size_t head_n = 0, head_len = 0, tail_len = 0;
    uint64_t call_gen = 0, head_gen = ~0ull;  // a sample lives for one call
    uint64_t alpha_gen = ~0ull;               // ... and so does the alphabet guessed from it (partition_alphabet)
    const uint8_t* alpha_of = nullptr;
    int alpha_format = -1, alpha_value = 0;
    uint64_t* d_overflow = nullptr;
    uint64_t overflow_cap = 0;
    int qual_offset = 33;
    std::vector<uint8_t> first_bytes;  // head of the lowest-pid shard (type guess, Take(1))
    int64_t first_pid = INT64_MAX;
    int first_format = -1;
    uint64_t last_overflow_total = 0;  // slot [5] of the vector the last bsk_stats_collect read (bsk_stats_overflow_total)
    std::string type_if_F;  // alphabet name the driver would guess from Take(1) (bigseqkit/stats.go:117-129)

    // ---- record table + per-record scratch (seq, grep, ...) ---------------------
    bsk::RecordTable table;          // ctx-owned arrays, grown on demand
    // custom --id-regexp (neither the default nor the --id-ncbi one): program of the position-reporting matcher and
    // the per-record ID spans of the last indexed shard (ops_idre.hip)
    bool id_custom = false;
    bsk::VmProgram id_prog;
    bsk::VmProgram* d_id_prog = nullptr;
    uint32_t* d_id_off = nullptr;
    uint32_t* d_id_len = nullptr;
    uint64_t id_cap = 0;
    uint64_t avg_record_bytes = 0;   // bytes per record in the head of the last indexed shard (0: unknown)
    bsk::RecordTable sparse;         // one-pass index: per-range slices, compacted into `table`
    uint64_t* d_range_count = nullptr;  // [cap_ranges]
    uint64_t* d_range_base = nullptr;   // [cap_ranges + 1]
    bsk::RangePart* d_parts = nullptr;  // [parts_cap] FASTA: parts of records that span ranges
    uint32_t parts_cap = 0;
    uint32_t* d_out_len = nullptr;      // [table.cap]
    uint64_t* d_out_off = nullptr;      // [table.cap + 1]
    uint64_t* d_scan_tmp = nullptr;
    uint64_t scan_tmp_cap = 0;
    uint64_t out_len_cap = 0;
    uint8_t* d_out = nullptr;           // output text of the last run
    uint64_t out_cap = 0;
    uint8_t* d_out_alt = nullptr;       // second output buffer of bsk_run_to_store (drained while the next chunk computes)
    uint64_t out_alt_cap = 0;
    uint8_t* d_group = nullptr;         // scratch of group_resolve (sorted keys, permutations, bucket bounds)
    uint64_t group_cap = 0;
    uint8_t* d_norm = nullptr;          // a multi-line FASTQ shard rewritten as 4-line FASTQ (ops_mlfq.hip)
    uint64_t norm_cap = 0;
    bool norm_active = false;           // the operator is running on d_norm
    bool fastq_multiline = false;       // stats: the head of the lowest shard showed wrapped records
    uint64_t* d_seg_src = nullptr;      // segmented copy (ops_segcopy.hip): source address per record (+ one counter)
    uint64_t seg_src_cap = 0;
    uint32_t* d_seg_first = nullptr;    // ... and the segment of the first byte of every 4 KiB output tile
    uint64_t seg_first_cap = 0;
    uint8_t* d_norm2 = nullptr;         // several rewritten FASTQ texts back to back (pair / common / concat on multi-line FASTQ)
    uint64_t norm2_cap = 0;
    uint64_t last_kernel_flags = 0;     // what kernel_error_to_status saw last (the strict FASTQ reader's complaints send a shard to the multi-line reader)
    uint8_t* d_slices = nullptr;        // per-range output slices of the names pass (stream_names.hip)
    uint64_t slices_cap = 0;
    uint64_t* d_names_aux = nullptr;    // [2 * (nranges + 2)]: bytes per range, scanned record counts
    uint64_t names_aux_cap = 0;
    uint8_t* d_lut = nullptr;           // 256-byte byte map (seq)
    double* d_qual_err = nullptr;       // 256 doubles (seq -Q/-R)
    uint64_t* d_counter = nullptr;      // scratch counter
    // FASTA text view (text_dev.hpp)
    uint32_t* d_text_w = nullptr;
    uint64_t* d_lin_off = nullptr;
    uint8_t* d_lin = nullptr;
    uint64_t text_cap = 0, lin_cap = 0;
    bool text_ready = false;
    // translate
    uint8_t* d_codon = nullptr;   // 4096 + 4096 bytes (aa table, start table)
    std::vector<int> frames;
    bool codon_ready = false;        // d_codon holds the tables of this context's options
    bool translate_uniform_ok = true;  // FASTA: try the table-free pass on records that all look alike first (UniformLayout)
    bool translate_stream_ok = true;  // FASTA of long records: try the one-pass translation (k_translate_stream) before any table
    // (round 6, ADVICE r05: a misfit -- one chromosome-sized record, one range with more than 512 records -- used to send
    // the context to the table paths FOR GOOD, and long-lived contexts, the pipe pool or bsk_run_to_store's chunks, lost the
    // fast path silently.  Now the next `translate_stream_skip` calls take the tables, then the one-pass kernel is tried
    // again; the back-off doubles with every misfit in a row (8, 16, ... 1 024 calls) so that a file of chromosomes does
    // not pay the wasted pass more than a few times.  bsk_profile_dump reports the misfits as "translate_stream_fallback".)
    uint32_t translate_stream_skip = 0, translate_stream_backoff = 8;
    uint64_t translate_stream_fallbacks = 0;
    bool translate_light_ok = true;  // FASTA: try the record table from the '>' bytes alone first (stream_fasta_light.hip)
    uint8_t* d_redo = nullptr;    // one byte per record: left by k_translate_wide to k_translate_frames4
    uint64_t redo_cap = 0;
    // rmdup
    uint64_t* d_keys = nullptr;
    uint64_t keys_cap = 0;
    uint64_t* d_table = nullptr;  // cap slots {key, ~first record} (key_table)
    uint64_t table_cap = 0;
    // multi-GPU rmdup: second keys, owner-side table, the shard the keys phase indexed
    uint64_t* d_keys2 = nullptr;
    uint64_t keys2_cap = 0;
    uint64_t* d_keys_sparse = nullptr;  // k1 ++ k2 in the per-range slices of the hashing index pass (stream_rmdup.hip)
    uint64_t keys_sparse_cap = 0;
    uint32_t* d_ovf = nullptr;          // rmdup: records whose first-of-key disagrees in the second key (+ a counter word)
    uint64_t ovf_cap = 0;
    uint64_t* d_own = nullptr;   // table_keys[cap] ++ table_first[cap] ++ table_k2[cap]
    uint64_t own_cap = 0;
    const uint8_t* dist_buf = nullptr;
    size_t dist_n = 0;
    int dist_format = -1;
    // round 6, the text comparison of duplicates whose survivor lives on another rank (ops_rmdup_xcheck.hip): this rank's
    // requests and the subjects that go with them, the records whose text differs from their survivor's ("flagged"), and
    // those of them that survive after the exact settlement (first of their TEXT over all ranks)
    uint64_t* d_xreq = nullptr;
    uint64_t xreq_cap = 0;          // words
    uint8_t* d_xtext = nullptr;
    uint64_t xtext_cap = 0;
    uint32_t* d_xflag = nullptr;    // [0] count, entries from [1]
    uint64_t xflag_cap = 0;
    uint32_t* d_xres = nullptr;
    uint64_t xres_cap = 0;
    uint32_t xres_n = 0;
    uint64_t x_m_req = 0, x_base = 0;
    const uint64_t* x_send = nullptr;
    const uint8_t* x_reply = nullptr;
    const uint64_t* x_surv = nullptr;
    bool dist_xchecked = false;     // bsk_rmdup_dist_xapply ran: the emit does not compare the local pairs again
    uint64_t dist_cross_pairs = 0;  // duplicates of the last exchange whose survivor lives on another rank (their text went there)
    std::vector<uint32_t> x_flag_host;  // the flagged records of this shard, ascending
    std::string x_flag_blob;            // ... serialised for the exchange (bsk_rmdup_dist_flagged_get)
    // -d / -D: what RmDupCheck accumulates until After() (rmdup.go:100-104, 224-238)
    std::string dup_seqs, dup_nums;
    uint64_t removed = 0;
    bool side_written = false;
    // grep / locate: patterns after Before() (CLI order, duplicates removed, lower-cased with -i)
    std::vector<std::string> patterns;
    uint8_t* d_pat = nullptr;
    uint32_t* d_pat_off = nullptr;
    uint64_t pat_cap = 0, pat_off_cap = 0;
    std::string pat_sig, ftab_sig;  // what d_pat / d_ftab hold (the uploads are skipped when a call needs the same again)
    uint8_t* d_ftab = nullptr;  // pair-hash table of the fused pattern filter (stream_filter.hpp): u32 tab[slots] ++ u16 ent[slots]
    uint64_t ftab_cap = 0;
    std::vector<std::string> pattern_names;  // locate: names as given (== the -p text, or the FASTA name with -f)
    std::vector<std::string> pattern_disp;   // locate -r: the expressions (pattern column); patterns[] then only carries the match length
    // class patterns (-d, -m, -F): one 256-bit accept set per pattern position (pattern_match_dev.hpp)
    std::vector<std::vector<std::array<uint32_t, 8>>> pattern_cls;
    bool general = false;     // match through pattern_cls instead of the exact byte compare
    int max_mm = 0;           // -m
    bool fmi_order = false;   // locate -m / -F: all patterns on '+', then all on '-' (locate.go:208-391)
    uint32_t* d_cls = nullptr;
    uint64_t cls_cap = 0;
    // grep by ID / name with many patterns: hash set on the device
    uint64_t* d_set_keys = nullptr;
    uint32_t* d_set_idx = nullptr;
    uint64_t set_keys_cap = 0, set_idx_cap = 0, set_slots = 0;
    // -r: compiled position automata (regex_nfa.hpp)
    std::vector<bsk::RegexProgram> regexes;
    std::vector<bsk::RegexProgram> locate_pre;  // locate -r (matcher): the same expressions as boolean automata -- which records match at all
    bsk::RegexProgram* d_regex = nullptr;
    uint64_t regex_cap = 0;
    // locate -r with matches of variable length: programs of the position-reporting matcher (regex_vm.hpp)
    bool locate_vm = false;
    bool grep_vm = false;    // grep -r: an expression the bit-parallel automaton does not take (\b, > 64 positions): all go to the thread-list matcher (vm_progs)
    std::vector<bsk::VmProgram> vm_progs;
    bsk::VmProgram* d_vm_progs = nullptr;
    uint64_t vm_progs_cap = 0;
    bool patterns_uploaded = false;  // exact patterns + set do not depend on the shard's alphabet when by name
    uint8_t* d_names = nullptr;
    uint32_t* d_names_off = nullptr;
    uint64_t names_cap = 0, names_off_cap = 0;
    // subseq --gtf / --bed: the first feature of every (lower-cased) sequence name (subseq.go:319-526)
    struct Feature { std::string name_lower, suffix; int64_t s, e; bool minus; };
    std::vector<Feature> features;
    bool features_uploaded = false;
    uint8_t* d_feat = nullptr;             // one allocation holding all feature arrays
    uint64_t feat_cap = 0;
    uint64_t feat_off[9] = {0};            // byte offsets of the arrays inside d_feat
    uint64_t feat_slots = 0;
    uint32_t long_thresh = 1u << 20;        // BSK_LONG_BYTES overrides (tests)
    uint64_t flat_long_count = 0;           // prepare_text(flatten): records of >= flat_long_thresh bases, listed in d_long_list
    uint32_t flat_long_thresh = 0;
    uint64_t dist_local_pairs = 0;  // multi-GPU rmdup: duplicates of the last emit whose survivor was in the same shard (byte-compared)
    uint64_t long_count = 0, long_max = 0;  // records with >= SEQ_LONG_THRESH output bytes in the last finish_sizes()
    uint32_t* d_long_list = nullptr;
    uint64_t long_list_cap = 0;
    // locate, long records: cells (pattern, strand, chunk) -- see LocateParams
    uint8_t* d_cellmeta = nullptr;         // cell counts[count] ++ cellbase[count + 1]
    uint8_t* d_cells = nullptr;            // cell_bytes[cells] ++ cell_off[cells + 1]
    uint64_t cells_cap = 0, cellmeta_cap = 0;
    uint32_t* d_hit_list = nullptr;        // locate: records with rows
    uint64_t hit_list_cap = 0;
    int64_t cur_pid = 0;                   // partition index of the running Call()
    // range / head / duplicate (ops_records.hpp)
    int64_t range_start = 0, range_end = 0;  // as the driver computes them (bigseqkit/range.go:46-86)
    bool range_needs_count = false, range_resolved = false;
    int64_t cur_first_record = 0;          // index of the shard's first record in the whole input
    uint64_t cur_base_offset = 0;          // faidx: file offset of the shard
    uint8_t* d_arena = nullptr;            // grow-only scratch of sort / rename / faidx (carved per call)
    uint64_t arena_cap = 0;
    uint32_t* d_tile_first = nullptr;      // record of the first byte of every output tile (k_records_copy)
    uint64_t tile_first_cap = 0;
    int region_start = 0, region_end = 0;  // parsed -R / -r
    bool region_on = false;
    uint64_t last_count = 0;               // grep -C result of the last run

    // ---- round 6: a result that is still a list of slices (bsk_out.d_seg_*; include/bsk.h) and what makes it one block
    struct PendingOut {
        int kind = 0;              // 0: none; 1: segments of the shard (k_seg_copy); 2: per-range slices of a streaming pass (k_names_compact)
        uint64_t total = 0, records = 0, nseg = 0;
        const uint64_t* seg_src = nullptr;
        const uint64_t* seg_off = nullptr;
        const uint32_t* first4k = nullptr;  // kind 1 (kind 2: built when a consumer gathers pieces)
        const uint8_t *lo = nullptr, *hi = nullptr;
        uint64_t slice_cap = 0;             // kind 2
        uint64_t gen = 0;                   // the run that produced it (call_gen of the producing call's scope)
    } pend_out;
    bool out_slices = false;                // switch "out" = "slices"
    bool force_contiguous = false;          // the running call needs one block whatever the switch says (bsk_run_to_store's chunks, operators that read the output again)
    uint64_t* d_slice_src = nullptr;        // kind 2: source address per range
    uint64_t slice_src_cap = 0;

    // ---- staging for host-resident shards ------------------------------------
    void* drainer = nullptr;   // pinned staging + events of the output drain (store.cpp: Drainer)
    uint8_t* pinned[2] = {nullptr, nullptr};
    uint8_t* d_stage[2] = {nullptr, nullptr};
    size_t stage_cap = 0;
    hipStream_t copy_stream[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};  // copy of buffer b finished
    hipEvent_t stage_free[2] = {nullptr, nullptr};  // kernels reading buffer b finished
    size_t stage_cap_b[2] = {0, 0};

    // ---- profiling (bench.py roofline leg) -----------------------------------
    bool profile = false;
    struct Prof { double ms = 0; uint64_t launches = 0; };
    std::map<std::string, Prof> prof;
    struct PendingEv { const char* name; hipEvent_t a, b; };
    std::vector<PendingEv> pending;  // event pairs recorded around launches, not yet read

    void set_error(const std::string& m) const {
        std::lock_guard<std::mutex> g(mu);
        last_error = m;
    }

    // ---- the reference's log.Warn / log.Info lines (bigseqkit-lib/seq.go:53-68, grep.go:66-206, locate.go:58-144,
    // subseq.go:99-158): written to stderr as "[WARN] ..." / "[INFO] ..." and kept for bsk_log_text().  Which messages
    // --quiet suppresses is the reference's choice, message by message (the callers pass `unless_quiet`).
    mutable std::string log_text;
    void log(const char* level, const std::string& m, bool unless_quiet = false) const {
        if (unless_quiet && opts.cb("Quiet")) return;
        const std::string line = std::string("[") + level + "] " + m + "\n";
        {
            std::lock_guard<std::mutex> g(mu);
            log_text += line;
        }
        fputs(line.c_str(), stderr);
    }
    void warn(const std::string& m, bool unless_quiet = false) const { log("WARN", m, unless_quiet); }
    void info(const std::string& m, bool unless_quiet = false) const { log("INFO", m, unless_quiet); }
};

// the scope of one C-ABI call that runs on the context's device state (BSK_ENTER in capi.cpp / store.cpp)
struct bsk_call_scope {
    bsk_ctx* c;
    bool owns;
    explicit bsk_call_scope(bsk_ctx* c_) : c(c_), owns(false) {
        bool expected = false;
        owns = c->busy.compare_exchange_strong(expected, true, std::memory_order_acquire);
        if (owns) ++c->call_gen;
    }
    ~bsk_call_scope() { if (owns) c->busy.store(false, std::memory_order_release); }
    bsk_call_scope(const bsk_call_scope&) = delete;
    bsk_call_scope& operator=(const bsk_call_scope&) = delete;
};
#define BSK_BUSY_TEXT "libbsk: context busy: another call is running on this context (one context per caller thread, include/bsk.h)"

// optional HIP-event bracket around one launch (bsk_profile_enable): read back by bsk_profile_read under `name`
struct Timed {
    bsk_ctx* c;
    const char* name;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(bsk_ctx* c_, const char* n, hipStream_t s) : c(c_), name(n), st(s) {
        if (c->profile) {
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a, st);
        }
    }
    ~Timed() {
        if (c->profile && a) {
            hipEventRecord(b, st);
            c->pending.push_back({name, a, b});
        }
    }
};

// Ranges of the persistent streaming kernels (k_stats, k_index): about 512 KiB each -- measured on 3-100 GB shards,
// ranges below ~200 KiB pay their start-up (anchor, LDS window, validation) and fewer than ~4 ranges per wave leave a
// tail: 100 GB 17.7 -> 17.1 ms with 16 ranges per wave instead of 4, small shards unchanged (scripts/sweep_ranges.sh).
// min_range_bytes (tuning min_range_bytes, tests) bounds the range size from below; `pinned` (tuning ranges_per_wave) pins the count.
inline uint64_t pick_nranges(uint64_t n, uint64_t waves, uint64_t min_range_bytes, int pinned = 0) {
    uint64_t nr;
    if (pinned) nr = waves * (uint64_t)pinned;
    else {
        nr = n / (512u * 1024u);
        const uint64_t lo = std::min<uint64_t>(waves * 2, n / (128u * 1024u));  // at least two ranges per wave while they stay >= 128 KiB
        nr = std::max(nr, lo);
        nr = std::min<uint64_t>(nr, waves * 16);
    }
    nr = std::min<uint64_t>(nr, n / std::max<uint64_t>(1, min_range_bytes));
    return std::max<uint64_t>(1, nr);
}

// the caller thread's error text (bsk_global_error) for entry points that have no context (capi.cpp)
namespace bsk {
int global_error_set(int code, const std::string& m);
}
