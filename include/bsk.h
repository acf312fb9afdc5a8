/* ============================================================================
 * bsk.h -- C ABI of libbsk.so, the MI355X (gfx950) engine behind BigSeqKit's
 * per-record map/filter hot path.
 *
 * Every entry point replaces one executor-side operator of the reference's Go
 * plugin (bigseqkit.so, looked up as "New"+Name by IgnisHPC; see
 * /root/reference/bigseqkit/helper.go:21-23).  Citations are file:line relative
 * to /root/reference/.  The lifecycle of the reference operators
 *     Before(ctx)  ->  Call(partition)...  ->  After(ctx)
 * maps to
 *     bsk_create() ->  bsk_<op>_run()...   ->  bsk_destroy().
 * Options travel exactly as in the reference: the JSON text produced by
 * bigseqkit.OptionsToString (bigseqkit/helper.go:47-55), i.e. the Go struct of
 * the command with exported field names and a nested "Config" (KitConfig).
 *
 * Plain C: pointers and sizes only; no torch / HIP types in any signature
 * (streams are passed as void* = hipStream_t, device buffers as void*).
 * All functions return 0 (BSK_OK) on success, else a BSK_ERR_* code; the
 * message (the reference's own error text where one exists) is available from
 * bsk_last_error(ctx) or, for failures that have no ctx, bsk_global_error().
 * There is NO CPU fallback: without a usable HIP device every compute entry
 * point fails with BSK_ERR_NO_DEVICE.
 *
 * THREADS.  The reference calls Call() of ONE operator struct from Threads()
 * goroutines at once (bigseqkit-lib/helper.go:413-416 per-thread channels,
 * rmdup.go:100,224 a mutex around the shared maps).  A bsk_ctx is NOT that
 * struct: it owns the device state of a call -- output buffer, record table,
 * control block, staging buffers -- so the rule is
 *     ONE CONTEXT PER CALLER THREAD (create them from the same options JSON),
 *     any number of contexts per device, any thread may use any context,
 *     but only one call at a time runs on a context.
 * A second call that arrives while one is running on the same context is
 * refused with BSK_ERR_INVALID_ARG, never raced -- the text "context busy" is in
 * the REFUSED caller's bsk_global_error() (thread-local; the context's own error
 * text and per-call state belong to the call that is running) --; this covers
 * every entry point that touches the context's device state (the *_run
 * family, bsk_stats_*, bsk_index_*, bsk_out_to_host, bsk_store_put,
 * bsk_run_to_store, the bsk_rmdup_dist_* phases, bsk_profile_read).
 * The bsk_out a run returns points into the context's output buffer and is
 * valid until the next run on that context.  Objects meant to be SHARED by
 * the threads of an executor: a bsk_store (FileStore: parts may arrive from
 * any thread in any order; internally locked) and the input shard (read-only).
 * bsk_global_error() is thread-local; bsk_last_error(ctx) belongs to the ctx.
 * Results that the reference merges across threads merge the same way here:
 * stats vectors add (bsk_stats_merge, or one device vector passed to several
 * contexts' runs on ONE stream), grep counts add, rmdup needs ONE context for
 * the whole partition (duplicates are global), the store orders the parts.
 * ==========================================================================*/
#ifndef BSK_H
#define BSK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSK_OK 0
#define BSK_ERR_INVALID_ARG 1 /* bad pointer / size / op name                         */
#define BSK_ERR_OPTS 2        /* option validation failed (reference's Before() error) */
#define BSK_ERR_FORMAT 3      /* malformed record (reference's Call() error)           */
#define BSK_ERR_UNSUPPORTED 4 /* input layout the HIP path does not accept (PARITY.md) */
#define BSK_ERR_HIP 5         /* HIP runtime error                                     */
#define BSK_ERR_NO_DEVICE 6   /* no gfx950 device visible                              */
#define BSK_ERR_CAPACITY 7    /* caller-provided output buffer too small               */
#define BSK_ERR_OVERFLOW_EXCHANGE 8 /* bsk_stats_collect on a vector reduced over ranks: slot [5] counts more lengths >=
                                     * hist_cap than this context's list holds -- exchange the lists, collect again */

#define BSK_FORMAT_FASTA 0
#define BSK_FORMAT_FASTQ 1

typedef struct bsk_ctx bsk_ctx;

/* ---- library ------------------------------------------------------------ */
int bsk_version(void);
/* number of visible HIP devices (0 when none); never fails */
int bsk_device_count(void);
const char* bsk_global_error(void);         /* thread-local */
const char* bsk_last_error(const bsk_ctx*); /* per context  */

/* ---- operator lifecycle ---------------------------------------------------
 * op_name is the reference's plugin symbol without "New":
 *   "Stats"            NewStats            bigseqkit-lib/stats.go:16
 *   "SeqTransform"     NewSeqTransform     bigseqkit-lib/seq.go:17
 *   "Grep"             NewGrep             bigseqkit-lib/grep.go:24
 *   "Locate"           NewLocate           bigseqkit-lib/locate.go:19
 *   "SubseqTransform"  NewSubseqTransform  bigseqkit-lib/subseq.go:22
 *   "Translate"        NewTranslate        bigseqkit-lib/translate.go:21
 *   "RmDup"            NewRmDupPrepare + NewRmDupCheck  bigseqkit-lib/rmdup.go:23,92
 * bsk_create == Before(): decodes opts_json (StringToOptions, helper.go:57-66),
 * fills defaults (setDefaults of the command), validates with the reference's
 * messages, uploads pattern / codon tables.  device < 0: options are parsed and
 * validated only (no HIP call is made; usable without a GPU). */
int bsk_create(const char* op_name, const char* opts_json, int device, bsk_ctx** out);
void bsk_destroy(bsk_ctx* ctx); /* == After() */
/* canonical JSON of the options after defaults (the text the reference's
 * executor would see); returned pointer lives as long as ctx */
const char* bsk_opts_json(const bsk_ctx* ctx);
/* The reference's log.Warn / log.Info lines of this context so far ("[WARN] ...\n[INFO] ...\n"; they are also written
 * to stderr as they occur): option advice of Before() -- bigseqkit-lib/seq.go:52-69, grep.go:57-98, 140-207,
 * locate.go:50-70, 96-98, 143-145, subseq.go:98-100, 127-133, 157-159.  Config.Quiet suppresses exactly the messages the
 * reference guards with it. */
const char* bsk_log_text(const bsk_ctx* ctx);
/* Run-time switches of ONE context (which kernel variant runs, thresholds): key = a name of INTEGRATION.md "Switches"
 * ("segcopy", "rmdup_keys", "min_range_bytes", ...; the old environment spelling "BSK_SEGCOPY" is accepted), value = its
 * text, NULL = unset.  A context takes the process environment's BSK_<NAME> values once, in bsk_create; nothing reads the
 * environment on the call path.  Unknown keys: BSK_ERR_INVALID_ARG. */
int bsk_ctx_set(bsk_ctx* ctx, const char* key, const char* value);

/* ---- record boundaries: PlainFile(path, delim) + ReadFixer ---------------
 * bigseqkit/helper.go:148-178, bigseqkit-lib/helper.go:41-66.
 * Finds, on the HOST, the first record start at or after `from` in a window of
 * file bytes (used to cut a file into per-GPU shards that begin on a record).
 * FASTQ: four lines per record or wrapped over several lines (SeqParser reads both,
 * bigseqkit-lib/helper.go:252-269); the window should hold the three records behind
 * the answer (1 MiB does for reads).  Returns n if there is none. */
int bsk_find_record_start(const uint8_t* buf, size_t n, size_t from, int format, size_t* out);

/* ---- Stats  (bigseqkit-lib/stats.go:27-117, StatsReduce :128-137) ---------
 * The per-partition result of Stats.Call -- a map[int64]int64 of
 * length -> count plus keys -1 (Q20), -2 (Q30), -3 (gap sum), -4 (type) --
 * is kept DEVICE-RESIDENT as one flat vector of uint64 ("stats vector"):
 *     [0]=q20 [1]=q30 [2]=gap_sum [3]=num_records [4]=error flags
 *     [5]=number of overflow lengths [6]=sum of lengths [7]=reserved
 *     [8 + L] = count of records with sequence length L, 0 <= L < hist_cap
 * so that StatsReduce across GPUs is ONE sum all-reduce (RCCL) on that vector.
 * Lengths >= hist_cap go to a ctx-owned overflow list merged by
 * bsk_stats_collect(); slot [5] counts them, so after a reduction over ranks the
 * collecting context must hold that many list entries (bsk_stats_overflow_get /
 * _add move the lists between ranks; bsk_stats_collect fails otherwise instead of
 * dropping chromosome-sized records).  Slot [4] is a per-shard diagnostic. */
#define BSK_STATS_HDR 8
size_t bsk_stats_vector_len(const bsk_ctx* ctx);
/* One Stats.Call over a shard.  `shard` holds n bytes of FASTA/FASTQ text
 * starting on a record; on_device != 0: device pointer (HBM-resident, the
 * measured path), else host pointer (staged through a pinned double buffer).
 * d_vec: device stats vector to ACCUMULATE into (caller zeroes it once, e.g. a
 * torch tensor), or NULL to use the ctx-owned vector.  stream: hipStream_t or
 * NULL.  Asynchronous w.r.t. the host when on_device; errors detected by the
 * kernels are reported by bsk_stats_collect(). */
int bsk_stats_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* d_vec,
                  void* stream);
/* Device-buffer plumbing for callers that do not link the HIP runtime themselves (the cgo shim, the C++ CLI):
 * shards and operator outputs can be chained on the device (the output text of seq / grep / subseq / rmdup / translate
 * is FASTA / FASTQ again) without a host round trip.  Buffers live on the current device of the calling thread
 * (device 0 unless a context of another device ran last). */
#define BSK_COPY_H2D 1
#define BSK_COPY_D2H 2
#define BSK_COPY_D2D 3
int bsk_device_select(int device); /* the current device of the calling thread (hipSetDevice): where bsk_device_alloc allocates */
void* bsk_device_alloc(size_t n);
void bsk_device_free(void* p);
int bsk_device_copy(void* dst, const void* src, size_t n, int kind); /* synchronous */

/* Pinned (page-locked) host memory for host-resident shards: read the file into such a buffer and the H2D copies of
 * bsk_stats_run(on_device = 0) are DMA at PCIe rate, overlapped with the kernels chunk by chunk (256 MiB record-aligned
 * chunks, two device buffers; BSK_STAGE_BYTES overrides the chunk size).  NULL when the allocation fails. */
void* bsk_host_alloc(size_t n);
void bsk_host_free(void* p);

/* ReadFASTA[N] / ReadFASTQ[N] (bigseqkit/helper.go:148-178: worker.PlainFile[N] gives every executor its byte ranges of the
 * file): the bytes [offset, offset + n) of the open file `fd` into a NEW buffer on `device` (*d_shard, bsk_device_free).
 * `threads` readers (<= 0: 8) pread() pieces of 16 MiB (BSK_SHARD_PIECE_BYTES) into two pinned buffers each and copy them
 * on streams of their own: reading the file and crossing PCIe overlap, 32 MiB per reader is pinned.  The caller cuts the
 * file at record starts (bsk_find_record_start on a window around size * k / world).  Errors: bsk_global_error(). */
int bsk_shard_load(int fd, uint64_t offset, size_t n, int device, int threads, void** d_shard);
int bsk_stats_reset(bsk_ctx* ctx, void* stream); /* zero the ctx-owned vector, error flags and overflow list */
/* The context's list of sequence lengths >= hist_cap (the part of the reference's map[int64]int64,
 * bigseqkit-lib/stats.go:86, that does not fit the dense vector).  _get copies it to the host (cap 0 + NULL: size
 * query); _add appends lengths that another rank's context collected, before bsk_stats_collect on a reduced vector. */
int bsk_stats_overflow_get(bsk_ctx* ctx, uint64_t* lens, size_t cap, size_t* n_out);
int bsk_stats_overflow_add(bsk_ctx* ctx, const uint64_t* lens, size_t n);
/* Synchronise, check the kernels' error flags (BSK_ERR_FORMAT /
 * BSK_ERR_UNSUPPORTED) and convert a stats vector (d_vec or the ctx-owned one)
 * into the reference's map form, sorted by key.  Key -4 is computed from the
 * first record seen by this ctx (bigseqkit-lib/stats.go:106-114). */
int bsk_stats_collect(bsk_ctx* ctx, const void* d_vec, int64_t* keys, int64_t* vals, size_t cap, size_t* n_out);
/* Slot [5] of the vector the LAST bsk_stats_collect of this context read (0 before any): after an all-reduce it is the
 * same number on every rank, so "is an exchange of the overflow lists needed" is decided by all ranks alike and WITHOUT a
 * device round trip of its own (the collect already brought the vector to the host).  Protocol of a reduced step:
 *   all-reduce -> collect; if the total is non-zero on a world of more than one rank: every rank exchanges its list
 *   (bsk_stats_overflow_get / _add), then collects again.  A collect that finds fewer list entries than slot [5] counts
 *   returns BSK_ERR_OVERFLOW_EXCHANGE (and still records the total). */
int bsk_stats_overflow_total(const bsk_ctx* ctx, uint64_t* total);
/* Same conversion for a stats vector that already lives in HOST memory (e.g. after a
 * reduction done elsewhere); needs no device.  first_record (may be NULL) is the text of
 * the first record of partition 0, used for the type column exactly like Take(1). */
int bsk_stats_collect_host(bsk_ctx* ctx, const uint64_t* h_vec, size_t vec_len, const uint8_t* first_record,
                           size_t first_len, int format, int64_t* keys, int64_t* vals, size_t cap, size_t* n_out);
/* StatsReduce.Call on two host maps (sums; PARITY.md Q2) */
int bsk_stats_merge(const int64_t* ka, const int64_t* va, size_t na, const int64_t* kb, const int64_t* vb, size_t nb,
                    int64_t* keys, int64_t* vals, size_t cap, size_t* n_out);

/* driver side: Stats() bigseqkit/stats.go:75-166 and StatInfo :290-311 */
typedef struct {
    char type[16]; /* "DNA" "RNA" "Protein" "Unlimit" "" ... */
    uint64_t num, len_sum, gap_sum, len_min, len_max, n50;
    int64_t l50;
    double len_avg, q1, q2, q3, q20, q30;
} bsk_statinfo;
int bsk_stats_finalize(const bsk_ctx* ctx, const int64_t* keys, const int64_t* vals, size_t n, bsk_statinfo* out);
/* StatsString() bigseqkit/stats.go:168-288 ; out is NUL-terminated */
int bsk_stats_string(const bsk_ctx* ctx, const char* name, const char* format, const bsk_statinfo* info, char* out,
                     size_t cap);

/* ---- record-producing operators ------------------------------------------
 * The reference operators return []string, one element per output record,
 * which FileStore writes as element + "\n" (bigseqkit-lib/helper.go:447).
 * Here the result of one Call() is that byte stream, left in a ctx-owned DEVICE
 * buffer (valid until the next run on the same ctx or bsk_destroy). */
typedef struct {
    void* d_data;     /* device pointer, `len` bytes -- or NULL while the text is an ordered list of slices (below) */
    size_t len;
    uint64_t records; /* number of elements (output records) */
    /* Round 6 -- the result as ORDERED SLICES (switch "out" = "slices", bsk_ctx_set; default "contiguous": these are 0).
     * The reference's Call returns []string whose elements are (for seq -n, subseq, rmdup ...) Go strings that share the
     * bytes of the partition or of per-goroutine buffers: nothing is moved into one block
     * (bigseqkit-lib/subseq.go:167-225, seq.go:81-269, rmdup.go:200-222).  With "slices" an operator whose output text
     * already sits somewhere in HBM in output order -- the survivors of rmdup inside the input shard, the per-range
     * buffers the streaming passes of `seq -n` and `subseq -r` write, the whole FASTQ records that `seq` (length / quality
     * filters) and `grep` KEEP (segments of the shard; a dropped record is a segment of no bytes) -- returns that: segment k is the
     * d_seg_off[k + 1] - d_seg_off[k] bytes at device address d_seg_src[k]; the text is their concatenation (len bytes).
     * bsk_out_to_host, bsk_store_put and the Go shim consume slices as they are (gathered piece by piece into the
     * staging buffers of the drain); bsk_out_materialize makes the one block for a consumer that needs it (the next
     * operator of a pipe).  LIFETIME: slices of rmdup / seq / grep point into the caller's SHARD -- it must stay until the output is
     * consumed (with "contiguous" it may go as soon as the run returns); like d_data they die with the context's next run. */
    const uint64_t* d_seg_src; /* device: [n_segments] source addresses */
    const uint64_t* d_seg_off; /* device: [n_segments + 1] offsets in the output text */
    uint64_t n_segments;       /* 0: d_data holds the text */
} bsk_out;
/* copy an operator result to host memory (synchronises) */
int bsk_out_to_host(bsk_ctx* ctx, const bsk_out* out, void* dst, size_t cap);
/* a result that is still a list of slices becomes ONE block in the context's output buffer (today's second move of the
 * text: k_seg_copy / k_names_compact); out->d_data is set, the slice fields are cleared.  No-op on a contiguous result. */
int bsk_out_materialize(bsk_ctx* ctx, bsk_out* out, void* stream);

/* ---- record table: SeqParser.Read (bigseqkit-lib/helper.go:219-325) --------
 * Builds, for a device- or host-resident shard, the SoA table of record slices the
 * per-record operators work on.  Exposed for tests and for callers that want the
 * offsets (faidx-style); the operators below build it themselves.
 * starts[i] = offset of the marker byte, head_len[i] = header line length incl.
 * the marker, seq_len[i] = bases.  Any output pointer may be NULL. */
int bsk_index_build(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, void* stream,
                    uint64_t* n_records);
int bsk_index_copy(bsk_ctx* ctx, uint64_t* starts, uint32_t* head_len, uint32_t* seq_len, uint32_t* aux, size_t cap);

/* ---- SeqTransform (bigseqkit-lib/seq.go:28-269) --------------------------- */
int bsk_seq_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                bsk_out* out);

/* ---- Grep (bigseqkit-lib/grep.go:24-549; exact patterns: by ID, by name -n, by
 * sequence -s on both strands, -i, -v, -R region, --circular, -C count).
 * With Count set the single element is the decimal count (GrepReduceCount sums
 * them across partitions, grep.go:598-611); bsk_grep_last_count returns it as a
 * number for a 1-word all-reduce. */
int bsk_grep_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                 bsk_out* out);
int bsk_grep_last_count(const bsk_ctx* ctx, uint64_t* count);
/* grep -r: the regular expression compiler of the HIP path (Go regexp / RE2 syntax subset -> position automaton,
 * bigseqkit_amd/csrc/regex_nfa.hpp) run on the HOST against one target: *matched = re.Match(text).  Needs no device;
 * lets callers (and the CPU tests) check an expression before creating a Grep context. */
int bsk_regex_match(const char* expr, const uint8_t* text, size_t n, int* matched);

/* ---- SubseqTransform by region (bigseqkit-lib/subseq.go:22-225, 314-317) -- */
int bsk_subseq_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                   bsk_out* out);

/* ---- Locate (bigseqkit-lib/locate.go:19-772; exact patterns on both strands, -i, -P,
 * --circular, -G non-greedy, TSV / -M / --gtf / --bed rows).  pid == 0 emits the header
 * row first (locate.go:198-204), exactly as MapPartitionsWithIndex does. */
int bsk_locate_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                   bsk_out* out);

/* ---- Translate (bigseqkit-lib/translate.go:21-145): one element per (record, frame),
 * ">Name" or ">ID_frame=N Desc" + the protein wrapped at Config.LineWidth. */
int bsk_translate_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                      bsk_out* out);

/* ---- Concat (bigseqkit/concat.go:41-90; ConcatPrepare + Union + GroupByKey + ConcatJoin, bigseqkit-lib/concat.go): for
 * every ID of both files, each record of file 1 joined with each record of file 2 of that ID: header = the ID, sequence
 * and quality = A followed by B.  With Full the records of IDs present in one file only are kept unchanged.  `shard` holds
 * file 1 followed by file 2, n_first = bytes of file 1 (ending in a newline). */
int bsk_concat_run(bsk_ctx* ctx, const void* shard, size_t n, size_t n_first, int on_device, int format, void* stream,
                   bsk_out* out);

/* ---- Common (bigseqkit/common.go:56-109; CommonPrepare + Union + GroupByKey + CommonJoin, bigseqkit-lib/common.go): the
 * records of the FIRST file whose ID (full name with ByName, sequence with BySeq; IgnoreCase) occurs in every file --
 * one record per key, file order.  `shard` holds the n_files files back to back (every file ends in a newline),
 * file_ends[f] = offset one past file f (host array, file_ends[n_files - 1] == n). */
int bsk_common_run(bsk_ctx* ctx, const void* shard, size_t n, const uint64_t* file_ends, uint32_t n_files, int on_device,
                   int format, void* stream, bsk_out* out);

/* ---- Pair (bigseqkit/pair.go:34-100; PairPrepare + Union + GroupByKey + Pair, bigseqkit-lib/pair.go:37-121): match up
 * the reads of two files by ID.  `shard` holds file 1 followed by file 2, n_first = bytes of file 1 (ending in a
 * newline).  outs[0], outs[1]: the paired records of file 1 / file 2, line for line in the same pair order (= PairIndex
 * 0 / 1); outs[2], outs[3]: the records without a mate (= UnpairedId "1" / "2"), filled with SaveUnpaired only. */
int bsk_pair_run(bsk_ctx* ctx, const void* shard, size_t n, size_t n_first, int on_device, int format, void* stream,
                 bsk_out outs[4]);

/* ---- Faidx index rows (FaidxOffset + Faidx, bigseqkit/faidx.go:61-95, bigseqkit-lib/faidx.go:29-229): one row per
 * record, "<ID>\t<length>\t<offset>\t<linebases>\t<linewidth>[\t<qualoffset>]" (the .fai columns; FullHead prints the
 * whole header as the name).  base_offset = file offset of the shard's first byte (what the FaidxOffset pass
 * accumulates per partition).  Records whose sequence lines do not have the .fai shape fail with the reference's
 * "different line length in sequence: <ID>" error. */
int bsk_faidx_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, uint64_t base_offset,
                  void* stream, bsk_out* out);

/* Region queries of the same context (FaidxQuery, bigseqkit-lib/faidx.go:231-432): with Regions / RegionFile in the
 * options ("id", "id:b-e", "id:b", "id:b-", "id:-e"; negative positions count from the end; b > e = reverse complement),
 * every record whose ID has a query comes back as FASTA: ">ID" or ">ID:b-e" and the region.  With UseRegexp the
 * queries are regular expressions on the ID and the hits come back whole. */
int bsk_faidx_query_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                        bsk_out* out);

/* ---- Sort (bigseqkit/sort.go:91-147; SortParseInputString / SortParseInputInt + SortByKey, bigseqkit-lib/sort.go):
 * by ID (default), full name (ByName), sequence prefix (BySeq, SeqPrefixLength), length (ByLength) or non-gap bases
 * (ByBases); IgnoreCase, Reverse.  Records with equal keys keep file order.  Global: ONE call sees the whole input of
 * a rank.  InNaturalOrder: natural order of IDs / names (digit runs compare as numbers). */
int bsk_sort_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                 bsk_out* out);

/* ---- Rename (RenamePrepare + GroupByKey + Rename, bigseqkit/rename.go:34-60, bigseqkit-lib/rename.go:39-131):
 * the k-th further record of an ID (of a whole name with ByName) becomes "<ID>_<k> <Desc>".  Global like rmdup:
 * ONE call must see the whole input of a rank; output in file order. */
int bsk_rename_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                   bsk_out* out);

/* ---- Fq2Fa (bigseqkit-lib/fq2fa.go:16-59): every record as FASTA, the sequence on one line (Format(0)). */
int bsk_fq2fa_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                  bsk_out* out);

/* ---- Range / Head (driver: bigseqkit/range.go:36-103, head.go:34-44; executor: RangePrepare + RangeFilter,
 * bigseqkit-lib/range.go:14-43).  A context is created from RangeOptions {"Range": "a:b"} or HeadOptions {"N": n}; the
 * records whose index in the WHOLE input lies in the range come back unchanged.  first_record = index of the shard's
 * first record (what MapWithIndex hands to RangePrepare.Call).  Ranges with a position below -1 need input.Count()
 * (range.go:69-80): bsk_range_needs_count says so, bsk_range_set_count supplies it before the first run. */
int bsk_range_needs_count(const bsk_ctx* ctx, int* needs);
int bsk_range_set_count(bsk_ctx* ctx, uint64_t n_records);
/* the resolved 0-based half-open record range [start, end) (after bsk_range_set_count when that was needed) */
int bsk_range_bounds(const bsk_ctx* ctx, int64_t* start, int64_t* end);
int bsk_range_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, uint64_t first_record,
                  void* stream, bsk_out* out);

/* ---- Duplicate (bigseqkit/duplicate.go:31-43, bigseqkit-lib/duplicate.go:13-30): every record Times times, copies
 * adjacent. */
int bsk_duplicate_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                      bsk_out* out);

/* ---- RmDup (RmDupPrepare + GroupByKey + RmDupCheck, bigseqkit-lib/rmdup.go:23-242):
 * duplicates are global, so ONE call must see the whole input of a rank; the first
 * record of every subject (file order) survives. */
int bsk_rmdup_run(bsk_ctx* ctx, const void* shard, size_t n, int on_device, int format, int64_t pid, void* stream,
                  bsk_out* out);
/* RmDupCheck.After() (bigseqkit-lib/rmdup.go:244-279): writes what the runs of this context accumulated for
 * -d (--dup-seqs-file: text of the removed records) and -D (--dup-num-file: "<n>\t<id>, <id>, ...") to
 * <file>/<device index>, nothing when no record was removed.  bsk_destroy() calls it if the caller did not. */
int bsk_rmdup_finish(bsk_ctx* ctx);

/* ---- rmdup across ranks (one process per GPU).  Duplicates are global, so the reference shuffles whole records
 * with GroupByKey (bigseqkit/rmdup.go:97).  Here a record travels as a 24-byte tuple
 *     (XXH64 key, second XXH64 with another seed, global record index)
 * to its owner rank = key % world; the owner keeps the lowest global index of every key (first in file order,
 * PARITY.md Q10) and answers one keep byte per tuple; equal keys with different second keys raise
 * BSK_ERR_UNSUPPORTED instead of a guess.  The caller runs the collectives between the phases
 * (bigseqkit_amd/dist.py: all_gather of counts, all_to_all_single of tuples, all_to_all_single of keep bytes):
 *   keys    : record table + both keys of the HBM-resident shard            -> *n_records
 *   pack    : tuples bucketed by owner into d_send (u64[3 * n_records]), counts[world] on the host;
 *             base_index = number of records on lower ranks
 *   resolve : owner side, d_tuples = the m tuples received (u64[3 * m])       -> d_keep (u8[m], 1 = first of its key)
 *   emit    : d_reply (u8[n_records]) = keep bytes in the order of d_send    -> survivors of this shard, file order
 * keys ... emit must run on the same context without another run in between (it holds the record table). */
int bsk_rmdup_dist_keys(bsk_ctx* ctx, const void* d_shard, size_t n, int format, void* stream, uint64_t* n_records);
int bsk_rmdup_dist_pack(bsk_ctx* ctx, uint64_t base_index, int world, void* d_send, uint64_t* counts, void* stream);
int bsk_rmdup_dist_resolve(bsk_ctx* ctx, const void* d_tuples, uint64_t m, void* d_keep, void* stream);
int bsk_rmdup_dist_emit(bsk_ctx* ctx, const void* d_send, const void* d_reply, uint64_t base_index, void* stream,
                        bsk_out* out);
/* The same two phases with the survivor's identity (round 5): _resolve_ex also writes d_survivor (u64[m]: the global index of
 * the record that survives for every tuple's subject); routed back like the keep bytes (d_survivor_reply, u64[n_records] in
 * the order of d_send) it lets _emit_ex compare the BYTES of every duplicate whose survivor lives in the same shard with that
 * survivor's (RmDupCheck's own test, bigseqkit-lib/rmdup.go:193-199; `-s` on FASTQ) -- without the cross-rank check below a
 * difference fails the call and pairs that cross ranks rest on the two keys; with it (bsk_rmdup_dist_x*, which
 * bsk_rmdup_dist_run runs) every pair is compared and _emit_ex only emits.  Either pointer NULL: the plain phases.
 * *local_pairs_verified (may be NULL): how many pairs inside the shard were compared. */
int bsk_rmdup_dist_resolve_ex(bsk_ctx* ctx, const void* d_tuples, uint64_t m, void* d_keep, void* d_survivor, void* stream);
int bsk_rmdup_dist_emit_ex(bsk_ctx* ctx, const void* d_send, const void* d_reply, const void* d_survivor_reply, uint64_t base_index,
                           void* stream, bsk_out* out, uint64_t* local_pairs_verified);

/* ---- RmDupCheck's text comparison across ranks (round 6; bigseqkit-lib/rmdup.go:193-211 compares the subject text of every
 * member of a hash group -- the reference has the text there because GroupByKey, bigseqkit/rmdup.go:97, shuffles whole
 * records).  After the replies of _resolve_ex are back, every duplicate whose survivor lives on ANOTHER rank sends its
 * subject to that rank: a 24-byte request {survivor's global index, offset of the text in the destination's segment,
 * length | local record << 32} and the bytes.  Between the phases the caller runs three all-to-all exchanges:
 *   _xpack    : rank_base[r] = global index of rank r's first record (world + 1 entries); req_counts[r] / byte_counts[r] =
 *               requests / text bytes for rank r; *d_requests / *d_text = the context's send buffers, grouped by destination
 *               [requests, 24 B each, and text to their destinations]
 *   _xcompare : survivor side.  req_from[r] / bytes_from[r] = what arrived from rank r (buffers in rank order);
 *               d_verdict[j] = 1 equal, 0 differs (-i: case-folded)        [verdicts back, the same routes reversed]
 *   _xapply   : d_verdict_back in the order of *d_requests.  Also compares the pairs INSIDE the shard.  *n_flagged = records
 *               of this shard whose text differs from their survivor's (two subjects under one pair of keys); *pairs_compared
 *               = local + cross-rank pairs.  bsk_rmdup_dist_emit[_ex] after it does not compare again.
 * Only when the SUM of n_flagged over the ranks is not zero (about N^2 / 2^129; the tests mask the keys): every rank hands
 * its list (_flagged_get: entries {u64 global index, u64 length, text padded to 8}; call with buf NULL for the size) to
 * every other, and _flagged_settle(concatenation of all ranks' lists, any order) regroups them by TEXT -- the lowest global
 * index of every text survives, as RmDupCheck's map keyed by the subject decides.  bsk_rmdup_dist_run does all of this. */
int bsk_rmdup_dist_xpack(bsk_ctx* ctx, const void* d_send, const void* d_reply, const void* d_survivor_reply, uint64_t base_index,
                         const uint64_t* rank_base, int world, uint64_t* req_counts, uint64_t* byte_counts, void** d_requests, void** d_text,
                         void* stream);
int bsk_rmdup_dist_xcompare(bsk_ctx* ctx, const void* d_requests_in, const uint64_t* req_from, const void* d_text_in,
                            const uint64_t* bytes_from, int world, void* d_verdict, void* stream);
int bsk_rmdup_dist_xapply(bsk_ctx* ctx, const void* d_verdict_back, uint64_t* n_flagged, uint64_t* pairs_compared, void* stream);
int bsk_rmdup_dist_flagged_get(bsk_ctx* ctx, void* buf, size_t cap, size_t* need);
int bsk_rmdup_dist_flagged_settle(bsk_ctx* ctx, const void* all_lists, size_t n_bytes);
/* what the last exchange of this context compared: pairs inside the shard, pairs whose survivor lives on another rank (their
 * text went there), and the records of this shard that were flagged (any pointer may be NULL) */
int bsk_rmdup_dist_stats(const bsk_ctx* ctx, uint64_t* local_pairs, uint64_t* cross_pairs, uint64_t* flagged);

/* ---- collectives behind the C ABI (round 5): RCCL over xGMI, no Python, no torch -----------------------------------------
 * In the reference the driver gets Reduce and GroupByKey from IgnisHPC, in the same binary (bigseqkit/stats.go:91,
 * grep.go:175, rmdup.go:97; the executors from ignisDriver, bigseqkit-cli/helper.go:87-132).  A bsk_comm is one rank's
 * handle on a group of `world` ranks, each with its own GPU and its own bsk_ctx:
 *   one rank per PROCESS : rank 0 calls bsk_comm_unique_id, the host hands the 128 bytes to the other ranks (a file, a
 *                          socket, the launcher's environment), every rank calls bsk_comm_init_rank   (ncclCommInitRank)
 *   all ranks in ONE process, a thread each : bsk_comm_init_all(ndev, devices, comms)                 (ncclCommInitAll);
 *                          devices that repeat (ranks sharing a GPU: RCCL refuses that) get the "local" backend -- the
 *                          same calls through host memory between the threads (tests on a one-GPU box)
 * librccl is loaded at the first of these calls, not with libbsk.so.  Every collective must be entered by ALL ranks.
 * Errors: the code, and bsk_comm_error(comm) (bsk_comm_error(NULL): the calling thread's last failure without a comm). */
typedef struct bsk_comm bsk_comm;
#define BSK_COMM_ID_BYTES 128
int bsk_comm_unique_id(void* id128);
int bsk_comm_init_rank(int world, int rank, const void* id128, int device, bsk_comm** out);
int bsk_comm_init_all(int ndev, const int* devices, bsk_comm** out /* [ndev] */);
int bsk_comm_destroy(bsk_comm* comm);
int bsk_comm_info(const bsk_comm* comm, int* world, int* rank, int* device, int* over_rccl);
const char* bsk_comm_error(const bsk_comm* comm);
int bsk_comm_barrier(bsk_comm* comm, void* stream);
/* in place on `count` device words; op: 0 sum, 1 max, 2 min */
int bsk_comm_allreduce_u64(bsk_comm* comm, void* d_buf, size_t count, int op, void* stream);
/* one word of every rank -> out[world] on the host (record counts, shard and part sizes: Range / Head, bigseqkit/range.go:69-103;
 * FaidxOffset, faidx.go:69-80; the offsets of FileStore's ordered single file, bigseqkit-lib/helper.go:399-429); synchronises */
int bsk_comm_allgather_u64(bsk_comm* comm, uint64_t value, uint64_t* out, void* stream);
/* GrepReduceCount (bigseqkit-lib/grep.go:598-611 through Reduce, bigseqkit/grep.go:175): *inout becomes the sum over ranks */
int bsk_count_allreduce(bsk_comm* comm, uint64_t* inout, void* stream);
/* StatsReduce (bigseqkit-lib/stats.go:128-137 through Reduce, bigseqkit/stats.go:91) + the driver's collect: ONE sum
 * all-reduce of the stats vector (d_vec, or the context's own when NULL) and bsk_stats_collect; only when the reduced vector
 * counts sequence lengths beyond the dense histogram do the ranks exchange their overflow lists and collect again.  Every
 * rank receives the whole map. */
int bsk_stats_collect_reduced(bsk_ctx* ctx, bsk_comm* comm, void* d_vec, void* stream, int64_t* keys, int64_t* vals, size_t cap,
                              size_t* n_out);
/* RmDup over the shards of all ranks in ONE call (GroupByKey, bigseqkit/rmdup.go:97): the four phases above with their
 * collectives in between -- all-gather of the record counts, tuples to their owners by grouped ncclSend / ncclRecv (in
 * rounds of at most 512 MiB per message), one keep byte per tuple back the same way.  `out`: the survivors of THIS rank's
 * shard in file order; the concatenation over the ranks equals the single-GPU output. */
int bsk_rmdup_dist_run(bsk_ctx* ctx, bsk_comm* comm, const void* d_shard, size_t n, int format, void* stream, bsk_out* out);

/* host-side self-test of the position-reporting regular-expression matcher (custom --id-regexp, locate -r):
 * leftmost-first match at or after `from`; caps4 = {match start, match end, group-1 start, group-1 end} (0xFFFFFFFF:
 * group 1 did not take part).  1 = match, 0 = none, -1 = expression rejected (bsk_global_error) */
int bsk_selftest_regex_find(const char* expr, const uint8_t* text, size_t n, size_t from, uint32_t* caps4, uint32_t* ngroups);

/* ---- FileStore / StoreFASTXN  (bigseqkit-lib/helper.go:378-460 NewFileStore; bigseqkit/helper.go:186-195 StoreFASTX[N]) ----
 * merge != 0: ONE file at `path`, the parts (partitions) in the order of their numbers whatever the order of the calls --
 * the reference passes an MPI token from executor to executor (helper.go:418-436), here a part that comes before its
 * turn waits in host memory.  merge == 0: directory `path` with one file part%05d per partition (SaveAsTextFile).
 * Parts must be numbered 0, 1, 2, ... without gaps for the single file to grow while the parts arrive. */
typedef struct bsk_store bsk_store;
int bsk_store_open(const char* path, int merge, bsk_store** out);
const char* bsk_store_error(const bsk_store* s);
/* the output of an operator call (device memory of `ctx`), copied to the host in 64 MiB pieces through two pinned
 * buffers -- the copy of a piece runs while the piece before it is written */
int bsk_store_put(bsk_store* s, bsk_ctx* ctx, uint64_t part, const bsk_out* out);
int bsk_store_put_host(bsk_store* s, uint64_t part, const void* data, size_t n);
int bsk_store_close(bsk_store* s, uint64_t* total_bytes); /* also frees s */
/* Call(partition) + FileStore for a partition in HOST memory (best: pinned, bsk_host_alloc): record-aligned chunks of
 * 256 MiB (BSK_STAGE_BYTES) through two device buffers -- H2D of chunk i+1, the kernels of chunk i, and D2H + write of
 * the output of chunk i-1 overlap (three streams and a writer thread).  seq, grep, locate, subseq, translate, fq2fa,
 * duplicate are chunked; rmdup, rename, sort, grep -C / --delete-matched see the whole partition (their output is still
 * drained in pieces).  out_bytes / out_records may be NULL. */
int bsk_run_to_store(bsk_ctx* ctx, const void* host_shard, size_t n, int format, int64_t pid, bsk_store* s, uint64_t part,
                     uint64_t* out_bytes, uint64_t* out_records);

/* ---- synthetic inputs (bench + tests only) --------
 * Deterministic, counter-based: byte k of record i depends on (seed, i, k)
 * only, so any shard can be produced on the host or directly in HBM. */
#define BSK_SYNTH_FASTQ150 0 /* 317 B/record                          */
#define BSK_SYNTH_FASTA1K 1  /* 1027 B/record, 60-column lines        */
#define BSK_SYNTH_FASTA5K_CDS 2
#define BSK_SYNTH_FASTA5K_VAR 3 /* the same CDS records with unpadded numbers in the header and 1 % of them 3 bases shorter /
                                 * longer: records of several sizes (bsk_synth_record_bytes = 0, bsk_synth_offset) */
#define BSK_SYNTH_FLAG_MOTIF 1u /* C3: plant ACGTTGCAAGCT / its revcomp */
#define BSK_SYNTH_FLAG_DUPS 2u  /* C5: 20 % sequence duplicates         */
size_t bsk_synth_record_bytes(int kind);
/* file offset of record `record` (== bytes of the records below it); n bytes from there: bsk_synth_host / _device */
uint64_t bsk_synth_offset(int kind, uint64_t record);
/* fill dst[0..n) with the bytes [first_record*record_bytes, ...+n) of the
 * synthetic file; n need not be a whole number of records */
int bsk_synth_host(int kind, uint64_t seed, unsigned flags, uint64_t first_record, uint8_t* dst, size_t n);
int bsk_synth_device(int kind, uint64_t seed, unsigned flags, uint64_t first_record, void* d_dst, size_t n, int device,
                     void* stream);

/* ---- timing helper for bench.py: HIP events on the stream the kernels use */
int bsk_event_create(void** ev);
int bsk_event_record(void* ev, void* stream);
int bsk_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */
int bsk_event_destroy(void* ev);
/* accumulated device time of the dominant kernel of the last run(s), measured
 * with HIP events around each launch when profiling is enabled */
int bsk_profile_enable(bsk_ctx* ctx, int on);
int bsk_profile_read(bsk_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches);
int bsk_profile_reset(bsk_ctx* ctx);
/* all timed stages of the context: "name=total_ms/launches;..." (NUL-terminated) */
int bsk_profile_dump(bsk_ctx* ctx, char* buf, size_t cap);

/* ---- device self tests used by tests/ (-m gpu) ---------------------------- */
int bsk_selftest_scan(int use_dpp, const uint32_t* in64, uint32_t* out64);
/* the two 64-bit keys (XXH64 seed 0, and the second key of csrc/hash_dev.hpp) of every record of the shard of the last
 * bsk_rmdup_dist_keys call, in record order.  k2 == NULL: k1 alone -- after a bsk_rmdup_run (-s on FASTQ, keys verified by
 * bytes) that is the chain-free GROUPING key of csrc/hash_dev.hpp, which the tests restate in Python */
int bsk_selftest_rmdup_keys(bsk_ctx* ctx, uint64_t* k1, uint64_t* k2, size_t cap, size_t* n_out);
/* streaming read of d_buf[0..n) with k_stats' tile/queue pattern and no per-byte work:
 * the read ceiling of that pattern and the FETCH_SIZE calibration run */
int bsk_selftest_stream_read(const void* d_buf, size_t n, int reps, int blocks_per_cu, float* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* BSK_H */
