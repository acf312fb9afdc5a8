// bsk_cgo.go -- the cgo binding a BigSeqKit maintainer adds to the driver package `bigseqkit` so that the seven hot-path
// commands run on libbsk.so (MI355X) instead of IgnisHPC executors.
//
// UNVERIFIED: no Go toolchain exists in the environment this file was written in (`go version`: not found), so it has never
// been compiled.  The same call sequences are exercised through ctypes by bigseqkit_amd/api.py and the `-m gpu` tests; the
// C declarations are include/bsk.h.  Citations are file:line in the reference tree (citiususc/BigSeqKit).
//
// What it replaces, function by function (the option structs, their builders and setDefaults() stay as they are):
//
//   Stats / StatsString   bigseqkit/stats.go:75-288     MapPartitions(Stats) + Reduce(StatsReduce) + driver arithmetic
//   Seq                   bigseqkit/seq.go:157-170      MapPartitions(SeqTransform)
//   Grep / GrepCount      bigseqkit/grep.go:121-180     MapPartitionsWithIndex(Grep) [+ Reduce(GrepReduceCount)]
//   Locate                bigseqkit/locate.go:122-134   MapPartitionsWithIndex(Locate)
//   Subseq                bigseqkit/subseq.go:86-100    MapPartitions(SubseqTransform)
//   Translate             bigseqkit/translate.go:87-100 MapPartitions(Translate)
//   RmDup                 bigseqkit/rmdup.go:70-108     MapPartitions(RmDupPrepare) + GroupByKey + Flatmap(RmDupCheck)
//   ReadFASTA/Q[N]        bigseqkit/helper.go:148-178   PlainFile(path, delim) + ReadFixer
//   StoreFASTX[N]         bigseqkit/helper.go:180-195   SaveAsTextFile / FileStore
//
// Threads (include/bsk.h "THREADS"): the reference calls Call() of one operator struct from Threads() goroutines
// (bigseqkit-lib/helper.go:413-416).  A bsk_ctx is single-caller, so mapPartitions below creates one context PER WORKER
// GOROUTINE from the same options JSON; the FileStore (bsk_store) is the one object the workers share.
package bigseqkit

/*
#cgo CFLAGS:  -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../bigseqkit_amd/lib -lbsk -Wl,-rpath,${SRCDIR}/../../bigseqkit_amd/lib
#include <stdlib.h>
#include <stdint.h>
#include "bsk.h"

// cgo cannot take the address of a Go function-typed field; thin wrappers keep one signature for the record operators
typedef int (*bsk_run_fn)(bsk_ctx*, const void*, size_t, int, int, int64_t, void*, bsk_out*);
static int bsk_call_run(bsk_run_fn f, bsk_ctx* c, const void* p, size_t n, int on_dev, int fmt, int64_t pid, bsk_out* o) {
	return f(c, p, n, on_dev, fmt, pid, NULL, o);
}
static bsk_run_fn bsk_fn_seq(void)       { return bsk_seq_run; }
static bsk_run_fn bsk_fn_grep(void)      { return bsk_grep_run; }
static bsk_run_fn bsk_fn_locate(void)    { return bsk_locate_run; }
static bsk_run_fn bsk_fn_subseq(void)    { return bsk_subseq_run; }
static bsk_run_fn bsk_fn_translate(void) { return bsk_translate_run; }
static bsk_run_fn bsk_fn_rmdup(void)     { return bsk_rmdup_run; }
*/
import "C"

import (
	"errors"
	"os"
	"runtime"
	"sync"
	"unsafe"
)

const (
	FormatFASTA = C.BSK_FORMAT_FASTA
	FormatFASTQ = C.BSK_FORMAT_FASTQ
)

// Shard is one partition: file bytes that begin on a record (PlainFile + ReadFixer, bigseqkit/helper.go:140-178;
// bigseqkit-lib/helper.go:41-66).  Data may be a slice of an mmap of the input file: nothing is copied to cut it.
type Shard struct {
	Data []byte
}

// SeqFrame stands where the reference has *api.IDataFrame[string]: the partitions of one input and its format.
type SeqFrame struct {
	Format int
	Shards []Shard
	Device int // GPU of this process (one process per GPU; see DESIGN.md section 5)
}

// ---- ReadFASTA / ReadFASTQ [N] --------------------------------------------------------------------------------------
// bigseqkit/helper.go:148-178.  minPartitions record-aligned shards; the cut points come from bsk_find_record_start (the
// rule of IgnisHPC's delimiter "\n@!\n+" for FASTQ, '>' at a line start for FASTA).
func readFASTX(data []byte, format, minPartitions, device int) (*SeqFrame, error) {
	if minPartitions < 1 {
		minPartitions = 1
	}
	n := len(data)
	cuts := []int{0}
	for k := 1; k < minPartitions && n > 0; k++ {
		var out C.size_t
		if rc := C.bsk_find_record_start((*C.uint8_t)(unsafe.Pointer(&data[0])), C.size_t(n), C.size_t(n*k/minPartitions),
			C.int(format), &out); rc != C.BSK_OK {
			return nil, errors.New(C.GoString(C.bsk_global_error()))
		}
		if int(out) > cuts[len(cuts)-1] {
			cuts = append(cuts, int(out))
		}
	}
	cuts = append(cuts, n)
	// (a cut that finds no record start behind it collapses onto the one before: the frame then has FEWER shards than
	// minPartitions, numbered 0, 1, ... without gaps -- the partition ids the operators see are the indices of this slice)
	f := &SeqFrame{Format: format, Device: device}
	for k := 0; k+1 < len(cuts); k++ {
		if cuts[k+1] > cuts[k] {
			f.Shards = append(f.Shards, Shard{data[cuts[k]:cuts[k+1]]})
		}
	}
	return f, nil
}
func ReadFASTA(data []byte, device int) (*SeqFrame, error) { return readFASTX(data, FormatFASTA, 1, device) }
func ReadFASTQ(data []byte, device int) (*SeqFrame, error) { return readFASTX(data, FormatFASTQ, 1, device) }
func ReadFASTAN(data []byte, minPartitions, device int) (*SeqFrame, error) {
	return readFASTX(data, FormatFASTA, minPartitions, device)
}
func ReadFASTQN(data []byte, minPartitions, device int) (*SeqFrame, error) {
	return readFASTX(data, FormatFASTQ, minPartitions, device)
}

// ---- operator lifecycle: plugin.Lookup("New"+name) + Before(ctx) ... After(ctx) ---------------------------------------
type bskOp struct{ ctx *C.bsk_ctx }

func newBskOp(name, optsJSON string, device int) (*bskOp, error) {
	cn, cj := C.CString(name), C.CString(optsJSON)
	defer C.free(unsafe.Pointer(cn))
	defer C.free(unsafe.Pointer(cj))
	var ctx *C.bsk_ctx
	if rc := C.bsk_create(cn, cj, C.int(device), &ctx); rc != C.BSK_OK {
		return nil, errors.New(C.GoString(C.bsk_global_error())) // the reference's Before() message
	}
	return &bskOp{ctx}, nil
}
func (o *bskOp) Close()     { C.bsk_destroy(o.ctx) }
func (o *bskOp) err() error { return errors.New(C.GoString(C.bsk_last_error(o.ctx))) }

// Result of a record command: the text of every partition in partition order (an element per record + '\n', what
// FileStore / SaveAsTextFile would write), or -- when a store is given -- already in the output file(s).
type Result struct {
	Parts   [][]byte
	Records uint64
	Bytes   uint64
}

// FileStore: bigseqkit-lib/helper.go:378-460 (merge) / SaveAsTextFile (directory of parts).
type FileStore struct{ s *C.bsk_store }

func OpenStore(path string, merge bool) (*FileStore, error) {
	cp := C.CString(path)
	defer C.free(unsafe.Pointer(cp))
	var s *C.bsk_store
	m := 0
	if merge {
		m = 1
	}
	if rc := C.bsk_store_open(cp, C.int(m), &s); rc != C.BSK_OK {
		return nil, errors.New(C.GoString(C.bsk_global_error()))
	}
	return &FileStore{s}, nil
}
func (f *FileStore) Close() (uint64, error) {
	var total C.uint64_t
	if rc := C.bsk_store_close(f.s, &total); rc != C.BSK_OK {
		return uint64(total), errors.New("bsk_store_close failed")
	}
	return uint64(total), nil
}

// mapPartitions == api.MapPartitions[WithIndex](input, libSource(name)+opts): Call() per partition, `workers` goroutines,
// ONE CONTEXT EACH.  Host shards go through bsk_run_to_store when a store is given (chunked H2D || kernels || D2H + write),
// else through bsk_<op>_run(on_device = 0) + bsk_out_to_host.  No Go pointer is retained by C past a call.
func mapPartitions(name string, fn C.bsk_run_fn, optsJSON string, in *SeqFrame, store *FileStore, workers int) (*Result, error) {
	if workers < 1 {
		workers = 1
	}
	if workers > len(in.Shards) {
		workers = len(in.Shards)
	}
	res := &Result{Parts: make([][]byte, len(in.Shards))}
	var mu sync.Mutex
	var firstErr error
	next := 0
	var wg sync.WaitGroup
	for w := 0; w < workers; w++ {
		wg.Add(1)
		go func() {
			defer wg.Done()
			runtime.LockOSThread() // the HIP runtime's current device is per OS thread
			defer runtime.UnlockOSThread()
			op, err := newBskOp(name, optsJSON, in.Device)
			if err != nil {
				mu.Lock()
				if firstErr == nil {
					firstErr = err
				}
				mu.Unlock()
				return
			}
			defer op.Close()
			// a result is copied to Go memory right after its run (bsk_out_to_host), while the staged shard is still the
			// context's: the operators that can leave their text as ordered slices do (include/bsk.h bsk_out.d_seg_*) --
			// the reference's []string elements share the partition's bytes just so
			cOut, cSlices := C.CString("out"), C.CString("slices")
			C.bsk_ctx_set(op.ctx, cOut, cSlices)
			C.free(unsafe.Pointer(cOut))
			C.free(unsafe.Pointer(cSlices))
			for {
				mu.Lock()
				pid := next
				next++
				failed := firstErr != nil
				mu.Unlock()
				if failed || pid >= len(in.Shards) {
					return
				}
				d := in.Shards[pid].Data
				var ptr unsafe.Pointer
				if len(d) > 0 {
					ptr = unsafe.Pointer(&d[0])
				}
				var e error
				if store != nil {
					var nb, nr C.uint64_t
					if rc := C.bsk_run_to_store(op.ctx, ptr, C.size_t(len(d)), C.int(in.Format), C.int64_t(pid), store.s,
						C.uint64_t(pid), &nb, &nr); rc != C.BSK_OK {
						e = op.err()
					}
					mu.Lock()
					res.Bytes += uint64(nb)
					res.Records += uint64(nr)
					mu.Unlock()
				} else {
					var out C.bsk_out
					if rc := C.bsk_call_run(fn, op.ctx, ptr, C.size_t(len(d)), 0, C.int(in.Format), C.int64_t(pid), &out); rc != C.BSK_OK {
						e = op.err()
					} else {
						buf := make([]byte, int(out.len))
						if out.len > 0 {
							if rc := C.bsk_out_to_host(op.ctx, &out, unsafe.Pointer(&buf[0]), out.len); rc != C.BSK_OK {
								e = op.err()
							}
						}
						mu.Lock()
						res.Parts[pid] = buf
						res.Bytes += uint64(out.len)
						res.Records += uint64(out.records)
						mu.Unlock()
					}
				}
				if e != nil {
					mu.Lock()
					if firstErr == nil {
						firstErr = e
					}
					mu.Unlock()
					return
				}
			}
		}()
	}
	wg.Wait()
	if firstErr != nil {
		return nil, firstErr
	}
	return res, nil
}

// ---- Stats (bigseqkit/stats.go:75-166) ---------------------------------------------------------------------------------
// Stats.Call per partition accumulates into the context's device vector; StatsReduce is implied (one context sums its
// partitions; several contexts merge with bsk_stats_merge); Stats()'s driver arithmetic is bsk_stats_finalize.
func Stats(name, format string, input *SeqFrame, o *SeqKitStatsOptions) (*StatInfo, error) {
	o.setDefaults()
	op, err := newBskOp("Stats", OptionsToString(o), input.Device)
	if err != nil {
		return nil, err
	}
	defer op.Close()
	for pid, s := range input.Shards {
		if len(s.Data) == 0 {
			continue
		}
		if rc := C.bsk_stats_run(op.ctx, unsafe.Pointer(&s.Data[0]), C.size_t(len(s.Data)), 0, C.int(input.Format),
			C.int64_t(pid), nil, nil); rc != C.BSK_OK {
			return nil, op.err()
		}
	}
	capN := 1 << 17
	for {
		keys := make([]C.int64_t, capN)
		vals := make([]C.int64_t, capN)
		var n C.size_t
		rc := C.bsk_stats_collect(op.ctx, nil, &keys[0], &vals[0], C.size_t(capN), &n)
		if rc == C.BSK_ERR_CAPACITY && int(n) > capN {
			capN = int(n)
			continue
		}
		if rc != C.BSK_OK {
			return nil, op.err() // e.g. "unmatched length of sequence and quality"
		}
		var info C.bsk_statinfo
		if rc := C.bsk_stats_finalize(op.ctx, &keys[0], &vals[0], n, &info); rc != C.BSK_OK {
			return nil, op.err()
		}
		return &StatInfo{name, format, C.GoString(&info._type[0]),
			uint64(info.num), uint64(info.len_sum), uint64(info.gap_sum), uint64(info.len_min),
			float64(info.len_avg), uint64(info.len_max), uint64(info.n50), int(info.l50),
			float64(info.q1), float64(info.q2), float64(info.q3), float64(info.q20), float64(info.q30)}, nil
	}
}

// StatsString (bigseqkit/stats.go:168-288) keeps its body: it only formats a *StatInfo.

// ---- the record commands ---------------------------------------------------------------------------------------------
func workersFor(in *SeqFrame) int {
	w := runtime.NumCPU() / 8 // a handful of contexts keeps one GPU busy; more only queue behind each other
	if w < 1 {
		w = 1
	}
	if w > 8 {
		w = 8
	}
	return w
}

// Seq: bigseqkit/seq.go:157-170
func Seq(input *SeqFrame, o *SeqKitSeqOptions) (*Result, error) {
	o.setDefaults()
	return mapPartitions("SeqTransform", C.bsk_fn_seq(), OptionsToString(o), input, nil, workersFor(input))
}

// Grep: bigseqkit/grep.go:121-159.  (Q3 of SURVEY section 9: the reference names "GrepPairMatched" here; the operator is Grep.)
// --delete-matched needs every pattern's FIRST record over the whole input (grep.go:144-156): one partition, one context.
func Grep(input *SeqFrame, o *SeqKitGrepOptions) (*Result, error) {
	o.setDefaults()
	f := false
	o.Count = &f
	in := input
	if o.DeleteMatched != nil && *o.DeleteMatched {
		in = joined(input)
	}
	return mapPartitions("Grep", C.bsk_fn_grep(), OptionsToString(o), in, nil, workersFor(in))
}

// GrepCount: bigseqkit/grep.go:161-180 -- Reduce(GrepReduceCount) is the sum of the partitions' record counts
func GrepCount(input *SeqFrame, o *SeqKitGrepOptions) (int64, error) {
	o.setDefaults()
	t := true
	o.Count = &t
	r, err := mapPartitions("Grep", C.bsk_fn_grep(), OptionsToString(o), input, nil, workersFor(input))
	if err != nil {
		return 0, err
	}
	var total int64
	for _, p := range r.Parts { // every partition answers one decimal line
		var v int64
		for _, ch := range p {
			if ch >= '0' && ch <= '9' {
				v = v*10 + int64(ch-'0')
			}
		}
		total += v
	}
	return total, nil
}

// Locate: bigseqkit/locate.go:122-134 (partition 0 carries the header row, bigseqkit-lib/locate.go:198-204)
func Locate(input *SeqFrame, o *SeqKitLocateOptions) (*Result, error) {
	o.setDefaults()
	return mapPartitions("Locate", C.bsk_fn_locate(), OptionsToString(o), input, nil, workersFor(input))
}

// Subseq: bigseqkit/subseq.go:86-100
func Subseq(input *SeqFrame, o *SeqKitSubseqOptions) (*Result, error) {
	o.setDefaults()
	return mapPartitions("SubseqTransform", C.bsk_fn_subseq(), OptionsToString(o), input, nil, workersFor(input))
}

// Translate: bigseqkit/translate.go:87-100
func Translate(input *SeqFrame, o *SeqKitTranslateOptions) (*Result, error) {
	o.setDefaults()
	return mapPartitions("Translate", C.bsk_fn_translate(), OptionsToString(o), input, nil, workersFor(input))
}

// RmDup: bigseqkit/rmdup.go:70-108.  Duplicates are global (GroupByKey): the partitions of this process are joined and
// seen by ONE context; across processes (one per GPU) the bsk_rmdup_dist_* phases exchange 24-byte tuples instead of
// records (INTEGRATION.md "Multi-GPU").  RmDupCheck.After (-d / -D files) runs inside bsk_destroy / bsk_rmdup_finish.
func RmDup(input *SeqFrame, o *SeqKitRmDupOptions) (*Result, error) {
	o.setDefaults()
	if o.BySeq != nil && *o.BySeq && o.ByName != nil && *o.ByName { // rmdup.go:79-81 (also checked by bsk_create)
		return nil, errors.New("only one/none of the flags -s (--by-seq) and -n (--by-name) is allowed")
	}
	return mapPartitions("RmDup", C.bsk_fn_rmdup(), OptionsToString(o), joined(input), nil, 1)
}

// joined: all partitions as one shard (Union + Repartition(1)); a missing final newline between partitions is added
func joined(in *SeqFrame) *SeqFrame {
	if len(in.Shards) <= 1 {
		return in
	}
	var all []byte
	for _, s := range in.Shards {
		all = append(all, s.Data...)
		if n := len(all); n > 0 && all[n-1] != '\n' {
			all = append(all, '\n')
		}
	}
	return &SeqFrame{Format: in.Format, Shards: []Shard{{all}}, Device: in.Device}
}

// ---- StoreFASTX / StoreFASTXN (bigseqkit/helper.go:180-195) -----------------------------------------------------------
// For results already in memory.  To stream a command straight into files pass a FileStore to mapPartitions instead
// (bsk_run_to_store: the partition never exists in Go memory).
func StoreFASTX(r *Result, path string) error {
	st, err := OpenStore(path, true)
	if err != nil {
		return err
	}
	for pid, p := range r.Parts {
		var ptr unsafe.Pointer
		if len(p) > 0 {
			ptr = unsafe.Pointer(&p[0])
		}
		if rc := C.bsk_store_put_host(st.s, C.uint64_t(pid), ptr, C.size_t(len(p))); rc != C.BSK_OK {
			msg := C.GoString(C.bsk_store_error(st.s))
			st.Close() // (also frees the store: an error path must not leak it)
			return errors.New(msg)
		}
	}
	_, err = st.Close()
	return err
}
func StoreFASTXN(r *Result, path string) error {
	if err := os.MkdirAll(path, 0o777); err != nil {
		return err
	}
	st, err := OpenStore(path, false)
	if err != nil {
		return err
	}
	for pid, p := range r.Parts {
		var ptr unsafe.Pointer
		if len(p) > 0 {
			ptr = unsafe.Pointer(&p[0])
		}
		if rc := C.bsk_store_put_host(st.s, C.uint64_t(pid), ptr, C.size_t(len(p))); rc != C.BSK_OK {
			msg := C.GoString(C.bsk_store_error(st.s))
			st.Close() // (also frees the store: an error path must not leak it)
			return errors.New(msg)
		}
	}
	_, err = st.Close()
	return err
}

// ---- several GPUs from Go: the collectives behind the C ABI (include/bsk.h "collectives", csrc/comm.cpp) ----------------
// In the reference Reduce and GroupByKey come from IgnisHPC in the same binary (bigseqkit/stats.go:91, grep.go:175,
// rmdup.go:97).  Here one goroutine per GPU (locked to its OS thread: the HIP device is per thread) owns a context and a
// communicator of one group made by bsk_comm_init_all -- librccl over xGMI when every rank has a GPU of its own.  No
// Python, no torch.  (UNVERIFIED like the rest of this file: no Go toolchain here; the same call sequence is what
// `bigseqkit <cmd> ... --devices` runs in C++ and tests/test_devices_native_gpu.py exercises.)

// Comms is a group of communicators, one per device of the list.
type Comms struct{ c []*C.bsk_comm }

func NewComms(devices []int) (*Comms, error) {
	if len(devices) == 0 {
		return nil, errors.New("bigseqkit: NewComms needs at least one device")
	}
	devs := make([]C.int, len(devices))
	for i, d := range devices {
		devs[i] = C.int(d)
	}
	out := make([]*C.bsk_comm, len(devices))
	if rc := C.bsk_comm_init_all(C.int(len(devices)), &devs[0], &out[0]); rc != C.BSK_OK {
		return nil, errors.New(C.GoString(C.bsk_comm_error(nil)))
	}
	return &Comms{out}, nil
}

func (g *Comms) Close() {
	for _, c := range g.c {
		C.bsk_comm_destroy(c)
	}
}

// onEveryDevice runs f(rank) on one locked goroutine per device and returns the first error.  EVERY rank must enter every
// collective f reaches, also after a failure of its own: f keeps its own error in a variable and calls g.agree before the
// collective that carries data (the way run_devices of cli/bigseqkit.cpp does it).
func onEveryDevice(n int, f func(rank int) error) error {
	errs := make([]error, n)
	var wg sync.WaitGroup
	for r := 0; r < n; r++ {
		wg.Add(1)
		go func(r int) {
			defer wg.Done()
			runtime.LockOSThread()
			defer runtime.UnlockOSThread()
			errs[r] = f(r)
		}(r)
	}
	wg.Wait()
	for _, e := range errs {
		if e != nil {
			return e
		}
	}
	return nil
}

var errOtherRank = errors.New("bigseqkit: another rank failed before the collective (its own error says why)")

// agree is the collective every rank enters before one that carries data: the sum over the ranks of "I have failed"
// (bsk_count_allreduce on one word).  It returns nil when nobody has -- the ranks go on together --, the rank's own error
// when it has one, errOtherRank when only a peer failed, and the communicator's error when the reduction itself failed
// (then nobody may enter another collective on g).  A rank that returned BEFORE this call would leave its peers waiting
// in ncclAllReduce for ever (ADVICE r05).
func (g *Comms) agree(rank int, own error) error {
	var bad C.uint64_t
	if own != nil {
		bad = 1
	}
	if rc := C.bsk_count_allreduce(g.c[rank], &bad, nil); rc != C.BSK_OK {
		if own != nil {
			return own
		}
		return errors.New(C.GoString(C.bsk_comm_error(g.c[rank])))
	}
	if own != nil {
		return own
	}
	if bad != 0 {
		return errOtherRank
	}
	return nil
}

// StatsN: Stats over the shards of `input` on the devices of `g`, shard k on device k (len(input.Shards) == len(devices)):
// bsk_stats_run per rank, then StatsReduce + collect in ONE call (bsk_stats_collect_reduced: a single ncclAllReduce of the
// dense vector; the overflow lists of chromosome-sized records are exchanged only when the reduced vector counts any).
func StatsN(name, format string, input *SeqFrame, o *SeqKitStatsOptions, g *Comms, devices []int) (*StatInfo, error) {
	o.setDefaults()
	js := OptionsToString(o)
	var result *StatInfo
	err := onEveryDevice(len(devices), func(rank int) error {
		// (everything that can fail on this rank alone happens before g.agree; its error travels with that reduction)
		op, own := newBskOp("Stats", js, devices[rank])
		if own == nil {
			defer op.Close()
			d := input.Shards[rank].Data
			if len(d) > 0 {
				if rc := C.bsk_stats_run(op.ctx, unsafe.Pointer(&d[0]), C.size_t(len(d)), 0, C.int(input.Format), C.int64_t(rank), nil, nil); rc != C.BSK_OK {
					own = op.err()
				}
			}
		}
		if e := g.agree(rank, own); e != nil {
			return e
		}
		capN := 1 << 20
		keys := make([]C.int64_t, capN)
		vals := make([]C.int64_t, capN)
		var n C.size_t
		if rc := C.bsk_stats_collect_reduced(op.ctx, g.c[rank], nil, nil, &keys[0], &vals[0], C.size_t(capN), &n); rc != C.BSK_OK {
			return op.err()
		}
		if rank != 0 {
			return nil
		}
		var info C.bsk_statinfo
		if rc := C.bsk_stats_finalize(op.ctx, &keys[0], &vals[0], n, &info); rc != C.BSK_OK {
			return op.err()
		}
		result = &StatInfo{name, format, C.GoString(&info._type[0]),
			uint64(info.num), uint64(info.len_sum), uint64(info.gap_sum), uint64(info.len_min),
			float64(info.len_avg), uint64(info.len_max), uint64(info.n50), int(info.l50),
			float64(info.q1), float64(info.q2), float64(info.q3), float64(info.q20), float64(info.q30)}
		return nil
	})
	return result, err
}

// GrepCountN: per-rank counts summed by bsk_count_allreduce (GrepReduceCount, bigseqkit/grep.go:161-180).
func GrepCountN(input *SeqFrame, o *SeqKitGrepOptions, g *Comms, devices []int) (uint64, error) {
	o.setDefaults()
	t := true
	o.Count = &t
	js := OptionsToString(o)
	var total uint64
	err := onEveryDevice(len(devices), func(rank int) error {
		var cnt C.uint64_t
		op, own := newBskOp("Grep", js, devices[rank])
		if own == nil {
			defer op.Close()
			d := input.Shards[rank].Data
			var ptr unsafe.Pointer
			if len(d) > 0 {
				ptr = unsafe.Pointer(&d[0])
			}
			var out C.bsk_out
			rc := C.bsk_grep_run(op.ctx, ptr, C.size_t(len(d)), 0, C.int(input.Format), C.int64_t(rank), nil, &out)
			if rc == C.BSK_OK {
				rc = C.bsk_grep_last_count(op.ctx, &cnt)
			}
			if rc != C.BSK_OK {
				own = op.err()
				cnt = 0
			}
		}
		if e := g.agree(rank, own); e != nil {
			return e
		}
		if rc := C.bsk_count_allreduce(g.c[rank], &cnt, nil); rc != C.BSK_OK {
			return errors.New(C.GoString(C.bsk_comm_error(g.c[rank])))
		}
		if rank == 0 {
			total = uint64(cnt)
		}
		return nil
	})
	return total, err
}

// RmDupN: the survivors of every rank's shard (file order; their concatenation equals the single-GPU output), duplicates
// found across ranks by bsk_rmdup_dist_run (24-byte tuples to owner = key % N by grouped ncclSend / ncclRecv, keep bytes
// back, and the byte comparison of every duplicate with its survivor -- also one on another rank; GroupByKey + RmDupCheck,
// bigseqkit/rmdup.go:97, bigseqkit-lib/rmdup.go:193-211).
func RmDupN(input *SeqFrame, o *SeqKitRmDupOptions, g *Comms, devices []int) (*Result, error) {
	o.setDefaults()
	js := OptionsToString(o)
	res := &Result{Parts: make([][]byte, len(devices))}
	var mu sync.Mutex
	err := onEveryDevice(len(devices), func(rank int) error {
		d := input.Shards[rank].Data
		var dev unsafe.Pointer
		op, own := newBskOp("RmDup", js, devices[rank])
		if own == nil {
			defer op.Close()
			C.bsk_device_select(C.int(devices[rank]))
			dev = C.bsk_device_alloc(C.size_t(len(d) + 1))
			if dev == nil {
				own = errors.New(C.GoString(C.bsk_global_error()))
			} else {
				defer C.bsk_device_free(dev)
				if len(d) > 0 {
					if rc := C.bsk_device_copy(dev, unsafe.Pointer(&d[0]), C.size_t(len(d)), C.BSK_COPY_H2D); rc != C.BSK_OK {
						own = errors.New(C.GoString(C.bsk_global_error()))
					}
				}
			}
		}
		if e := g.agree(rank, own); e != nil {
			return e
		}
		// (from here on bsk_rmdup_dist_run carries every phase's outcome with its next collective itself: csrc/comm.cpp)
		var out C.bsk_out
		if rc := C.bsk_rmdup_dist_run(op.ctx, g.c[rank], dev, C.size_t(len(d)), C.int(input.Format), nil, &out); rc != C.BSK_OK {
			return op.err()
		}
		buf := make([]byte, int(out.len))
		if out.len > 0 {
			if rc := C.bsk_out_to_host(op.ctx, &out, unsafe.Pointer(&buf[0]), out.len); rc != C.BSK_OK {
				return op.err()
			}
		}
		mu.Lock()
		res.Parts[rank] = buf
		res.Bytes += uint64(out.len)
		res.Records += uint64(out.records)
		mu.Unlock()
		return nil
	})
	if err != nil {
		return nil, err
	}
	return res, nil
}
