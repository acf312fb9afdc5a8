// `translate`: per (record, frame) element.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {

struct TextTableH {  // host-visible mirror of TextTable (text_dev.hpp)
    const uint32_t* text_w;
    const uint64_t* lin_off;
    const uint8_t* lin;
    uint64_t lin_n = 0;  // bytes in lin
};

// A FASTA shard in which EVERY record has the same header length, line width, sequence length and therefore the same
// byte stride (synthetic benchmarks, fixed-length amplicon / CDS sets): records, text layout and output places follow from
// the record number, so no pass over the 50 GB has to find the '>' bytes first (k_fasta_starts: 8.8 ms of a 58 ms call at
// C4) and no table has to be built, scanned or read.  The host proposes the layout from the head of the shard
// (translate_uniform_probe); k_translate_wide<G, true> VERIFIES every byte of every record against it -- the '>' at i * S
// behind a '\n', no line break inside the header and one at its end, every sequence line break at its place and nothing but
// A C G T between them, the final '\n' at i * S + S - 1 (or the end of the shard) -- and one record that differs sends the
// whole call back to the table paths (redo count), with nothing of this pass's output kept.
struct UniformLayout {
    uint32_t on;       // 0: the record table decides
    uint32_t H;        // header line length, marker included (== RecordTable::l_head)
    uint32_t L;        // bases per record
    uint32_t W;        // bases per full line (0: the sequence is on one line)
    uint64_t S;        // bytes from one '>' to the next
    uint64_t n;        // records
    uint64_t out_S;    // output bytes per record (all its frames)
    uint32_t len[6];   // output bytes of element k of a record ...
    uint32_t off[6];   // ... and where it begins inside the record's output
};

struct TranslateParams {  // Translate options after Before() (bigseqkit-lib/translate.go:33-64)
    int fastq;
    int nframes;
    int frames[6];
    int trim, clean, allow_unknown, init_m, append_frame;
    int line_width;
    int id_mode;
    const uint8_t* codon;   // device, 4096 bytes: aa of the IUPAC-coded codon (c1 << 8 | c2 << 4 | c3); 0 = unknown
    const uint8_t* start;   // device, 4096 bytes: 1 when the codon is a start codon of the table
    const uint8_t* codon_rc; // device, 4096 bytes: aa of the REVERSE COMPLEMENT of the codon with that index
    const uint8_t* start_rc; // device, 4096 bytes
    const uint8_t* iupac;    // device, 256 bytes: byte -> 4-bit IUPAC code (0 = not a base)
    const uint8_t* baked;    // device, 8192 bytes, 16-byte aligned: codon ++ codon_rc with -x (0 -> 'X') and --clean ('*' -> 'X') applied
    // device, 16384 bytes, 16-byte aligned: for two codons of plain letters in 2-bit codes ((b >> 1) & 3: A 0, C 1, T 2, G 3; first
    // base in the low bits; index = first codon | second << 6) the two residues of `baked` as uint16: [0, 8192) forward, first
    // codon's residue in the low byte; [8192, 16384) reverse complement, SECOND codon's residue in the low byte
    const uint8_t* pair;
    // chromosome-sized records (l_seq >= long_thresh): skipped by the per-record kernels, translated by whole blocks
    // (launch_translate_long: one block per 16 KiB of an element's body)
    const uint32_t* long_list;
    uint64_t long_count;
    uint32_t long_thresh;
    UniformLayout uni;
};

constexpr uint32_t ERR_UNKNOWN_CODON = 256u;

// round 5: FASTA records that do not all look alike, in ONE pass -- record starts, sizes, output offsets (a chain over the
// ranges) and the translation itself by persistent blocks over small line-start ranges (anchors / queue from launch_prep in
// line mode).  chain: nranges words, fin[0..1] and *redo_count zeroed by the caller; fin[0] = bytes written, fin[1] = records,
// *redo_count != 0: something did not fit, the output is to be discarded and the table paths take the call
hipError_t launch_translate_stream(int blocks, const uint8_t* buf, uint64_t buf_n, const uint64_t* anchors, uint32_t nranges, uint32_t* queue,
                                   const TranslateParams& P, uint8_t* out, uint64_t out_cap, uint64_t* chain, uint64_t* fin,
                                   uint64_t* redo_count, uint64_t* status, hipStream_t st,
                                   uint32_t span_hint = 0 /* mean record bytes of the head sample: predict-and-verify the record starts; 0: search every range in full */);
int translate_stream_max_blocks_per_cu();
// elements are numbered record * nframes + f
hipError_t launch_translate_size(const uint8_t* buf, const RecordTable& t, const TextTableH& tt,
                                 const TranslateParams& P, uint32_t* out_len, uint64_t* status, hipStream_t st);
// all requested frames of a record from ONE read of its bases (G lanes per record).  With `redo` (one zeroed byte per
// record) k_translate_wide<wide_lanes> translates the records of plain A/C/G/T text first and k_translate_frames4 the ones
// it flags; buf_n = bytes in the shard (the wide kernel reads 64 bytes at a time and must not pass the end)
// k_translate_wide asks for 52 bytes at `buf` on behalf of idle lanes: shards smaller than this go to k_translate_frames4
// alone -- which validates nothing, so a record table whose lengths are DERIVED (the light table) needs the full pass there
// (a 62-byte shard with lines of 46, 5 and 1 letters came out as 53 bases with a line break among them: fuzz seed 1414, round 6)
constexpr uint64_t TRANSLATE_WIDE_MIN_BYTES = 64;
hipError_t launch_translate_frames(int lanes_per_record, const uint8_t* buf, const RecordTable& t, const TextTableH& tt,
                                   const TranslateParams& P, const uint32_t* out_len, const uint64_t* out_off,
                                   uint8_t* out, uint64_t* status, hipStream_t st, uint64_t buf_n = 0, uint8_t* redo = nullptr,
                                   int wide_lanes = 16, uint64_t* redo_count = nullptr /* zeroed device word */,
                                   int variant = 0 /* 1: the round-1 frames kernel only, 2: frames4 without the wide kernel */,
                                   uint64_t* redo_left_stop = nullptr /* receives the number of records the wide kernel left; when
                                   it is non-zero nothing more is launched */);
// the uniform-layout pass (P.uni.on): k_translate_wide<wide_lanes, true> alone, no tables; *redo_left receives the number
// of records that did not verify (the caller then starts over on the table paths)
hipError_t launch_translate_uniform(int wide_lanes, const uint8_t* buf, uint64_t buf_n, const TranslateParams& P, uint8_t* out,
                                    uint64_t* redo_count /* zeroed device word */, uint64_t* status, hipStream_t st);
// elements of the records in P.long_list; max_len = longest of those sequences
hipError_t launch_translate_long(const uint8_t* buf, const RecordTable& t, const TextTableH& tt, const TranslateParams& P,
                                 const uint32_t* out_len, const uint64_t* out_off, uint8_t* out, uint64_t* status,
                                 uint64_t max_len, hipStream_t st);
hipError_t launch_translate_emit(const uint8_t* buf, const RecordTable& t, const TextTableH& tt,
                                 const TranslateParams& P, const uint32_t* out_len, const uint64_t* out_off,
                                 uint8_t* out, uint64_t* status, hipStream_t st);

}  // namespace bsk
