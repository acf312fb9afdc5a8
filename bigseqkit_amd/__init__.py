"""bigseqkit_amd -- MI355X-native engine behind BigSeqKit's per-record hot path.

The product is libbsk.so (hand-written HIP for gfx950 behind the C ABI in
include/bsk.h).  This package is the thin host-side mirror of the reference's
driver library used by tests and bench.py; it has no CPU fallback.
"""
from ._lib import BskError, FORMAT_FASTA, FORMAT_FASTQ, lib  # noqa: F401  (fails loudly if libbsk.so is missing)
from .options import (SeqKitConfig, SeqKitStatsOptions, SeqKitSeqOptions, SeqKitGrepOptions,  # noqa: F401
                      SeqKitLocateOptions, SeqKitSubseqOptions, SeqKitTranslateOptions, SeqKitRmDupOptions,
                      SeqKitFq2FaOptions, SeqKitRangeOptions, SeqKitHeadOptions, SeqKitDuplicateOptions, SeqKitRenameOptions, SeqKitSortOptions, SeqKitFaidxOptions, SeqKitPairOptions, SeqKitCommonOptions, SeqKitConcatOptions)
from .api import (SeqFrame, ReadFASTA, ReadFASTAN, ReadFASTQ, ReadFASTQN, Operator, Stats, StatsString,  # noqa: F401
                  stats_map, Seq, build_index, Grep, GrepCount, Subseq, Translate, RmDup, Locate, Fq2Fa, Range, Head,
                  Duplicate, Count, Rename, Sort, Faidx, Pair, Common, Concat, FaidxQuery)
