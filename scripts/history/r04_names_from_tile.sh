#!/usr/bin/env bash
# (the knobs were taken out of stream_subseq.hip again: 5.65 ms per 25 GB = 22.6 ms at C2 against 20.0 for k_names as it is)
# what a names pass that copies the header lines out of the LDS tile (the subseq sink, header pieces only) would cost:
# k_subseq_stream with only the role-0 pieces, with and without non-temporal tile loads, at 25 GB (x 4 = C2)
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "-DBSK_SUBSEQ_ONLY_HEAD=1" "-DBSK_SUBSEQ_ONLY_HEAD=1 -DBSK_SUBSEQ_NT=1" "-DBSK_SUBSEQ_ONLY_HEAD=1 -DBSK_SUBSEQ_NT=1 -DBSK_SUBSEQ_WAVES=6" "-DBSK_SUBSEQ_NT=1"; do bash scripts/variant_src.sh stream_subseq.hip "$f" subseq; done
