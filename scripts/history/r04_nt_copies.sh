#!/usr/bin/env bash
# (the knobs -DBSK_SEGCOPY_NT / -DBSK_COMPACT_NT -- __builtin_nontemporal_load / _store in k_seg_copy and k_names_compact -- were taken out of the
# sources again after this measurement: DESIGN.md "Measured and dropped")
# non-temporal loads / stores in the copy kernels: k_seg_copy (rmdup) and the gather of the slices (subseq, seq -n)
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "" "-DBSK_SEGCOPY_NT=1" "-DBSK_SEGCOPY_NT=2" "-DBSK_SEGCOPY_NT=3"; do bash scripts/variant_src.sh ops_segcopy.hip "$f" rmdup; done
bash scripts/variant_src.sh ops_segcopy.hip "" grep
for f in "" "-DBSK_COMPACT_NT=1" "-DBSK_COMPACT_NT=3"; do bash scripts/variant_src.sh stream_names.hip "$f" subseq,seq; done
bash scripts/variant_src.sh stream_names.hip "" grep
