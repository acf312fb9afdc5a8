#!/usr/bin/env bash
# (knobs removed again: verify 2.36 -> 3.37 ms with nt loads -- neighbouring lanes share sectors; compact 0.93 -> 1.01)
# non-temporal loads in the byte comparison of rmdup (gathers, no reuse) and in the table compaction (a stream)
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "" "-DBSK_VERIFY_NT=1"; do bash scripts/variant_src.sh ops_rmdup.hip "$f" rmdup; done
bash scripts/variant_src.sh ops_rmdup.hip "" grep >/dev/null
for f in "" "-DBSK_COMPACT_RM_NT=1"; do bash scripts/variant_src.sh stream_rmdup.hip "$f" rmdup; done
