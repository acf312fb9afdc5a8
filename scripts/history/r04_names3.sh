#!/usr/bin/env bash
# what the second fetch of the header sectors costs k_names: the same kernel copying its names from a line it has in cache
# (EXP=7: wrong output, timing only), with ordinary and with non-temporal tile loads; EXP=8: no name stores either
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "" "-DBSK_NAMES_EXP=7" "-DBSK_NAMES_EXP=7 -DBSK_NAMES_NT=1" "-DBSK_NAMES_NT=1" "-DBSK_NAMES_EXP=8" "-DBSK_NAMES_EXP=8 -DBSK_NAMES_NT=1"; do bash scripts/variant_src.sh stream_names.hip "$f" seq; done
bash scripts/variant_src.sh stream_names.hip "" grep > /dev/null
