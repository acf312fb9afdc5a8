#!/usr/bin/env bash
# k_names with the header bytes gathered asynchronously into LDS (global_load_lds_dwordx4) by the call before the one that
# writes them: 7 waves per SIMD (164 spilled registers), 6 (8), 5 (none); and the direct path for comparison
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "" "-DBSK_NAMES_WAVES=6" "-DBSK_NAMES_WAVES=5" "-DBSK_NAMES_ASYNC=0"; do bash scripts/variant_src.sh stream_names.hip "$f" seq; done
bash scripts/variant_src.sh stream_names.hip "" grep > /dev/null
