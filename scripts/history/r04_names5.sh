#!/usr/bin/env bash
# k_names, where the 3 - 4 ms above k_stats go: EXP=9 no global stores (loads + LDS kept), EXP=10 no header loads (stores kept)
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "" "-DBSK_NAMES_EXP=9" "-DBSK_NAMES_EXP=10" "" "-DBSK_NAMES_EXP=9" "-DBSK_NAMES_EXP=10"; do bash scripts/variant_src.sh stream_names.hip "$f" seq; done
bash scripts/variant_src.sh stream_names.hip "" grep > /dev/null
