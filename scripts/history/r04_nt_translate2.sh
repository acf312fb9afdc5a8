#!/usr/bin/env bash
# (the knobs -DBSK_TRW_LD_NT / -DBSK_TRW_ST_AUX were taken out of ops_translate.hip again after this measurement: DESIGN.md "Measured and dropped")
# k_translate_wide: cache policy of the 16-byte output stores (aux: 1 sc0, 2 nt, 16 sc1 and their sums), twice each
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "" "-DBSK_TRW_ST_AUX=2" "-DBSK_TRW_ST_AUX=16" "-DBSK_TRW_ST_AUX=18" "-DBSK_TRW_ST_AUX=17" "" "-DBSK_TRW_ST_AUX=2"; do bash scripts/variant_src.sh ops_translate.hip "$f" translate; done
