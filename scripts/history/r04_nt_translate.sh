#!/usr/bin/env bash
# (the knobs -DBSK_TRW_LD_NT / -DBSK_TRW_ST_AUX were taken out of ops_translate.hip again after this measurement: DESIGN.md "Measured and dropped")
# k_translate_wide: non-temporal window loads / non-temporal 16-byte stores (the output is written once, the input read once)
cd $GRAFT_REPO_ROOT
export BSK_BENCH_PROFILE=1
for f in "" "-DBSK_TRW_LD_NT=1" "-DBSK_TRW_ST_AUX=2" "-DBSK_TRW_LD_NT=1 -DBSK_TRW_ST_AUX=2"; do bash scripts/variant_src.sh ops_translate.hip "$f" translate; done
