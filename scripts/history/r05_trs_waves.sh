#!/usr/bin/env bash
# round 5: k_translate_stream at 3 / 4 / 5 waves per SIMD in ONE visit (gpurun_alt/libbsk_w<N>.so: -DBSK_TRS_WAVES=N)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for w in 3 4 5; do echo -n "w$w: "; BSK_LIB=$PWD/gpurun_alt/libbsk_w$w.so python scripts/bench_translate_var.py 50 5 2>&1 | tail -1 | cut -c1-120; done; done
