#!/usr/bin/env bash
# RESULT (k_rmdup_stream, 25 GB): as shipped 6.37-6.44 ms; without the hash but WITH the tile staged in LDS 7.14 (!); without
# hash and staging 5.32-5.35.  NOT a clean price of the staging: without tile and key the compiler drops the
# LDS tile (29.8 -> 9.2 KB per block) and 21 VGPRs (91 -> 70), so that variant runs 7 waves per SIMD instead of 5 -- occupancy as much
# as work.  The multiply-adds of the key alone are not what bounds the pass.  (The source switch BSK_EXP_NOHASH gave wrong keys on purpose: not in the tree.)
# round 5 (experiment, wrong answers on purpose): what the hashing pass of rmdup costs without its hash (nh1: the key is a
# function of the line's position and length) and without hash AND tile staging (nh2) -- k_rmdup_stream only
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in base nh1 nh2; do echo -n "$v: "; BSK_LIB=$PWD/gpurun_alt/libbsk_$v.so BSK_BENCH_PROFILE=1 python -O scripts/bench_ops.py 1 5 rmdup 2>&1 | tail -1 | grep -o 'k_rmdup_stream[^,]*'; done; done
