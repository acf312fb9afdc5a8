#!/usr/bin/env bash
# (needs gpurun_alt/libbsk_<variant>.so built by hand: HIPCC="/opt/rocm/bin/hipcc -DBSK_PLACE_CH=.. -DBSK_PLACE_W=.." ./build.sh;
#  the seqref variant was a source change that is not in the tree: see the comment in k_rmdup_place)
# round 5: k_rmdup_place variants in ONE visit (boxes differ by several per cent): libraries built with -DBSK_PLACE_CH / _W,
# and the one-word-per-record gather (BSK_RMDUP_SEQREF=off: three gathers per survivor)
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; BSK_BENCH_PROFILE=1 python scripts/bench_ops.py 1 5 rmdup 2>&1 | tail -1 | python -c "
import sys,json,re
d=json.loads(sys.stdin.read()); v=list(d.values())[0]; n=v['note']; k=json.loads(n[n.index('{'):]); print(v['ms'], k['k_rmdup_place'], k['k_rmdup_compact'], k['k_seg_copy'], k['k_rmdup_stream'])"; }
for rep in 1 2 3; do
  BSK_LIB=$PWD/gpurun_alt/libbsk_ch4.so run ch4
  run seqref
  BSK_RMDUP_SEQREF=off run seqref_off
done
