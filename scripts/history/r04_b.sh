#!/usr/bin/env bash
# round 4, visit b: the whole GPU suite on the one-read-back host paths, then the kernel timelines of the five bench operators
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -q -x -n 4 2>&1 | tail -8) > $O/r04b_tests.log 2>&1
cat $O/r04b_tests.log
for o in grep seq subseq rmdup translate; do echo "=== $o"; bash scripts/timeline_ops.sh $o 1.0 r04tlb > $O/r04tlb_$o.txt 2>&1; tail -45 $O/r04tlb_$o.txt; done
