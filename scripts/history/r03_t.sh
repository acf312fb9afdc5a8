#!/usr/bin/env bash
# targeted GPU tests + optional bench:  bash scripts/r03_t.sh "<pytest args>" [bench tag] [bench args...]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
T="$1"; TAG="$2"; shift; shift
if [ -n "$T" ]; then (timeout 1500 python -m pytest $T -q -x 2>&1 | tail -25) > $O/t_tests.log 2>&1; cat $O/t_tests.log; fi
if [ -n "$TAG" ]; then bash scripts/r03_bench.sh $TAG --no-cpu-baseline "$@"; fi
