#!/usr/bin/env bash
# the whole GPU suite with durations -> gpurun_out/<tag>_gputests.log
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=40 2>&1 | tail -70 ) > $O/${TAG}_gputests.log 2>&1
tail -75 $O/${TAG}_gputests.log
