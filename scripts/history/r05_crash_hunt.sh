#!/usr/bin/env bash
# round 5: the full suite aborted / segfaulted in pure-Python code at 59 % and at 71 % -- something earlier corrupts memory.
cd $GRAFT_REPO_ROOT
export MALLOC_CHECK_=3 MALLOC_PERTURB_=165
timeout 2400 python -X faulthandler -m pytest $@ -m gpu -v -x --timeout 600 > gpurun_out/r05_hunt.log 2>&1
echo "rc=$?"; grep -c PASSED gpurun_out/r05_hunt.log; grep -n "Fatal\|Aborted\|Segmentation\|corrupt\|invalid pointer\|double free" gpurun_out/r05_hunt.log | head -5
grep -n "PASSED\|FAILED" gpurun_out/r05_hunt.log | tail -2 | cut -c1-200
grep -n "Fatal Python" -A6 gpurun_out/r05_hunt.log | head -20 | cut -c1-200
