#!/usr/bin/env bash
# round 5, visit 7: --devices in one process (worker threads + librccl behind the C ABI)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest ${2:-tests/test_devices_native_gpu.py} -q -x --timeout 150 -k "${1:-test}" 2>&1 | grep -v "^  File\|^$" | tail -40) > $O/r05g_tests.log 2>&1
cat $O/r05g_tests.log
