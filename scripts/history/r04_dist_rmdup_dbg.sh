#!/usr/bin/env bash
# which of the two changes of the multi-GPU rmdup path (sort-based owner resolve / segment-copy emit) loses a record
cd $GRAFT_REPO_ROOT
for env in "X=1" "BSK_RMDUP=table" "BSK_SEGCOPY=off"; do
  echo "== $env"
  env $env timeout 600 python -m pytest tests/test_run_multi_gpu.py -m gpu -q -x -k "rmdup" 2>&1 | tail -2
done
