#!/usr/bin/env bash
# round 6, A/B 4c: where the time of k_names goes (experiments with wrong results: parts of the sink removed)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for v in "-DBSK_NAMES_DIAG=1" "-DBSK_NAMES_DIAG=2" "-DBSK_NAMES_DIAG=2 -DBSK_NAMES_HEAD16=0"; do
  BSK_DIAG=1 BSK_OUT=slices BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_names.hip "$v" seq
done
} > gpurun_out/r06_ab4c.log 2>&1
tail -40 gpurun_out/r06_ab4c.log
