#!/usr/bin/env bash
# round 6, A/B 4b: stage times and FETCH_SIZE of k_names with / without the header bytes from LDS
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for v in "-DBSK_NAMES_HEAD16=0" ""; do
  BSK_OUT=slices BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_names.hip "$v" seq
  bash scripts/pmc_ops_traffic.sh seq 1.0 | grep -i "k_names"
done
} > gpurun_out/r06_ab4b.log 2>&1
tail -40 gpurun_out/r06_ab4b.log
