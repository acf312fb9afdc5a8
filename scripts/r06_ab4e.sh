#!/usr/bin/env bash
# round 6, A/B 4e: k_names -- non-temporal tile loads x waves per SIMD x window, twice each (run-to-run spread)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for rep in 1 2; do
for nt in 0 1; do for w in 6 7; do for win in 320 384; do
  BSK_OUT=slices BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_names.hip "-DBSK_NAMES_NT=$nt -DBSK_NAMES_WAVES=$w -DBSK_NAMES_WINDOW=$win" seq
done; done; done; done
} > gpurun_out/r06_ab4e.log 2>&1
grep -A1 "^==" gpurun_out/r06_ab4e.log | grep -v "^--" | paste - - | awk '{print $3,$4,$5, $12, $13}' 
