#!/usr/bin/env bash
# round 6, A/B 4f: k_names -- window sizes at 7 waves per SIMD, temporal loads, twice each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for rep in 1 2; do
for win in 256 288 320 352; do
  BSK_OUT=slices BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_names.hip "-DBSK_NAMES_NT=0 -DBSK_NAMES_WAVES=7 -DBSK_NAMES_WINDOW=$win" seq
done
BSK_OUT=slices BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_names.hip "-DBSK_NAMES_NT=0 -DBSK_NAMES_WAVES=8 -DBSK_NAMES_WINDOW=256" seq
BSK_OUT=slices BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_names.hip "-DBSK_NAMES_NT=0 -DBSK_NAMES_WAVES=7 -DBSK_NAMES_WINDOW=320 -DBSK_NAMES_TE=0" seq
done
} > gpurun_out/r06_ab4f.log 2>&1
grep -A1 "^==" gpurun_out/r06_ab4f.log | grep -v "^--" | paste - - | awk '{print $3,$4,$5,$6, $12, $13, $14}' 
