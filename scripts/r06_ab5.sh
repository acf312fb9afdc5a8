#!/usr/bin/env bash
# round 6, A/B 5: scalar-register pressure of the streaming passes -- instructions per tile of k_names with the once-per-range
# arguments read from the argument block (BSK_KARG) and with the newline prefilter's constant as a literal
cd "$(dirname "$0")/.."
{
timeout 600 python -m pytest tests/test_names_gpu.py tests/test_seq_gpu.py -x -q -m gpu 2>&1 | tail -2
for v in "" "-DBSK_NL_SGPR=0"; do
  BSK_OUT=slices bash scripts/variant_valu.sh stream_names.hip "$v" seq k_names 100e9
  BSK_OUT=slices BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_names.hip "$v" seq
done
} > gpurun_out/r06_ab5.log 2>&1
grep -v "^  File\|^    " gpurun_out/r06_ab5.log | tail -12
