#!/usr/bin/env bash
# round 6, A/B 4: k_names with the header bytes from the pass's LDS (BSK_NAMES_HEAD16), non-temporal tile loads (BSK_NAMES_NT)
# and the small-name stores without a scratch array -- parity first, then time at C2, then FETCH_SIZE / WRITE_SIZE
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_names_gpu.py tests/test_seq_gpu.py tests/test_fuzz_gpu.py tests/test_out_slices_gpu.py -x -q -m gpu 2>&1 | tail -5
for v in "-DBSK_NAMES_HEAD16=0" "-DBSK_NAMES_NT=0" "-DBSK_NAMES_WAVES=6" "-DBSK_NAMES_WINDOW=256" "-DBSK_NAMES_WINDOW=320" ""; do
  BSK_OUT=slices BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_names.hip "$v" seq
done
bash scripts/pmc_ops_traffic.sh seq 1.0 | grep -i "k_names"
} > gpurun_out/r06_ab4.log 2>&1
tail -40 gpurun_out/r06_ab4.log
