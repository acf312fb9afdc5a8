#!/usr/bin/env bash
# GPU box: the randomised cross-check with ranges far below a tile (edge tiles everywhere) -- see scripts/fuzz_soak.sh
cd "$(dirname "$0")/.."
SEEDS=${1:-1200}
mkdir -p gpurun_out
for sel in "BSK_MIN_RANGE_BYTES=256" "BSK_MIN_RANGE_BYTES=64,BSK_OUT=slices" "BSK_MIN_RANGE_BYTES=512,BSK_TRANSLATE_STREAM=force,BSK_SEGCOPY=force"; do
  tag=$(echo "$sel" | tr -c 'A-Za-z0-9\n' '_' | cut -c1-60)
  BSK_FUZZ_ENV="$sel" BSK_FUZZ_SEEDS=$SEEDS timeout 1500 python -m pytest tests/test_fuzz_gpu.py -q -m gpu -n 6 -p no:cacheprovider > gpurun_out/fuzz_soak_$tag.log 2>&1
  echo "== $sel: $(grep -E 'passed|failed' gpurun_out/fuzz_soak_$tag.log | tail -1)"
  grep -E "^FAILED" gpurun_out/fuzz_soak_$tag.log | head -5 | cut -c1-300
  grep -n "AssertionError" -B2 -A12 gpurun_out/fuzz_soak_$tag.log | head -60 | cut -c1-800
done
