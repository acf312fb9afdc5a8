#!/usr/bin/env bash
# GPU box: the randomised cross-check of tests/test_fuzz_gpu.py over many seeds under several selections of the run-time
# switches (BSK_FUZZ_ENV); one line per selection.  usage: bash scripts/fuzz_soak.sh [seeds]
cd "$(dirname "$0")/.."
SEEDS=${1:-1600}
mkdir -p gpurun_out
for sel in "BSK_OUT=slices,BSK_SEGCOPY=force" "BSK_FILTER=off,BSK_NAMES=off,BSK_SUBSEQ=table,BSK_SEGCOPY=off" \
           "BSK_TRANSLATE_INDEX=light,BSK_TEXT=view,BSK_RMDUP_KEYS=two-key" "BSK_TRANSLATE_INDEX=full,BSK_RMDUP_PLACE=off,BSK_RMDUP_HASH=xxh64,BSK_STATS_FASTA=events" \
           "BSK_MIN_RANGE_BYTES=1024,BSK_TRANSLATE_STREAM=force,BSK_LONG_BYTES=600" "BSK_INDEX=twopass,BSK_GREP_SHIFTAND=off,BSK_LOCATE_NOPRE=1,BSK_SORT=lsd,BSK_RMDUP=table"; do
  tag=$(echo "$sel" | tr -c 'A-Za-z0-9\n' '_' | cut -c1-60)
  BSK_FUZZ_ENV="$sel" BSK_FUZZ_SEEDS=$SEEDS timeout 1500 python -m pytest tests/test_fuzz_gpu.py -q -m gpu -n 6 -p no:cacheprovider > gpurun_out/fuzz_soak_$tag.log 2>&1
  echo "== $sel: $(grep -E 'passed|failed' gpurun_out/fuzz_soak_$tag.log | tail -1)"
  grep -E "^FAILED" gpurun_out/fuzz_soak_$tag.log | head -5 | cut -c1-300
done
