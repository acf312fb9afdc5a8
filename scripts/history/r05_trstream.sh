#!/usr/bin/env bash
# round 5: the one-pass translation of records that differ -- range size of the pass (4 waves per SIMD, 7 spilled registers)
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_translate_stream_gpu.py -q --timeout 120 2>&1 | tail -3)
python scripts/bench_translate_var.py 50 3 2>&1 | tail -1
for c in 524288 1048576 2097152; do BSK_MIN_RANGE_BYTES=$c python scripts/bench_translate_var.py 50 3 2>&1 | tail -1; done
BSK_TRANSLATE_STREAM=off python scripts/bench_translate_var.py 50 3 2>&1 | tail -1
KIND=2 python scripts/bench_translate_var.py 50 3 2>&1 | tail -1
KIND=2 BSK_TRANSLATE_INDEX=light python scripts/bench_translate_var.py 50 3 2>&1 | tail -1
