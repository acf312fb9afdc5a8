#!/usr/bin/env bash
# round 5: k_translate_stream with 4 / 8 / 16 loads per thread in flight in the '>' search, and the range size, in ONE visit
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for f in 4 8 16; do echo -n "f$f: "; BSK_LIB=$PWD/gpurun_alt/libbsk_f$f.so python scripts/bench_translate_var.py 50 5 2>&1 | tail -1 | grep -o "[0-9.]* ms per call"; done; done
for rep in 1 2; do for c in 786432 1048576 1572864; do echo -n "chunk $c: "; BSK_MIN_RANGE_BYTES=$c BSK_LIB=$PWD/gpurun_alt/libbsk_f8.so python scripts/bench_translate_var.py 50 5 2>&1 | tail -1 | grep -o "[0-9.]* ms per call"; done; done
