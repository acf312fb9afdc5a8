#!/usr/bin/env bash
# variants of ONE source on the GPU box, stage times included:  bash scripts/r03_var.sh <src> <ops> <scale> "<defs 1>" "<defs 2>" ...
cd "$(dirname "$0")/.."
SRC=$1; OPS=$2; SCALE=$3; shift; shift; shift
export BSK_BENCH_PROFILE=1
for D in "$@"; do bash scripts/variant_src.sh $SRC "$D" $OPS $SCALE 2>&1 | grep -v "amdgpu.ids"; done
bash scripts/variant_src.sh $SRC "" $OPS $SCALE > /dev/null 2>&1   # leave the default build behind
