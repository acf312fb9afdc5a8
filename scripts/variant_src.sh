#!/usr/bin/env bash
# experiment helper (GPU box): rebuild ONE source of libbsk.so with extra -D flags and time some commands of bench_ops.py
# usage: bash scripts/variant_src.sh stream_filter.hip "-DBSK_FILTER_WAVES=5" grep,locate [scale]
cd "$(dirname "$0")/.."
SRC=$1; FLAGS=$2; OPS=${3:-grep}; SCALE=${4:-1.0}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $FLAGS -c bigseqkit_amd/csrc/$SRC -o bigseqkit_amd/lib/$SRC.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
echo "== $SRC $FLAGS"; python scripts/bench_ops.py $SCALE 3 $OPS 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print('   %-48s %8.2f ms  frac %.3f  %s' % (k[:48], v['ms'], v['frac_of_8TBps'], v['note']))"
