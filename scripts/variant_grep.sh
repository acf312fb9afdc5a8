#!/usr/bin/env bash
# experiment: rebuild ops_grep.hip with extra -D flags on the GPU box and time grep at C3
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $1 -c bigseqkit_amd/csrc/ops_grep.hip -o bigseqkit_amd/lib/ops_grep.hip.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
echo "== $1"; python scripts/bench_ops.py 1.0 3 grep 2>&1 | tail -1 | cut -c1-200
