#!/usr/bin/env bash
# experiment: rebuild ops_translate.hip with extra -D flags on the GPU box and time translate at C4 (half size)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $1 -c bigseqkit_amd/csrc/ops_translate.hip -o bigseqkit_amd/lib/ops_translate.hip.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
echo "== $1"; bash scripts/prof_ops.sh translate 0.5 2>&1 | grep "k_translate_frames4"
