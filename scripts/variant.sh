#!/usr/bin/env bash
# build libbsk.so with extra -D flags for stream_stats.hip and print the bench roofline numbers
# usage: bash scripts/variant.sh "-DBSK_NPIECE=2 -DBSK_PREFETCH=0"
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $1 -c bigseqkit_amd/csrc/stream_stats.hip -o bigseqkit_amd/lib/stream_stats.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o
