#!/usr/bin/env bash
# experiment: rebuild stream_index.hip with extra -D flags on the GPU box and print the k_index times of translate (FASTA) and seq (FASTQ)
# usage: bash scripts/variant_index.sh "-DBSK_NPIECE=3 -DBSK_EXPERIMENT"
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $1 -c bigseqkit_amd/csrc/stream_index.hip -o bigseqkit_amd/lib/stream_index.hip.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
echo "== $1"
bash scripts/prof_ops.sh translate 0.5 2>&1 | grep "k_index<"
bash scripts/prof_ops.sh grep 1.0 2>&1 | grep "k_index<"
