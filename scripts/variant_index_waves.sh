#!/usr/bin/env bash
# experiment: k_index with a waves-per-SIMD hint; prints the k_index time of grep (12.5 GB FASTQ) and translate (50 GB FASTA)
cd "$(dirname "$0")/.."
for FL in "" "-DBSK_INDEX_WAVES=7"; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $FL -c bigseqkit_amd/csrc/stream_index.hip -o bigseqkit_amd/lib/stream_index.hip.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
echo "== $FL"
bash scripts/prof_all_ops.sh 1.0 var "grep translate" 2>&1 | grep "k_index<"
done
