#!/usr/bin/env bash
# FASTA stats / index: 7 waves per SIMD with 2 spilled VGPRs against 6 without
cd $GRAFT_REPO_ROOT
for w in 7 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -DBSK_STATS_WAVES_FASTA=$w -c bigseqkit_amd/csrc/stream_stats.hip -o bigseqkit_amd/lib/stream_stats.hip.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
  echo "== FASTA waves $w"; python scripts/bench_stats_fasta.py 2>&1 | tail -1
done
