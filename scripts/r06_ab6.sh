#!/usr/bin/env bash
# round 6, A/B 6: k_subseq_stream with a sixth wave per SIMD (a 256-byte carry in front of the LDS tile: 26.1 KB per block,
# 78 registers) -- parity of the subseq tests first, then time at 25 GB, twice each
cd "$(dirname "$0")/.."
{
for rep in 1 2; do
for v in "" "-DBSK_TILE_CARRY=256 -DBSK_SUBSEQ_WAVES=6" "-DBSK_TILE_CARRY=256" "-DBSK_TILE_CARRY=128 -DBSK_SUBSEQ_WAVES=6"; do
  BSK_OUT=slices BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_subseq.hip "$v" subseq
done; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DBSK_TILE_CARRY=256 -DBSK_SUBSEQ_WAVES=6 -c bigseqkit_amd/csrc/stream_subseq.hip -o bigseqkit_amd/lib/stream_subseq.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o
timeout 900 python -m pytest tests/test_grep_subseq_gpu.py tests/test_fuzz_gpu.py tests/test_out_slices_gpu.py tests/test_subseq_stream_gpu.py -q -m gpu 2>&1 | tail -3
} > gpurun_out/r06_ab6.log 2>&1
grep -v "^  File\|^    " gpurun_out/r06_ab6.log | tail -24
