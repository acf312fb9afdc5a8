#!/usr/bin/env bash
# round 6, A/B 6: k_subseq_stream with a sixth wave per SIMD (a 256-byte carry in front of the LDS tile: 26.1 KB per block,
# 78 registers) against the 5 waves of round 5 (512-byte carry, the compiler's 81 registers) -- five alternations in one
# visit, both output contracts (the times of one build spread by 5 % from box to box)
cd "$(dirname "$0")/.."
{
for rep in 1 2 3 4 5; do
for v in "-DBSK_TILE_CARRY=512 -DBSK_SUBSEQ_WAVES=0" "-DBSK_TILE_CARRY=256 -DBSK_SUBSEQ_WAVES=6"; do
  BSK_OUT=slices BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_subseq.hip "$v" subseq
  BSK_BENCH_PROFILE=1 bash scripts/variant_src.sh stream_subseq.hip "$v" subseq
done; done
} > gpurun_out/r06_ab6.log 2>&1
grep -A1 "^==" gpurun_out/r06_ab6.log | grep -v "^--" | paste - - | sed 's/.*CARRY=\([0-9]*\).*WAVES=\([0-9]\).*FASTQ) *\([0-9.]* ms\).*k_subseq_stream": \([0-9.]*\).*/carry \1 waves \2: \3  kernel \4/'
