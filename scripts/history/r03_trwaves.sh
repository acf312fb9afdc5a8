#!/usr/bin/env bash
# k_translate_wide at 5 / 6 waves per SIMD (-DBSK_TRW_WAVES=n) against the compiler's choice
cd "$(dirname "$0")/.."
for v in 5 6; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -DBSK_TRW_WAVES=$v -Rpass-analysis=kernel-resource-usage -c bigseqkit_amd/csrc/ops_translate.hip -o bigseqkit_amd/lib/ops_translate.hip.o 2>&1 | grep -A9 "k_translate_wideILi64" | grep "VGPRs\|Spill\|Occupancy\|Scratch" | sed 's/.*remark: //' | tr '\n' ' '; echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
echo "== waves $v"; python scripts/bench_ops.py 1 3 translate 2>&1 | tail -1 | cut -c1-200
bash scripts/prof_ops.sh translate 1.0 2>&1 | grep "k_translate_wide"
done
