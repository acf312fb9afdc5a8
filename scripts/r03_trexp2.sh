#!/usr/bin/env bash
# k_translate_wide: what do the 17th-byte stores, the header stores and the misalignment of the blocks cost?  (-DBSK_TRW_EXP=4/5/6: wrong output, timing only)
cd "$(dirname "$0")/.."
for v in 0 4 5 6; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -DBSK_TRW_EXP=$v -c bigseqkit_amd/csrc/ops_translate.hip -o bigseqkit_amd/lib/ops_translate.hip.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
echo "== EXP $v"; bash scripts/prof_ops.sh translate 1.0 2>&1 | grep "k_translate_wide"
done
