#!/usr/bin/env bash
# translate: k_translate_wide with the window as asm loads + counted vmcnt (default) against the compiler's loads (-DBSK_TRW_ASM=0)
cd "$(dirname "$0")/.."
python -m pytest tests/test_translate_wide_gpu.py tests/test_translate_light_gpu.py tests/test_translate_rmdup_gpu.py tests/test_golden_gpu.py -m gpu -q -x 2>&1 | tail -4
echo "== asm"; python scripts/bench_ops.py 1 3 translate 2>&1 | tail -1
bash scripts/prof_ops.sh translate 1.0 2>&1 | grep "k_translate\|k_fasta"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -DBSK_TRW_ASM=0 -c bigseqkit_amd/csrc/ops_translate.hip -o bigseqkit_amd/lib/ops_translate.hip.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
echo "== compiler loads"; python scripts/bench_ops.py 1 3 translate 2>&1 | tail -1
bash scripts/prof_ops.sh translate 1.0 2>&1 | grep "k_translate\|k_fasta"
