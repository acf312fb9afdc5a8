#!/usr/bin/env bash
# Build libbsk.so (HIP kernels + C ABI, gfx950 only) in-tree, plus the CPU oracle.
set -euo pipefail
cd "$(dirname "$0")"
SRC=bigseqkit_amd/csrc
OUT=bigseqkit_amd/lib
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
OBJS=()
for f in $SRC/*.hip $SRC/*.cpp; do
  o="$OUT/$(basename "$f").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find $SRC include -newer "$o" \( -name '*.hpp' -o -name '*.h' \) -print -quit)" ]; then
    if [[ "$f" == *.hip ]]; then $HIPCC --offload-arch=gfx950 $FLAGS -c "$f" -o "$o"; else ${CXX:-g++} $FLAGS -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$f" -o "$o"; fi
  fi
  OBJS+=("$o")
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbsk.so" "${OBJS[@]}"
mkdir -p bigseqkit_amd/bin
${CXX:-g++} $FLAGS -o bigseqkit_amd/bin/bigseqkit cli/bigseqkit.cpp -L"$OUT" -lbsk -Wl,-rpath,'$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib
make -s -C oracle
echo "built $OUT/libbsk.so and oracle/_build/liboracle.so"
