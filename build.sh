#!/usr/bin/env bash
# Build libbsk.so (HIP kernels + C ABI, gfx950 only) in-tree, plus the command line and the CPU oracle.
# What is rebuilt is decided by CONTENT, not by time stamps (the objects travel to the GPU box
# prebuilt, and an object older than a file that `git checkout` put back would ship silently): every object carries the
# SHA-256 of its source, of every header of the library and of the compile command (<object>.key); the library and the
# command line carry the hash of what they were linked from.  `BSK_BUILD_FORCE=1` rebuilds everything.
set -euo pipefail
cd "$(dirname "$0")"
SRC=bigseqkit_amd/csrc
OUT=bigseqkit_amd/lib
mkdir -p "$OUT" bigseqkit_amd/bin
exec 9>"$OUT/.build.lock"   # (pytest-xdist workers and the driver may arrive together: one build at a time, the others verify)
flock 9
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CXX=${CXX:-g++}
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
HIPFLAGS="--offload-arch=gfx950 $FLAGS"
CXXFLAGS="$FLAGS -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
hash_of() { sha256sum "$@" | sha256sum | cut -d' ' -f1; }
HDR=$(hash_of $(ls $SRC/*.hpp $SRC/*.inc include/*.h | LC_ALL=C sort))
OBJS=()
KEYS=""
for f in $(ls $SRC/*.hip $SRC/*.cpp | LC_ALL=C sort); do
  o="$OUT/$(basename "$f").o"
  if [[ "$f" == *.hip ]]; then cmd="$HIPCC $HIPFLAGS"; else cmd="$CXX $CXXFLAGS"; fi
  key="$(sha256sum "$f" | cut -d' ' -f1) $HDR $(printf '%s' "$cmd" | sha256sum | cut -d' ' -f1)"
  if [ -n "${BSK_BUILD_FORCE:-}" ] || [ ! -f "$o" ] || [ ! -f "$o.key" ] || [ "$(cat "$o.key")" != "$key" ]; then
    rm -f "$o.key"
    $cmd -c "$f" -o "$o"
    printf '%s' "$key" > "$o.key"
  fi
  OBJS+=("$o")
  KEYS+="$key;"
done
LIBKEY=$(printf '%s' "$KEYS" | sha256sum | cut -d' ' -f1)
if [ ! -f "$OUT/libbsk.so" ] || [ ! -f "$OUT/libbsk.so.key" ] || [ "$(cat "$OUT/libbsk.so.key")" != "$LIBKEY" ]; then
  rm -f "$OUT/libbsk.so.key"
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbsk.so" "${OBJS[@]}" -ldl -lpthread
  printf '%s' "$LIBKEY" > "$OUT/libbsk.so.key"
fi
CLIKEY="$(hash_of cli/bigseqkit.cpp $SRC/json.hpp include/bsk.h) $LIBKEY"
if [ ! -f bigseqkit_amd/bin/bigseqkit ] || [ ! -f bigseqkit_amd/bin/bigseqkit.key ] || [ "$(cat bigseqkit_amd/bin/bigseqkit.key)" != "$CLIKEY" ]; then
  rm -f bigseqkit_amd/bin/bigseqkit.key
  $CXX $FLAGS -pthread -o bigseqkit_amd/bin/bigseqkit cli/bigseqkit.cpp -L"$OUT" -lbsk -Wl,-rpath,'$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib
  printf '%s' "$CLIKEY" > bigseqkit_amd/bin/bigseqkit.key
fi
# the Go shim's twin in plain C (tests/host_c/ranks.c: bsk_comm_init_all + one pthread per rank through include/bsk.h alone) --
# built here so that it travels to the GPU box with the library, and so that every build checks that bsk.h is C, not C++
HCKEY="$(hash_of tests/host_c/ranks.c include/bsk.h) $LIBKEY"
if [ ! -f bigseqkit_amd/bin/host_c_ranks ] || [ ! -f bigseqkit_amd/bin/host_c_ranks.key ] || [ "$(cat bigseqkit_amd/bin/host_c_ranks.key)" != "$HCKEY" ]; then
  rm -f bigseqkit_amd/bin/host_c_ranks.key
  gcc -O2 -Wall -std=c11 -Iinclude -pthread -o bigseqkit_amd/bin/host_c_ranks tests/host_c/ranks.c -L"$OUT" -lbsk -Wl,-rpath,'$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib
  printf '%s' "$HCKEY" > bigseqkit_amd/bin/host_c_ranks.key
fi
make -s -C oracle
echo "built $OUT/libbsk.so and oracle/_build/liboracle.so"
