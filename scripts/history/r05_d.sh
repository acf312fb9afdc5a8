#!/usr/bin/env bash
# round 5, visit 4: variants of the FASTA pass (non-temporal tile loads, waves per SIMD)
cd $GRAFT_REPO_ROOT
run() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $1 -c bigseqkit_amd/csrc/stream_stats.hip -o bigseqkit_amd/lib/stream_stats.hip.o 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
  echo "== $1"; python scripts/bench_stats_fasta.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if not k.endswith('-a'): print('   %-32s %8.3f ms  frac %.3f' % (k, v['ms'], v['frac_of_8TBps']))"; }
run ""
run "-DBSK_F2_NT=1"
run "-DBSK_F2_NT=1 -DBSK_STATS_WAVES=6"
run "-DBSK_STATS_WAVES=8"
run "-DBSK_F2_NT=1 -DBSK_STATS_WAVES=8"
