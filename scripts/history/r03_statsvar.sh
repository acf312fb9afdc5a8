#!/usr/bin/env bash
# bash scripts/r03_statsvar.sh "<defs>" ... : rebuild stream_stats.hip with the defs and print the bench's stats numbers
cd "$(dirname "$0")/.."
for D in "$@"; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $D -c bigseqkit_amd/csrc/stream_stats.hip -o bigseqkit_amd/lib/stream_stats.hip.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
echo "== $D"; python bench.py --no-cpu-baseline --no-ops --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['bit_exact_vs_expected_row'], 'stats -a', d['stats_all']['k_stats_avg_launch_ms'], d['stats_all']['verified'])"
done
