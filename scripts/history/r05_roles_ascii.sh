#!/usr/bin/env bash
# RESULT: no difference (k_stats -a 21.0-21.2 ms either way, three alternating runs) -- the source change is not in the tree:
# for a tile without a byte above 0x7F, ((v & 0x7F..) + k | v) & 0x80.. became (v + k) & 0x80.. in both role-count paths
# (about 35 VALU of 575 per tile).  The pass is not bound by its VALU count.
# round 5: `stats -a` on FASTQ with / without the plain-sum form of "byte >= t" for tiles without a byte above 0x7F (one visit)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_stats_gpu.py tests/test_crlf_gpu.py -q -x -m gpu 2>&1 | tail -2
for rep in 1 2 3; do for v in a0 a1; do echo -n "$v: "; BSK_LIB=$PWD/gpurun_alt/libbsk_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ops --no-scaling-model 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stats_all']['ms_per_step'], d['stats_all']['k_stats_avg_launch_ms'], d['stats_all']['verified'])"; done; done
