#!/usr/bin/env bash
# round 5: waves per SIMD of the streaming passes once more, variants alternating in ONE visit (gpurun_alt/libbsk_<v>.so built with
# -DBSK_NAMES_WAVES / BSK_FILTER_WAVES / BSK_SUBSEQ_WAVES / BSK_STATS_WAVES_ALL; base = the tree)
cd $GRAFT_REPO_ROOT
ops() { BSK_LIB=$PWD/gpurun_alt/libbsk_$1.so BSK_BENCH_PROFILE=1 python scripts/bench_ops.py 1 5 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); v=list(d.values())[0]; n=v['note']; k=json.loads(n[n.index('{'):]) if '{' in n else {}
print('$1', '$2', v['ms'], {a:b for a,b in k.items() if b > 0.3})"; }
for rep in 1 2; do
  for v in base nw6 nw8; do ops $v seq; done
  for v in base fw3 fw5; do ops $v grep; done
  for v in base sw5 sw6; do ops $v subseq; done
  for v in base aw4 aw6; do echo -n "$v stats -a: "; BSK_LIB=$PWD/gpurun_alt/libbsk_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ops --no-scaling-model 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['stats_all']['k_stats_avg_launch_ms'], d['stats_all']['verified'])"; done
done
