#!/usr/bin/env bash
# RESULT (source change not in the tree: plan_ranges() in ctx.hpp + a boundary rule in k_prep): 100 GB 15.73 -> 15.59 and 15.61 ->
# 15.55 ms, but the shards of an N-GPU job LOSE: 50 GB 7.96 -> 8.10, 25 GB 4.12 -> 4.16, 12.5 GB 2.18 -> 2.23 ms (quarter ranges of
# 128 KiB pay more start-up than the shorter tail returns); seq -n / subseq 1 % slower.  Dropped.
# round 5: guided ranges (the last eighth of the ranges in quarters) against uniform ones (BSK_RANGES_TAIL=off), one visit
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_stats_gpu.py tests/test_filter_gpu.py tests/test_names_gpu.py tests/test_subseq_stream_gpu.py tests/test_rmdup_keys_gpu.py tests/test_segcopy_gpu.py -q -x -m gpu 2>&1 | tail -2
for rep in 1 2; do for v in on off; do echo -n "tail=$v: "; BSK_RANGES_TAIL=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ops 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); m=d['scaling_model']['per_n_gpus']; print(d['ms_per_step'], d['roofline']['frac'], d['stats_all']['ms_per_step'], [m[k]['ms_per_step'] for k in '1248'], d['bit_exact_vs_expected_row'])"; done; done
for v in on off; do echo "tail=$v:"; BSK_RANGES_TAIL=$v python scripts/bench_ops.py 1 5 seq,subseq,grep,rmdup 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
for k,v in d.items(): print('   ',k[:30], v['ms'])"; done
