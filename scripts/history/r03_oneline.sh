#!/usr/bin/env bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1200 python -m pytest tests/test_oneline_fasta_gpu.py tests/test_stats_gpu.py tests/test_seq_gpu.py tests/test_faidx_gpu.py tests/test_round2_gaps_gpu.py -q -x 2>&1 | tail -12)
ls scripts/bench_shapes.py && (python scripts/bench_shapes.py chrom1line 2 2>&1 | tail -30) | tee $O/r03_shapes_oneline.txt
