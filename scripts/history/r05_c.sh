#!/usr/bin/env bash
# round 5, visit 3: the FASTA pass that publishes two kinds of newlines (stream_fasta2_dev.hpp) against the one that publishes all
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_stats_fasta2_gpu.py tests/test_stats_gpu.py tests/test_oneline_fasta_gpu.py tests/test_round2_gaps_gpu.py tests/test_golden_gpu.py -q -x 2>&1 | tail -15) > $O/r05c_tests.log 2>&1
cat $O/r05c_tests.log
echo "== every newline an event (round 4)"; BSK_STATS_FASTA=events python scripts/bench_stats_fasta.py 2>&1 | tail -12
echo "== two kinds of events"; python scripts/bench_stats_fasta.py 2>&1 | tail -12
