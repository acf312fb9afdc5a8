#!/usr/bin/env bash
# round 5, visit 5: kernel times of a FASTA stats step (rocprofv3 kernel trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r05e -o st -- python $R/scripts/bench_stats_fasta.py) > $O/prof_r05e.log 2>&1
tail -2 $O/prof_r05e.log | cut -c1-600
for f in $(find $O/prof_r05e -name '*kernel_stats.csv'); do head -12 $f; done
