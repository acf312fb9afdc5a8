#!/usr/bin/env bash
# Evidence for the hot-path operators at the BASELINE sizes (one GPU visit):
#   bash scripts/ops_evidence.sh r02 [ops]
# 1. scripts/bench_ops.py           -> times, algorithmic GB/s, fraction of the 8 TB/s roof
# 2. rocprofv3 --kernel-trace --stats of the same command -> profiles/<tag>_ops_kernel_stats.csv
# 3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes)  -> HBM bytes per command (gfx950 read correction)
# merged into profiles/<tag>_ops.json: one "roofline" object per command (achieved, frac, traffic, traffic / algorithmic).
TAG=${1:-r02}; OPS=${2:-seq,subseq,grep,locate,rmdup,translate}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O $R/profiles
cd $R
python scripts/bench_ops.py 1 3 $OPS > $O/ops_$TAG.json 2> $O/ops_$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ops_$TAG -o ops -- python $R/scripts/bench_ops.py 1 1 $OPS > $O/prof_ops_$TAG.log 2>&1
cd $R
for op in $(echo $OPS | tr ',' ' '); do
  bash scripts/pmc_ops_traffic.sh $op 1.0 $O/traffic_${TAG}_$op.json > $O/traffic_${TAG}_$op.txt 2>&1
done
python scripts/ops_evidence_merge.py $TAG $OPS   # (run it again on the build box: profiles/ does not travel back)
