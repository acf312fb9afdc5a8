#!/usr/bin/env bash
# round 5 evidence (GPU box): the FETCH_SIZE calibration, bench_ops.py timings + PMC passes per command, merged with a
# factor per access shape -> gpurun_out/ (profiles/<tag>_ops_traffic.json is written by the merge step; run the merge again
# on the build box: profiles/ does not travel back)
TAG=${1:-r05}; OPS=${2:-seq,subseq,grep,rmdup,translate}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd $R
bash scripts/fetch_calibration.sh $TAG | tail -5
cp $O/${TAG}_fetch_calibration.json profiles/${TAG}_fetch_calibration.json
python scripts/bench_ops.py 1 3 $OPS > $O/ops_$TAG.json 2> $O/ops_$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ops_$TAG -o ops -- python $R/scripts/bench_ops.py 1 1 $OPS > $O/prof_ops_$TAG.log 2>&1
cd $R
for op in $(echo $OPS | tr ',' ' '); do
  bash scripts/pmc_ops_traffic.sh $op 1.0 $O/traffic_${TAG}_$op.json > $O/traffic_${TAG}_$op.txt 2>&1
done
python scripts/ops_traffic_merge.py $TAG $OPS
cp profiles/${TAG}_ops_traffic.json $O/
