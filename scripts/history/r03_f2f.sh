#!/usr/bin/env bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_store_gpu.py tests/test_cli.py tests/test_golden_gpu.py -q -x 2>&1 | tail -8)
python scripts/bench_file_to_file.py 8 --disk --parts > $O/r03_file_to_file.json 2> $O/r03_file_to_file.err; cat $O/r03_file_to_file.json | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['rows'].items(): print(k, v)"; tail -2 $O/r03_file_to_file.err
