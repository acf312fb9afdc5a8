#!/usr/bin/env bash
# round 3, first GPU visit: the driver-style bench line with the new "ops" object, then the prefetch experiment on the
# streaming sinks that store after every tile (k_index, k_names)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_bench_gpu.py -q -x -k "ops_object or single_gpu" 2>&1 | tail -5) > $O/r03a_tests.log 2>&1
(timeout 1500 python bench.py 2>$O/r03a_bench.err | tail -1) > $O/r03a_bench.json
python scripts/bench_ops.py 1 3 seq,rmdup,subseq 2>&1 | tail -1 > $O/r03a_ops_base.json
bash scripts/variant_src.sh stream_index.hip "-DBSK_PREFETCH=1" rmdup,subseq > $O/r03a_var_index_prefetch.txt 2>&1
bash scripts/variant_src.sh stream_index.hip "-DBSK_PREFETCH=1 -DBSK_INDEX_WAVES=6" rmdup,subseq > $O/r03a_var_index_prefetch_w6.txt 2>&1
bash scripts/variant_src.sh stream_index.hip "" rmdup > /dev/null 2>&1
bash scripts/variant_src.sh stream_names.hip "-DBSK_PREFETCH=1" seq > $O/r03a_var_names_prefetch.txt 2>&1
cat $O/r03a_tests.log; head -c 600 $O/r03a_bench.json; echo; tail -3 $O/r03a_bench.err; cat $O/r03a_var_*.txt
