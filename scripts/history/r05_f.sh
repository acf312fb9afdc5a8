#!/usr/bin/env bash
# round 5, visit 6: bench.py with the FASTA stats legs and the non-uniform translate leg (small, then the full line)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_bench_gpu.py -q -x 2>&1 | tail -15) > $O/r05f_tests.log 2>&1
cat $O/r05f_tests.log
(timeout 900 python bench.py --no-cpu-baseline 2>$O/r05f_bench.err | tail -1) > $O/r05f_bench.json
python - <<PY
import json
d=json.load(open("$O/r05f_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","bit_exact_vs_expected_row")}, d["roofline"]["frac"])
for k,v in d["ops"].items():
    if isinstance(v,dict) and "ms" in v: print("%-46s %8.3f ms frac %.4f exact %s %s" % (k, v["ms"], v["frac"], v["exact"], json.dumps(v["kernels_ms_per_call"])))
    else: print(k, v)
PY
tail -3 $O/r05f_bench.err
