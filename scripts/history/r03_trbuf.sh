#!/usr/bin/env bash
# translate: tests, time and SQ counters of the current build
cd "$(dirname "$0")/.."
python -m pytest tests/test_translate_wide_gpu.py tests/test_translate_light_gpu.py tests/test_translate_rmdup_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x 2>&1 | tail -4
python scripts/bench_ops.py 1 3 translate 2>&1 | tail -1
bash scripts/prof_ops.sh translate 1.0 2>&1 | grep "k_translate\|k_fasta"
bash scripts/pmc_sq_ops.sh translate 0.25 trsq > /dev/null 2>&1
python - <<PY
import csv, collections, glob, re
out={}
for f in sorted(glob.glob("gpurun_out/pmc_trsq_*/pmc_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name=r["Kernel_Name"].replace("void ","").replace("bsk::(anonymous namespace)::","")
        name=re.split(r"\(", name)[0]
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,cn),v in sorted(agg.items()):
        out.setdefault(k,{})[cn]=sum(v)/len(v)
for k,v in out.items():
    if 'translate_wide' in k:
        w=v["SQ_WAVES"]
        print(k,'waves',w,'valu/wave %.0f salu/wave %.0f lds/wave %.0f vmem_rd/wave %.1f vmem_wr/wave %.1f'%(v["SQ_INSTS_VALU"]/w, v["SQ_INSTS_SALU"]/w, v["SQ_INSTS_LDS"]/w, v["SQ_INSTS_VMEM_RD"]/w, v["SQ_INSTS_VMEM_WR"]/w))
PY
