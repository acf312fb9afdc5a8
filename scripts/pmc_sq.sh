#!/usr/bin/env bash
# SQ instruction-mix / stall counters for the bench kernels (two PMC passes)
TAG=${1:-sq}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_${TAG}_a -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_${TAG}_b -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_${TAG}_b.log 2>&1
cd $R
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/pmc_${TAG}_*/pmc_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_stats" in r["Kernel_Name"]:
            key=("ALL" if "<true, true" in r["Kernel_Name"] else "DEF", r["Counter_Name"])
            agg[key].append(float(r["Counter_Value"]))
    for k in sorted(agg): print(k, "%.4g" % (sum(agg[k])/len(agg[k])), "vgpr/sgpr see csv")
PY
