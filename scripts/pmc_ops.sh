#!/usr/bin/env bash
# SQ instruction mix of the kernels of one bench_ops.py command:  bash scripts/pmc_ops.sh translate 0.1 [kernel-substring]
OP=${1:-translate}; SCALE=${2:-0.1}; KSUB=${3:-k_}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmcop_${OP}_a -o pmc -- python $R/scripts/bench_ops.py $SCALE 1 $OP > $O/pmcop_${OP}_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmcop_${OP}_b -o pmc -- python $R/scripts/bench_ops.py $SCALE 1 $OP > $O/pmcop_${OP}_b.log 2>&1
cd $R
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/pmcop_${OP}_*/pmc_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Kernel_Name"] and "k_synth" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(agg): print(k, "%.4g" % (sum(agg[k])/len(agg[k])))
PY
