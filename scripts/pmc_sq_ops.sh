#!/usr/bin/env bash
# SQ instruction-mix / stall counters per kernel of one command of scripts/bench_ops.py:  bash scripts/pmc_sq_ops.sh translate 0.25 [tag]
OPS=${1:-translate}; SCALE=${2:-0.25}; TAG=${3:-sqops}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_${TAG}_a -o pmc -- python $R/scripts/bench_ops.py $SCALE 1 $OPS > $O/pmc_${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_${TAG}_b -o pmc -- python $R/scripts/bench_ops.py $SCALE 1 $OPS > $O/pmc_${TAG}_b.log 2>&1
cd $R
python - <<PY
import csv, collections, glob, json
out={}
for f in sorted(glob.glob("$O/pmc_${TAG}_*/pmc_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name=r["Kernel_Name"].split("(")[0].replace("void ","").replace("bsk::(anonymous namespace)::","")
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,cn),v in sorted(agg.items()):
        out.setdefault(k,{})[cn]=sum(v)/len(v)
        out[k]["dispatches"]=len(v)
for k,v in out.items():
    if v.get("SQ_WAVES",0)>0:
        v["valu_per_wave"]=v.get("SQ_INSTS_VALU",0)/v["SQ_WAVES"]
    print(k, json.dumps({a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items()}))
json.dump(out, open("$O/${TAG}_sq.json","w"), indent=1)
PY
