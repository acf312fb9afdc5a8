#!/usr/bin/env bash
# where do k_translate_wide's vector instructions go?  variants -DBSK_TRW_EXP=n, SQ counters each
cd "$(dirname "$0")/.."
for v in 0 1 2 3; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -DBSK_TRW_EXP=$v -c bigseqkit_amd/csrc/ops_translate.hip -o bigseqkit_amd/lib/ops_translate.hip.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
bash scripts/pmc_sq_ops.sh translate 0.25 trexp$v > /dev/null 2>&1
python - <<PY
import csv, collections, glob, re
out={}
for f in sorted(glob.glob("gpurun_out/pmc_trexp${v}_*/pmc_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name=r["Kernel_Name"].replace("void ","").replace("bsk::(anonymous namespace)::","")
        name=re.split(r"\(", name)[0]
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        if r["Counter_Name"]=="SQ_WAVES": out.setdefault(name,{}).setdefault("ns",[]).append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
    for (k,cn),vv in sorted(agg.items()):
        out.setdefault(k,{})[cn]=sum(vv)/len(vv)
for k,vv in out.items():
    if 'translate_wide' in k:
        w=vv["SQ_WAVES"]
        print("EXP $v",k,'valu/wave %.0f salu/wave %.0f lds/wave %.0f vmem_rd %.1f vmem_wr %.1f  ms(0.25 scale, with counters) %.2f'%(vv["SQ_INSTS_VALU"]/w, vv["SQ_INSTS_SALU"]/w, vv["SQ_INSTS_LDS"]/w, vv["SQ_INSTS_VMEM_RD"]/w, vv["SQ_INSTS_VMEM_WR"]/w, sum(vv["ns"])/len(vv["ns"])/1e6))
PY
done
