#!/usr/bin/env bash
# round 5: instruction budgets of the streaming passes (grouping key instead of XXH64 in k_rmdup_stream_g; the rest as in round 4) -- SQ
# counters per kernel (VALU / SALU / LDS / VMEM instructions, busy and wait cycles) for the commands at full scale, and for
# `stats` / `stats -a` through bench.py; per 4 KiB tile = counter / (bytes of the pass / 4096).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
bash scripts/pmc_sq_ops.sh seq,subseq,grep,rmdup,translate 1.0 r05sq > $O/r05sq.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_r05sqstats_a -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ops > $O/pmc_r05sqstats_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_r05sqstats_b -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ops > $O/pmc_r05sqstats_b.log 2>&1
cd $R
python - <<PY
import csv, collections, glob, json
BYTES = {"k_stats<true, false": 99999999921, "k_stats<true, true": 99999999921, "k_names": 99999999921, "k_subseq_stream": 24999999901,
         "k_filter": 12499999886, "k_rmdup_stream": 24999999901, "k_translate_uniform": 49999997088, "k_translate_wide": 49999997088, "k_seg_copy": 24999999901, "k_rmdup_place": 24999999901}
out = {}
for f in sorted(glob.glob("$O/pmc_r05sq_*/pmc_counter_collection.csv") + glob.glob("$O/pmc_r05sqstats_*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("bsk::", "")
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, cn), v in sorted(agg.items()):
        out.setdefault(k, {})[cn] = sum(v) / len(v)
res = {"source": "scripts/r05_sq_budgets.sh: rocprofv3 --pmc (two passes of 8 SQ counters), means per dispatch; per_tile = counter / (bytes of the pass / 4096)", "kernels": {}}
for k, v in out.items():
    key = next((b for b in BYTES if k.startswith(b)), None)
    if key is None:
        continue
    tiles = BYTES[key] / 4096.0
    e = {a: round(b, 1) for a, b in v.items()}
    e["per_tile"] = {a.replace("SQ_INSTS_", "").lower(): round(v[a] / tiles, 1) for a in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if a in v}
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_BUSY_CYCLES" in v and v.get("SQ_WAVE_CYCLES"):
        e["valu_active_share_of_wave_cycles"] = round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 3)
    if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
        e["wait_any_share_of_wave_cycles"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3)
    res["kernels"][k] = e
    print(k, e["per_tile"], e.get("valu_active_share_of_wave_cycles"), e.get("wait_any_share_of_wave_cycles"))
json.dump(res, open("$O/r05_sq_budgets.json", "w"), indent=1)
PY
