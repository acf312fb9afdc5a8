O=gpurun_out; mkdir -p $O
R=$(pwd)
(timeout 900 python -m pytest tests -m gpu -q -x -k "stats or fuzz or golden or bench or multirank" 2>&1 | tail -4) > $O/tests_roles3.log 2>&1
(timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_roles3.json
cat $O/tests_roles3.log
python -c "
import json; d=json.load(open('$O/bench_roles3.json')); a=d['stats_all']
print('stats %.3f ms   stats -a %.3f ms (kernel %.3f) verified=%s' % (d['ms_per_step'], a['ms_per_step'], a['k_stats_avg_launch_ms'], a['verified']))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/$O/pmc_roles3_a -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/pmc_roles3_a.log 2>&1
cd $R
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/pmc_roles3_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_stats" in r["Kernel_Name"]:
            key=("ALL" if "<true, true" in r["Kernel_Name"] else "DEF", r["Counter_Name"])
            agg[key].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        if k[0]=="ALL": print(k, "%.4g" % (sum(agg[k])/len(agg[k])), "per tile %.1f" % (sum(agg[k])/len(agg[k])/24.4e6))
PY
