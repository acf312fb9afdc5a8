#!/usr/bin/env bash
# experiment helper (GPU box): rebuild ONE source with extra -D flags and count the instructions of its kernels per 4 KiB tile
# usage: bash scripts/variant_valu.sh stream_names.hip "-DBSK_NAMES_WINDOW=384" seq k_names 100e9
cd "$(dirname "$0")/.."
SRC=$1; FLAGS=$2; OPS=${3:-seq}; KERNEL=${4:-k_names}; BYTES=${5:-100e9}
R=$(pwd); O=$R/gpurun_out/valu_tmp; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $FLAGS -c bigseqkit_amd/csrc/$SRC -o bigseqkit_amd/lib/$SRC.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bigseqkit_amd/lib/libbsk.so bigseqkit_amd/lib/*.o || exit 1
( cd /tmp && export TMPDIR=/tmp && BSK_DIAG=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O -o pmc -- python $R/scripts/bench_ops.py 1 1 $OPS > $O/log 2>&1 )
python - "$O" "$KERNEL" "$BYTES" "$FLAGS" <<'PY'
import csv, glob, sys, collections
O, kern, nbytes, flags = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern + "<" in r["Kernel_Name"] or kern + "(" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = nbytes / 4096
print("== %-60s %s" % (flags, "  ".join("%s %.1f" % (k.replace("SQ_INSTS_", "").lower(), max(v) / tiles) for k, v in sorted(agg.items()))))
PY
