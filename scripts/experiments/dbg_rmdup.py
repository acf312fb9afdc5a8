import sys, os, json, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, oracle, bigseqkit_amd as bsk
from test_translate_rmdup_gpu import dup_fastq, frame, _Opts
os.environ["BSK_MIN_RANGE_BYTES"] = "4096"
for seed in range(800, 806):
    rng = random.Random(seed)
    data = dup_fastq(rng, 1500)
    want = oracle.rmdup(data, True, json.dumps({"BySeq": True}))
    for mode in ("buckets", "table"):
        if mode == "table": os.environ["BSK_RMDUP"] = "table"
        else: os.environ.pop("BSK_RMDUP", None)
        for rep in range(3):
            try:
                got = bsk.RmDup(frame(data, True), _Opts({"BySeq": True}))
                print(seed, mode, rep, got == want, len(got), len(want))
            except Exception as e:
                print(seed, mode, rep, "ERR", str(e)[:80])
