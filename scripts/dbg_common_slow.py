import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BSK_MIN_RANGE_BYTES"] = "4096"
import bigseqkit_amd as bsk
import torch
from test_common_gpu import make, _Opts, dev
which = int(sys.argv[1])
fastq, nfiles = True, 2
rng = random.Random(40 + nfiles + fastq)
pool = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 150))) for _ in range(200)]
files = [make(rng, 500, fastq, pool) for _ in range(nfiles)]
t0 = time.time(); frames = [bsk.SeqFrame(bsk.FORMAT_FASTQ, [dev(f)]) for f in files]; torch.cuda.synchronize(); print("to device", round(time.time() - t0, 2), flush=True)
opts = ({}, {"ByName": True}, {"BySeq": True}, {"IgnoreCase": True}, {"BySeq": True, "IgnoreCase": True},
        {"ByName": True, "IgnoreCase": True, "Config": {"LineWidth": 20}})
o = opts[which]
for rep in range(2):
    t0 = time.time(); got = bsk.Common(frames[0], frames[1], _Opts(o)); print(which, o, "rep", rep, len(got), round(time.time() - t0, 2), "s", flush=True)
