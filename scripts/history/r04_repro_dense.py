import sys, os, random, json, ctypes as C
os.environ["BSK_MIN_RANGE_BYTES"] = sys.argv[1] if len(sys.argv) > 1 else "4096"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
import oracle
ALL = len(sys.argv) > 2
def run(fq, tag):
    t = torch.frombuffer(bytearray(fq), dtype=torch.uint8).cuda()
    try:
        got = bsk.StatsString("x", "N/A", bsk.SeqFrame(1, [t]), bsk.SeqKitStatsOptions().Tabular(True).All(ALL)).splitlines()[1]
        want = oracle.stats_string(fq, True, json.dumps({"Tabular": True, "All": ALL}), name="x").splitlines()[1]
        print(tag, "ok" if got == want else ("DIFF " + got + " | " + want))
    except Exception as e:
        print(tag, "EXC", str(e)[:80])
rng = random.Random(6)
ids = [f"r{rng.randrange(300)}" for _ in range(2000)]
orig = [f"@{i} n{k}\nACGT\n+\nIIII\n" for k, i in enumerate(ids)]
run("".join(orig).encode(), "orig 2000")
for n in (100, 150, 190, 200, 250, 400, 800, 1200):
    run("".join(orig[:n]).encode(), "orig first %d (%d bytes)" % (n, len("".join(orig[:n]))))
run("".join(r.replace(" ", "_") for r in orig).encode(), "no space")
run("".join(f"@{i:>4} n{k:>5}\nACGT\n+\nIIII\n" for k, i in enumerate(ids)).encode(), "fixed width")
run("".join(f"@r{k % 7 * 'x'}\nACGT\n+\nIIII\n" for k in range(2000)).encode(), "varying 0..6")

t = torch.frombuffer(bytearray("".join(orig).encode()), dtype=torch.uint8).cuda()
fq = "".join(orig).encode()
for name, fn, opts in (("SeqTransform", lib.bsk_seq_run, {"Reverse": True}), ("SeqTransform", lib.bsk_seq_run, {"Name": True})):
    with bsk.Operator(name, json.dumps(opts), 0) as op:
        out = _lib.Out()
        rc = fn(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, 1, 0, None, C.byref(out))
        print(name, opts, "rc", rc, lib.bsk_last_error(op.ctx) if rc else "", out.len, out.records)
        if rc == 0:
            buf = C.create_string_buffer(max(1, out.len)); check(lib.bsk_out_to_host(op.ctx, C.byref(out), buf, out.len), op.ctx)
            print("  equal oracle:", buf.raw[:out.len] == oracle.seq(fq, True, json.dumps(opts)))
