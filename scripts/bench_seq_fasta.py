"""`seq` on FASTA (10 GB of 5 kb / 1 kb records) in the modes that leave the verbatim path: unwrap, rewrap, reverse complement, names, bases, case."""
import ctypes as C, json, sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib
from bigseqkit_amd._lib import lib, check
def synth(kind, nbytes):
    rb = lib.bsk_synth_record_bytes(kind); n = int(nbytes)//rb*rb
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    check(lib.bsk_synth_device(kind, 42, 0, 0, C.c_void_p(t.data_ptr()), n, 0, None)); torch.cuda.synchronize(); return t, n//rb
def run(opts, t, fmt, reps=3):
    out = _lib.Out()
    with bsk.Operator("SeqTransform", json.dumps(opts), 0) as op:
        check(lib.bsk_seq_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, fmt, 0, None, C.byref(out)), op.ctx); torch.cuda.synchronize()
        t0=time.perf_counter()
        for _ in range(reps):
            check(lib.bsk_seq_run(op.ctx, C.c_void_p(t.data_ptr()), t.numel(), 1, fmt, 0, None, C.byref(out)), op.ctx); torch.cuda.synchronize()
        return (time.perf_counter()-t0)/reps*1e3, out.len
for kind,name,gb in ((2,"FASTA-5k",10e9),(1,"FASTA-1k",10e9)):
    t,n = synth(kind, gb)
    for opts in ({}, {"Config":{"LineWidth":0}}, {"Config":{"LineWidth":70}}, {"Reverse":True,"Complement":True}, {"Name":True}, {"Seq":True}, {"UpperCase":True}):
        ms,ol = run(opts, t, 0)  # BSK_FORMAT_FASTA
        print("%-9s %-45s %8.2f ms  out %.2f GB  %.0f GB/s" % (name, json.dumps(opts), ms, ol/1e9, (t.numel()+ol)/ms/1e6))
    del t
