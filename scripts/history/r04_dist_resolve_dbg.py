#!/usr/bin/env python3
"""owner-side resolve of the multi-GPU rmdup at C5 scale: the sort-based grouping against the HBM table (rmdup=table) on the
same tuples; prints the tuples on which the keep bytes differ"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bigseqkit_amd as bsk
from bigseqkit_amd import _lib, dist as bdist
from bigseqkit_amd._lib import lib, check

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
n = int(gb * 1e9) // 317 * 317
t = torch.empty(n, dtype=torch.uint8, device="cuda")
check(lib.bsk_synth_device(0, 42, _lib.SYNTH_FLAG_DUPS, 0, C.c_void_p(t.data_ptr()), n, 0, None))
torch.cuda.synchronize()
b = bdist.HipRmDupBackend(json.dumps({"BySeq": True}), 0)
nrec = b.keys(t, bsk.FORMAT_FASTQ)
send, counts = b.pack(0, 1)
k_sort = b.resolve(send).clone()
k_sort2 = b.resolve(send).clone()
check(lib.bsk_ctx_set(b.op.ctx, b"rmdup", b"table"), b.op.ctx)
k_tab = b.resolve(send).clone()
torch.cuda.synchronize()
print("records", nrec, "kept sort", int(k_sort.sum()), "again", int(k_sort2.sum()), "table", int(k_tab.sum()), "expected", nrec - nrec // 5)
d = (k_sort != k_tab).nonzero().flatten()
print("differ", d.numel(), "run-to-run differ", int((k_sort != k_sort2).sum()))
for p in d[:12].tolist():
    k1, k2, g = [int(x) for x in send[p].tolist()]
    same = ((send[:, 0] == k1)).nonzero().flatten().tolist()
    print("pos", p, "gidx", g, "keep sort/table", int(k_sort[p]), int(k_tab[p]), "group:", [(q, int(send[q, 2]), int(send[q, 1]) == k2, int(k_sort[q]), int(k_tab[q])) for q in same])
b.close()
