import faulthandler, sys, time, os
faulthandler.dump_traceback_later(30, repeat=True, file=sys.stderr)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
def T(): return time.time()
if mode == "bsk_first":       # libbsk loaded AND the HIP runtime initialised (device count) before torch comes
    from bigseqkit_amd import lib
    n = lib.bsk_device_count(); t1 = T()
    import torch; t2 = T()
    x = torch.zeros(4000, dtype=torch.uint8).cuda(); torch.cuda.synchronize(); t3 = T()
    print("bsk_first      : bsk+count %.2f import torch %.2f first cuda %.2f" % (t1 - t0, t2 - t1, t3 - t2), flush=True)
elif mode == "bsk_nocount":   # libbsk loaded, no HIP call before torch
    from bigseqkit_amd import lib; t1 = T()
    import torch; t2 = T()
    x = torch.zeros(4000, dtype=torch.uint8).cuda(); torch.cuda.synchronize(); t3 = T()
    n = lib.bsk_device_count(); t4 = T()
    print("bsk_nocount    : bsk %.2f import torch %.2f first cuda %.2f count %.2f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3), flush=True)
else:                         # torch first
    import torch; t1 = T()
    from bigseqkit_amd import lib; t2 = T()
    n = lib.bsk_device_count(); t3 = T()
    x = torch.zeros(4000, dtype=torch.uint8).cuda(); torch.cuda.synchronize(); t4 = T()
    print("torch_first    : import torch %.2f bsk %.2f count %.2f first cuda %.2f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3), flush=True)
