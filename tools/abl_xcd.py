import importlib, os, sys
import torch
sys.path.insert(0, "/root/repo")
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
N = 8
for P in (48, 24):
    x = torch.randn((N, P, P, P, 64), device="cuda"); w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.02
    res = torch.randn_like(x)
    wf, wd = ops.pack_conv64_weights(w); y = torch.empty_like(x)
    pad = torch.empty((N, P + 2, P + 2, P + 2, 64), device="cuda"); out = torch.empty_like(x)
    for bits in (256, 0, 256, 0):
        lib.fdn_debug_set_conv64_wino_dbg(bits)
        def fwd(): ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wf, out=y)
        def dgr(): ops.conv3d_dgrad_fused(x, wd, pad, out, skip=res, y_prev=y, act=ops.ACT_LEAKY)
        for name, fn in (("forward", fwd), ("fused dgrad", dgr)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): fn()
            e1.record(); torch.cuda.synchronize()
            print("P=%d %-12s bits %3d (256 = old face tiling) %7.4f ms" % (P, name, bits, e0.elapsed_time(e1) / 30))
