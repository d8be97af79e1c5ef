import importlib, os, sys
import torch
sys.path.insert(0, "/root/repo")
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
N = 8
for P in (48, 24):
    x = torch.randn((N, P, P, P, 64), device="cuda"); w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.02
    res = torch.randn_like(x); wf, wd = ops.pack_conv64_weights(w); y = torch.randn_like(x)
    pad = torch.empty((N, P + 2, P + 2, P + 2, 64), device="cuda"); out = torch.empty_like(x)
    for mt in (0, 1, 3, 5, 0):
        lib.fdn_debug_set_conv64_mt(mt)
        # layouts 1..6 force the DIRECT kernel for everything; we only want the w-face launch -> time shell part only
        def shell(): ops.conv3d_dgrad_fused(x, wd, pad, out, skip=res, y_prev=y, act=ops.ACT_LEAKY, parts=ops.DGRAD_SHELL)
        for _ in range(3): shell()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): shell()
        e1.record(); torch.cuda.synchronize()
        print("P=%d layout %d: shell part %.1f us" % (P, mt, e0.elapsed_time(e1) / 30 * 1e3))
lib.fdn_debug_set_conv64_mt(0)
