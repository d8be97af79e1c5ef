"""Forward / fused-dgrad time of conv64_wino_kernel for forced main-region tiles (test build): python tools/sweep_wino_tile.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
N = 8
for P in (48, 24):
    x = torch.randn((N, P, P, P, 64), device="cuda"); w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.02
    res = torch.randn_like(x); wf, wd = ops.pack_conv64_weights(w); y = torch.empty_like(x)
    pad = torch.empty((N, P + 2, P + 2, P + 2, 64), device="cuda"); out = torch.empty_like(x)
    for (td, th, tg) in ((0, 0, 0), (8, 8, 1), (4, 8, 2), (8, 4, 2), (4, 4, 4), (2, 8, 4), (8, 2, 4), (4, 16, 1), (16, 4, 1), (6, 8, 1), (8, 6, 1), (4, 8, 1), (8, 8, 1)):
        if tg and (P // 4) % tg: continue
        lib.fdn_debug_set_conv64_wino_tile(td | th << 8 | tg << 16)
        def fwd(): ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wf, out=y)
        def dgr(): ops.conv3d_dgrad_fused(x, wd, pad, out, skip=res, y_prev=y, act=ops.ACT_LEAKY)
        r = []
        for fn in (fwd, dgr):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            r.append(e0.elapsed_time(e1) / 20)
        print("P=%d tile %2dx%2dx%d (0 = planner): forward %.4f ms, fused dgrad %.4f ms" % (P, td, th, tg, r[0], r[1]))
lib.fdn_debug_set_conv64_wino_tile(0)
