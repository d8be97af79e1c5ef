"""Ablations of wgrad64_wino_kernel (test build): python tools/abl_wgrad_wino.py [N] [P]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for P in ([int(sys.argv[2])] if len(sys.argv) > 2 else [48, 24]):
    x = torch.randn((N, P, P, P, 64), device="cuda"); dz = torch.randn_like(x)
    ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1, device="cuda"); dw = torch.empty((3, 3, 3, 64, 64), device="cuda")
    flop = 2.0 * 27 * 64 * 64 * N * P ** 3
    for bits, name in ((0, "full (warm-up)"), (0, "full"), (1, "no raw loads"), (2, "no transform / LDS writes"), (3, "no loads, no transform / writes"),
                       (4, "no LDS operand reads"), (7, "MFMAs + barrier only"), (39, "MFMAs + barrier, no tile walk"), (39 + 64, "MFMAs only, no barrier, no walk"), (64, "full, no barrier (wrong results)"), (32, "full, no tile walk"), (0, "full")):
        lib.fdn_debug_set_wgrad64_wino_dbg(bits)
        for _ in range(3): ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("P=%d %-36s %7.3f ms (incl. reduce)  matrix pipe %.3f" % (P, name, ms, flop / 3 / ms * 1e-9 / 157.3))
    lib.fdn_debug_set_wgrad64_wino_dbg(0)
