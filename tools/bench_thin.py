"""Thin-layer kernels at the cfg2 shapes (3 -> 64 forward / wgrad at (8,24^3)): python tools/bench_thin.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
N, P = 8, 24
x3 = torch.randn((N, P, P, P, 3), device="cuda"); w3 = torch.randn((3, 3, 3, 3, 64), device="cuda") * 0.1
b = torch.randn(64, device="cuda"); dz = torch.randn((N, P, P, P, 64), device="cuda"); y = torch.empty_like(dz)
ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 3, 64, 3) // 4 + 1, device="cuda"); dw = torch.empty_like(w3)
for mf in (1, 0, 1):
    lib.fdn_debug_set_cin3_mfma(mf)
    print("cin3 %s: fwd %.1f us, wgrad(+reduce) %.1f us" % ("mfma" if mf else "valu",
          timeit(lambda: ops.conv3d_fwd(x3, w3, b, ops.ACT_RELU, out=y)), timeit(lambda: ops.conv3d_wgrad(x3, dz, 3, 3, 64, dw=dw, workspace=ws))))
