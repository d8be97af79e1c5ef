"""Weight-gradient time per 8 patches at 24^3 as the batch grows (what ONE batched launch over the 19 low-res layers could save: 0.11 -> 0.076 ms per layer-equivalent; measured, not built).  python tools/abl_wgrad_batch.py"""
import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
P = 24
for N in (8, 16, 32, 76, 152):
    x = torch.randn(N, P, P, P, 64, device="cuda"); dz = torch.randn_like(x)
    ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1, device="cuda"); dw = torch.empty(3, 3, 3, 64, 64, device="cuda")
    t = timeit(lambda: ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws))
    print("wgrad 24^3 N=%3d: %.3f ms = %.4f ms per 8 patches" % (N, t, t * 8 / N), flush=True)
