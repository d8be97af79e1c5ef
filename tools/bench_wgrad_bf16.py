import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bops = importlib.import_module("4dflownet_amd.ops_bf16")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters
for N, P in ((8, 24), (8, 48), (4, 32), (4, 128)):
    x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
    dz = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
    dw = torch.empty(3, 3, 3, 64, 64, device="cuda")
    ws = torch.empty(bops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1, device="cuda")
    ms = timeit(lambda: bops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws))
    gf = 2 * 27 * 64 * 64 * N * P ** 3 / 1e9
    print("wgrad64 bf16 N=%d P=%-3d: %.3f ms  %.0f TF" % (N, P, ms, gf / ms), flush=True)
