import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
for N, P in ((8, 48), (8, 24)):
    x = torch.randn((N, P, P, P, 64), device="cuda"); dz = torch.randn_like(x)
    ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1, device="cuda"); dw = torch.empty((3, 3, 3, 64, 64), device="cuda")
    for rep in range(3):
        for _ in range(3): ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        print("wgrad (%d,%d^3) %.4f ms incl. reduce" % (N, P, e0.elapsed_time(e1) / 40), flush=True)
