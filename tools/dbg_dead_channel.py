"""wgrad of operands with dead channels (x[..., ci] == 0 or dz[..., co] == 0): the result must be exactly zero there."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
g = torch.Generator(device="cuda").manual_seed(1)
for shape in [(8, 24, 24, 24), (2, 48, 48, 48)]:
    x = torch.randn(shape + (64,), device="cuda", generator=g).clamp_min(0) * 0.1
    dz = torch.randn(shape + (64,), device="cuda", generator=g) * 1e-4
    x[..., 0] = 0; x[..., 17] = 0; dz[..., 5] = 0; dz[..., 40] = 0
    ws = torch.empty(ops.wgrad_workspace_bytes(*shape, 64, 64, 3) // 4 + 1, device="cuda")
    for algo in (ops.ALGO_AUTO, ops.ALGO_DIRECT):
        dw, _ = ops.conv3d_wgrad(x, dz, 3, 64, 64, workspace=ws, algo=algo)
        a = dw[:, :, :, [0, 17], :].abs().max().item(); b = dw[..., [5, 40]].abs().max().item()
        print(shape, "algo", algo, "max |dW| on dead cin rows %.3e, dead cout cols %.3e, overall %.3e" % (a, b, dw.abs().max().item()))
