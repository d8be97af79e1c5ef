import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("4dflownet_amd.ops_bf16")
N, P = 4, 128
x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
w = torch.randn(3, 3, 3, 64, 1, device="cuda") * 0.1
b = torch.randn(1, device="cuda")
pred = torch.zeros(N, P, P, P, 3, device="cuda")
for _ in range(4):
    ops.conv3d_fwd(x, w, b, 0, out=pred, ldy=3, y_coff=1)
torch.cuda.synchronize()
