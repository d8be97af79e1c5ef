"""The 64 -> 1 head kernels in bf16 storage at the cfg4 shape (4,128^3): python tools/bench_heads_bf16.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops; bops = importlib.import_module("4dflownet_amd.ops_bf16")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
N, P = 4, 128
g = torch.randn((N, P, P, P, 64), device="cuda").to(torch.bfloat16); w = torch.randn((3, 3, 3, 64, 1), device="cuda") * 0.1
b = torch.randn(1, device="cuda"); pred = torch.empty((N, P, P, P, 3), device="cuda"); dpred = torch.randn_like(pred)
wsb = torch.empty(2048 * 64, device="cuda"); gb = torch.empty(64, device="cuda")
print("bf16 head fwd %.1f us" % timeit(lambda: bops.conv3d_fwd(g, w, b, ops.ACT_NONE, out=pred, ldy=3, y_coff=1)))
print("bf16 head dgrad (folded, + act', + bias grad of the producer) %.1f us" % timeit(
    lambda: bops.conv_cout1_dgrad_folded(dpred, w, (N, P, P, P), g, ops.ACT_RELU, lddz=3, dz_coff=1, dbias_prev=gb, workspace=wsb)))
print("bf16 head wgrad (+reduce) %.1f us" % timeit(lambda: bops.conv3d_wgrad(g, dpred, 3, 64, 1, want_bias=True, lddz=3, dz_coff=1)))
