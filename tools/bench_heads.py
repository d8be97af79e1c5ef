"""The 64 -> 1 head kernels at the cfg2 shape (8,48^3): python tools/bench_heads.py"""
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
N, P = 8, 48
g = torch.randn((N, P, P, P, 64), device="cuda"); w = torch.randn((3, 3, 3, 64, 1), device="cuda") * 0.1
b = torch.randn(1, device="cuda"); pred = torch.empty((N, P, P, P, 3), device="cuda"); dpred = torch.randn_like(pred)
ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 64, 1, 3) // 4 + 1, device="cuda"); dw = torch.empty_like(w)
wsb = torch.empty(2048 * 64, device="cuda"); gb = torch.empty(64, device="cuda")
for v in (3, 1, 3, 1):
    lib.fdn_debug_set_heads_mfma(v)
    print("head fwd (xcd walk %s) %.1f us" % ("off" if v & 2 else "on", timeit(lambda: ops.conv3d_fwd(g, w, b, ops.ACT_NONE, out=pred, ldy=3, y_coff=1))))
print("head dgrad (folded, + act', + bias grad of the producer) %.1f us" % timeit(
    lambda: ops.conv_cout1_dgrad_folded(dpred, w, (N, P, P, P), g, ops.ACT_RELU, lddz=3, dz_coff=1, dbias_prev=gb, workspace=wsb)))
bits = (g > 0).view(N * P * P * P, 4, 16).to(torch.int32)
mask = (bits << torch.arange(16, device="cuda", dtype=torch.int32)).sum(dim=2).t().contiguous()
mask = torch.where(mask >= 32768, mask - 65536, mask).to(torch.int16)
del bits
print("head dgrad with the producer's sign mask instead of y (+ bias grad) %.1f us" % timeit(
    lambda: ops.conv_cout1_dgrad_folded(dpred, w, (N, P, P, P), None, ops.ACT_RELU, lddz=3, dz_coff=1, dbias_prev=gb, workspace=wsb, mask=mask)))
print("head wgrad (+reduce) %.1f us" % timeit(lambda: ops.conv3d_wgrad(g, dpred, 3, 64, 1, dw=dw, workspace=ws, lddz=3, dz_coff=1)))
print("head dgrad without the act' mask (no y_prev loads) %.1f us" % timeit(
    lambda: ops.conv_cout1_dgrad_folded(dpred, w, (N, P, P, P), None, ops.ACT_NONE, lddz=3, dz_coff=1)))
print("head dgrad without bias grad %.1f us" % timeit(
    lambda: ops.conv_cout1_dgrad_folded(dpred, w, (N, P, P, P), g, ops.ACT_RELU, lddz=3, dz_coff=1)))
d1 = torch.randn((N, P, P, P, 1), device="cuda")
print("head dgrad, dz contiguous (lddz=1) %.1f us" % timeit(
    lambda: ops.conv_cout1_dgrad_folded(d1, w, (N, P, P, P), g, ops.ACT_RELU, lddz=1, dz_coff=0)))
