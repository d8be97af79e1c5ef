"""What the epilogue operands cost the bf16 conv at (4,128^3): forward with / without residual, fused dgrad with none / y / skip / skip + y.
python tools/abl_bf16_operands.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bops = importlib.import_module("4dflownet_amd.ops_bf16")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / it)
    return best
w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
wf, wd = bops.pack_conv64_weights(w)
N, P = 4, 128
x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16); res = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
y = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
out = torch.empty_like(x); pad = torch.empty(N, P + 2, P + 2, P + 2, 64, device="cuda")
for rep in range(2):
    f0 = t(lambda: bops.conv64_fwd(x, wf, None, 2, 0.2, None, out))
    f1 = t(lambda: bops.conv64_fwd(x, wf, None, 2, 0.2, res, out))
    d0 = t(lambda: bops.conv64_dgrad_fused(x, wd, pad, out, skip=None, y_prev=None, act=0))
    d1 = t(lambda: bops.conv64_dgrad_fused(x, wd, pad, out, skip=None, y_prev=y, act=2))
    d2 = t(lambda: bops.conv64_dgrad_fused(x, wd, pad, out, skip=res, y_prev=None, act=0))
    d3 = t(lambda: bops.conv64_dgrad_fused(x, wd, pad, out, skip=res, y_prev=y, act=2))
    print("bf16 (4,128^3): fwd %.3f  fwd+res %.3f | fused dgrad: none %.3f  y %.3f  skip %.3f  skip+y %.3f ms" % (f0, f1, d0, d1, d2, d3), flush=True)
    m = bops.new_sign_mask(out)
    f0m = t(lambda: bops.conv64_fwd(x, wf, None, 2, 0.2, None, out, mask=m))
    f1m = t(lambda: bops.conv64_fwd(x, wf, None, 2, 0.2, res, out, mask=m))
    d1m = t(lambda: bops.conv64_dgrad_fused(x, wd, pad, out, skip=None, y_prev=y, act=2, mask=m))
    d3m = t(lambda: bops.conv64_dgrad_fused(x, wd, pad, out, skip=res, y_prev=y, act=2, mask=m))
    print("   with sign masks: fwd %.3f  fwd+res %.3f (writing the mask) | fused dgrad: mask %.3f  skip+mask %.3f ms" % (f0m, f1m, d1m, d3m), flush=True)
