import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bops = importlib.import_module("4dflownet_amd.ops_bf16")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
wf, wd = bops.pack_conv64_weights(w)
for N, P in ((4, 128), (4, 32)):
    x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16); res = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(x); pad = torch.empty(N, P + 2, P + 2, P + 2, 64, device="cuda")
    f = t(lambda: bops.conv64_fwd(x, wf, None, 2, 0.2, res, out))
    d = t(lambda: (bops.conv64_dgrad_fused(x, wd, pad, out, skip=res, y_prev=res, act=2), bops.fold_halo_border([pad], out, res, res, 2)))
    print("bf16 (%d,%d^3): fwd %.3f ms, fused dgrad + border %.3f ms (dgrad - fwd = %.0f us)" % (N, P, f, d, (d - f) * 1e3))
