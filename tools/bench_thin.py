"""1x1x1 (64+64)->64 layer: MFMA kernels (conv1x1_mfma.hip) vs the VALU kernels (test-build switch), at the cfg2 / cfg4 shapes."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for N, P in ((8, 24), (4, 32)):
    xa = torch.randn(N, P, P, P, 64, device="cuda"); xb = torch.randn_like(xa); dz = torch.randn_like(xa)
    w = torch.randn(1, 1, 1, 128, 64, device="cuda") * 0.1; b = torch.randn(64, device="cuda")
    ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 128, 64, 1) // 4 + 1, device="cuda")
    res = {}
    for on in (0, 1, 0, 1):
        lib.fdn_debug_set_conv1x1_mfma(on)
        f = t(lambda: ops.conv3d_fwd(xa, w, b, ops.ACT_RELU, x2=xb))
        d = t(lambda: ops.conv1x1_dgrad(dz, w, xa, xb))
        g = t(lambda: ops.conv3d_wgrad(xa, dz, 1, 128, 64, x2=xb, workspace=ws))
        res[on] = (ops.conv3d_fwd(xa, w, b, ops.ACT_RELU, x2=xb), ops.conv1x1_dgrad(dz, w, xa, xb), ops.conv3d_wgrad(xa, dz, 1, 128, 64, x2=xb, workspace=ws)[0].clone())
        print("N=%d P=%d %s: fwd %.1f us, dgrad %.1f us, wgrad %.1f us" % (N, P, "MFMA" if on else "VALU", f, d, g), flush=True)
    rel = lambda a, b_: ((a - b_).abs().max() / b_.abs().max()).item()
    print("   MFMA vs VALU: fwd %.1e dgrad %.1e/%.1e wgrad %.1e" % (rel(res[1][0], res[0][0]), rel(res[1][1][0], res[0][1][0]), rel(res[1][1][1], res[0][1][1]), rel(res[1][2], res[0][2])))
lib.fdn_debug_set_conv1x1_mfma(1)
