"""A/B of the fused dgrad's w faces (round 3): a region of the Winograd launch (xi in {0,5} K loop) vs the round-2 separate launch of the
direct kernel.  Test build (fdn_debug_set_conv64_wface_direct).  Times the whole fused dgrad + border fold, and the shell part alone."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
N = 8
def t(fn, it=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for P in (48, 24):
    x = torch.randn((N, P, P, P, 64), device="cuda"); w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.02
    res = torch.randn_like(x); wf, wd = ops.pack_conv64_weights(w); y = torch.randn_like(x)
    pad = torch.empty((N, P + 2, P + 2, P + 2, 64), device="cuda"); out = torch.empty_like(x)
    outs = {}
    for direct in (1, 0, 1, 0):
        lib.fdn_debug_set_conv64_wface_direct(direct)
        def full():
            ops.conv3d_dgrad_fused(x, wd, pad, out, skip=res, y_prev=y, act=ops.ACT_LEAKY)
            ops.fold_halo_border([pad], out, res, y, ops.ACT_LEAKY)
        def shell(): ops.conv3d_dgrad_fused(x, wd, pad, out, skip=res, y_prev=y, act=ops.ACT_LEAKY, parts=ops.DGRAD_SHELL)
        print("P=%d w faces %s: fused dgrad + fold %.1f us, shell part alone %.1f us" % (
            P, "direct launch" if direct else "wino region  ", t(full), t(shell)), flush=True)
        pad.fill_(float("nan")); full(); outs[direct] = out.clone()
    d = (outs[0] - outs[1]).abs().max().item() / outs[1].abs().max().item()
    print("P=%d max |wino-region - direct| / max = %.2e" % (P, d))
lib.fdn_debug_set_conv64_wface_direct(0)
