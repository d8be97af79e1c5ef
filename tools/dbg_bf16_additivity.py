"""bf16 conv forward / fused dgrad at N = 2 against the two N = 1 halves (bit-identical expected), MODE 2 on and off (debugging aid)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
bops = importlib.import_module("4dflownet_amd.ops_bf16")
torch.manual_seed(0)
w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
wf, wd = bops.pack_conv64_weights(w)
with fdn._lib.test_build() as lib:
    for P in (16, 32, 64, 128):
        x = torch.randn(2, P, P, P, 64, device="cuda").to(torch.bfloat16)
        for mode2 in (1, 0):
            lib.fdn_debug_set_conv64_bf16_mode2(mode2)
            y2 = bops.conv64_fwd(x, wf, None, 1, 0.2, None, None).float()
            ya = bops.conv64_fwd(x[0:1].contiguous(), wf, None, 1, 0.2, None, None).float()
            yb = bops.conv64_fwd(x[1:2].contiguous(), wf, None, 1, 0.2, None, None).float()
            y2b = bops.conv64_fwd(x, wf, None, 1, 0.2, None, None).float()
            d = (y2 - torch.cat([ya, yb])).abs()
            print("P=%3d mode2=%d: max |N=2 - halves| = %.3e (sample 0: %.3e, sample 1: %.3e), run-to-run %.3e, scale %.2f" %
                  (P, mode2, d.max().item(), d[0].max().item(), d[1].max().item(), (y2 - y2b).abs().max().item(), y2.abs().max().item()), flush=True)
            if d.max().item() > 0:
                idx = (d > 0).nonzero()
                print("   mismatching elements: %d; first few (n,d,h,w,c): %s" % (idx.shape[0], idx[:6].tolist()))
                print("   d range of mismatches:", idx[:, 1].min().item(), idx[:, 1].max().item(), " h:", idx[:, 2].min().item(), idx[:, 2].max().item(),
                      " w:", idx[:, 3].min().item(), idx[:, 3].max().item())
    lib.fdn_debug_set_conv64_bf16_mode2(1)
