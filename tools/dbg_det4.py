import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
bops = importlib.import_module("4dflownet_amd.ops_bf16")
lib = fdn._lib.load()
torch.manual_seed(0)
w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
wf, wd = bops.pack_conv64_weights(w)
wt = w.to(torch.bfloat16).float().permute(4, 3, 0, 1, 2).contiguous()
for P, N, mt in ((24, 2, 4), (24, 3, 4), (32, 1, 4), (32, 2, 4), (32, 2, 8), (32, 3, 8)):
    x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
    xp = torch.nn.functional.pad(x.float().permute(0, 4, 1, 2, 3), (1, 1, 1, 1, 1, 1), mode="replicate")
    ref = torch.nn.functional.conv3d(xp, wt).permute(0, 2, 3, 4, 1)
    lib.fdn_debug_set_conv64_bf16_mt(mt)
    y = bops.conv64_fwd(x, wf, None, 0).float()
    torch.cuda.synchronize()
    bad = ((y - ref).abs() > 0.05)
    print("P=%d N=%d mt=%d nbad %d; per-n bad frac %s; bad by d-plane (n=0): %s" % (P, N, mt, bad.sum().item(), [round(bad[n].float().mean().item(), 3) for n in range(N)], [round(bad[0, d].float().mean().item(), 2) for d in range(P)]), flush=True)
