import importlib, sys, torch
sys.path.insert(0, "/root/repo")
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
for P in (48, 24):
    N = 8
    pad = torch.randn((N, P + 2, P + 2, P + 2, 64), device="cuda"); out = torch.randn((N, P, P, P, 64), device="cuda")
    res = torch.randn_like(out); y = torch.randn_like(out)
    for _ in range(5): ops.fold_halo_border([pad], out, res, y, ops.ACT_LEAKY)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.fold_halo_border([pad], out, res, y, ops.ACT_LEAKY)
    e1.record(); torch.cuda.synchronize()
    print("fold_halo_border P=%d: %.1f us" % (P, e0.elapsed_time(e1) / 50 * 1e3))
