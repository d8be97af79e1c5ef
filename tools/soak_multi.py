"""Soak of the multi-source fused dgrad (fdn_conv64_dgrad_fused_multi / _bf16_multi) against the chained single-source launches on random
shapes, source counts, pack orders and mask / y operands.   python tools/soak_multi.py [count] [seed] [max_extent] [f32|bf16]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 40
bf = len(sys.argv) > 4 and sys.argv[4] == "bf16"
ops = importlib.import_module("4dflownet_amd.ops_bf16") if bf else fdn.ops
adt = torch.bfloat16 if bf else torch.float32
rng = np.random.default_rng(seed)
worst, ran = 0.0, 0
for k in range(count):
    N, D = int(rng.integers(1, 5)), int(rng.integers(1, mx + 1))
    H, W = (int(rng.integers(1, mx + 1)), int(rng.integers(1, mx + 1))) if bf else (4 * int(rng.integers(1, mx // 4 + 1)), 4 * int(rng.integers(1, mx // 4 + 1)))
    if N * D * H * W > 120000: N = 1
    if not bf and not ops.conv64_mask_ok(N, D, H, W): continue
    nsrc = int(rng.integers(1, 4))
    g = torch.Generator(device="cuda").manual_seed(k + 31 * seed)
    r = lambda *s: torch.randn(s, device="cuda", generator=g)
    dzs = [r(N, D, H, W, 64).to(adt) for _ in range(nsrc)]
    y, skip = r(N, D, H, W, 64).to(adt), r(N, D, H, W, 64).to(adt)
    order = list(rng.permutation(nsrc))
    if bf:
        packs = [ops.pack_conv64_weights(r(3, 3, 3, 64, 64) * 0.05)[1] for _ in range(nsrc)]
    else:
        buf = torch.zeros((nsrc, 2, ops.CONV64_PACK_FLOATS), device="cuda")
        for s in range(nsrc): ops.pack_conv64_weights(r(3, 3, 3, 64, 64) * 0.05, buf[order[s], 0], buf[order[s], 1])
        packs = [buf[order[s], 1] for s in range(nsrc)]
    use_mask = bool(rng.integers(0, 2))
    mask = None
    if use_mask:
        if bf:
            bits = (y.float() > 0).view(N, D, H, W, 4, 16).to(torch.int32)
            mask = (bits << torch.arange(16, device="cuda", dtype=torch.int32)).sum(dim=-1)
        else:
            bits = (y > 0).view(N * D * H * W, 4, 16).to(torch.int32)
            mask = (bits << torch.arange(16, device="cuda", dtype=torch.int32)).sum(dim=2).t()
        mask = torch.where(mask >= 32768, mask - 65536, mask).to(torch.int16).contiguous()
    # chained reference (fp32 accumulation of the running sum in a float buffer for bf16: the chain itself rounds to bf16 per source)
    out_c = torch.zeros((N, D, H, W, 64), device="cuda", dtype=adt); pads = []
    for s in range(nsrc):
        pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda"); last = s == nsrc - 1
        ops.conv3d_dgrad_fused(dzs[s], packs[s], pad, out_c, skip=(skip if s == 0 else out_c), y_prev=y if last else None, act=2 if last else 0)
        pads.append(pad)
    ops.fold_halo_border(pads, out_c, skip, y, 2)
    pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
    out_m = torch.full((N, D, H, W, 64), float("nan"), device="cuda", dtype=adt)
    ops.conv3d_dgrad_fused_multi(dzs, packs, pad, out_m, skip=skip, y_prev=None if use_mask else y, act=2, mask=mask)
    ops.fold_halo_border([pad], out_m, skip, y, 2)
    err = (out_m.float() - out_c.float()).abs().max().item() / max(out_c.float().abs().max().item(), 1e-30)
    tol = 2.0 ** -6 if bf else 2e-5                  # bf16: the chain rounds the running sum twice more than the multi-source launch
    if not (err <= tol) or not torch.isfinite(out_m.float()).all(): print("MISMATCH", (N, D, H, W), nsrc, order, use_mask, err, flush=True)
    worst = max(worst, err if err == err else 1e9); ran += 1
print("%d shapes (%s): multi-source fused dgrad vs the chained launches, worst relative difference %.3e" % (ran, "bf16" if bf else "f32", worst))
