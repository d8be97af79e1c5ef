"""One-off soak of the tile / region / XCD-order logic: Winograd kernels of the product library vs the direct MFMA kernels (test build) on many
random shapes (the same comparison as tests/test_gpu_kernels.py::test_winograd_kernels_equal_direct_kernels_on_random_shapes).
    python tools/soak_wino.py [count] [seed] [max_extent] [h4|anyw|-] [auto|bf16x3|masks]
    h4: H a multiple of 4 as well -> the F(4,3) x F(4,3) kernel on every shape (half-size and full tiles); anyw: any W, N up to 40 -> grids off
    the multiple-of-4 raster (aligned box + direct strips above 24 576 voxels, all direct below); bf16x3: FDN_ALGO_WINO_BF16X3 for forward / dgrad;
    masks (with h4): additionally the sign-mask forms (fdn_conv64_fwd_mask / fdn_conv64_dgrad_fused_mask) against the y forms, bit for bit"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 40
h4 = len(sys.argv) > 4 and sys.argv[4] == "h4"
anyw = len(sys.argv) > 4 and sys.argv[4] == "anyw"
algo = ops.ALGO_WINO_BF16X3 if len(sys.argv) > 5 and sys.argv[5] == "bf16x3" else ops.ALGO_AUTO
masks = len(sys.argv) > 5 and sys.argv[5] == "masks"
nmask = 0
rng = np.random.default_rng(seed)
worst = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for k in range(count):
    N, D, H, W = int(rng.integers(1, 5)), int(rng.integers(1, mx + 1)), int(rng.integers(1, mx + 1)), 4 * int(rng.integers(1, mx // 4 + 1))
    if h4: H = 4 * int(rng.integers(1, mx // 4 + 1))
    if anyw:
        W = int(rng.integers(1, mx + 1))
        N = int(rng.integers(1, 41))
        while N > 1 and N * D * H * W > 120000: N //= 2
    if N * D * H * W > 120000: N = 1
    g = torch.Generator(device="cuda").manual_seed(k + 17 * seed)
    x = torch.randn((N, D, H, W, 64), device="cuda", generator=g); res = torch.randn((N, D, H, W, 64), device="cuda", generator=g)
    dz = torch.randn((N, D, H, W, 64), device="cuda", generator=g); w = torch.randn((3, 3, 3, 64, 64), device="cuda", generator=g) * 0.05
    b = torch.randn((64,), device="cuda", generator=g); yfix = torch.randn((N, D, H, W, 64), device="cuda", generator=g)
    wf, wd = ops.pack_conv64_weights(w)
    Wg = max(1, W + 3 - int(rng.integers(0, 4)))
    xg = torch.randn((N, D, H, Wg, 64), device="cuda", generator=g); dzg = torch.randn((N, D, H, Wg, 64), device="cuda", generator=g)
    def run(algo=ops.ALGO_AUTO):
        y = ops.conv3d_fwd(x, w, b, ops.ACT_LEAKY, 0.2, res, wpack=wf, algo=algo)
        pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda"); out = torch.zeros_like(x)
        ops.conv3d_dgrad_fused(dz, wd, pad, out, skip=res, y_prev=yfix, act=ops.ACT_LEAKY, algo=algo)
        ops.fold_halo_border([pad], out, res, yfix, ops.ACT_LEAKY)
        dw, _ = ops.conv3d_wgrad(xg, dzg, 3, 64, 64, algo=algo)
        return y, out, dw
    got = run(algo)
    if masks and ops.conv64_mask_ok(N, D, H, W):
        nmask += 1
        m = ops.new_sign_mask(got[0]); m.fill_(0x1234)
        ym = ops.conv3d_fwd(x, w, b, ops.ACT_LEAKY, 0.2, res, wpack=wf, mask=m)
        bits = (ym > 0).view(N * D * H * W, 4, 16).to(torch.int32)
        words = (bits << torch.arange(16, device="cuda", dtype=torch.int32)).sum(dim=2).t().contiguous()
        pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda"); outm = torch.zeros_like(x)
        ops.conv3d_dgrad_fused(dz, wd, pad, outm, skip=res, y_prev=None, act=ops.ACT_LEAKY, mask=m)      # m is the mask of ym: compare with y_prev = ym
        pad2 = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda"); outy = torch.zeros_like(x)
        ops.conv3d_dgrad_fused(dz, wd, pad2, outy, skip=res, y_prev=ym, act=ops.ACT_LEAKY)
        if not (torch.equal(ym, got[0]) and torch.equal(words, m.to(torch.int32) & 0xffff) and torch.equal(outm, outy)
                and torch.equal(torch.nan_to_num(pad, nan=-7.0), torch.nan_to_num(pad2, nan=-7.0))):
            print("MASK MISMATCH", (N, D, H, W), flush=True); worst["fwd"] = 1e9
    with fdn._lib.test_build() as lib:
        lib.fdn_debug_set_conv64_mt(5); lib.fdn_debug_set_wgrad64_direct(1)
        try: ref = run()
        finally: lib.fdn_debug_set_conv64_mt(0); lib.fdn_debug_set_wgrad64_direct(0)
    for name, a, r in zip(("fwd", "dgrad", "wgrad"), got, ref):
        err = (a - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        if not (err <= 1e-5): print("MISMATCH", name, (N, D, H, W, Wg), err, flush=True)
        worst[name] = max(worst[name], err if err == err else 1e9)
print("%d shapes (%s, %s), worst relative difference:" % (count, sys.argv[4] if len(sys.argv) > 4 else "-", "bf16x3" if algo == ops.ALGO_WINO_BF16X3 else "masks" if masks else "auto"), worst,
      ("; %d shapes with sign masks: forward, mask bits and fused dgrad bit-identical to the y forms" % nmask) if masks else "")
