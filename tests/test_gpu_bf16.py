"""Parity of the bf16 activation path (BASELINE.json configs[3]) against the CPU oracle, through the C-ABI.

Model of the arithmetic: operands are bf16 values (inputs rounded with oracle.bf16_round), products are exact, the
accumulation is fp32, the stored result is rounded to bf16 once.  The oracle accumulates in float64, so a GPU result may
differ from round_bf16(oracle) only where fp32 accumulation noise crosses a rounding boundary: every element must be
within one bf16 ulp (2^-8 relative) + fp32 noise, and all but a small fraction must be bit-identical after rounding."""
import numpy as np
import pytest
import torch

from oracle import flownet_oracle as O
from test_gpu_kernels import variant_lib

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")


def devb(a):
    return dev(a).to(torch.bfloat16)


def close_bf16(got, ref, name="", max_flip_frac=0.02):
    got = got.detach().float().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref)
    bound = 2.0 ** -8 * np.abs(ref) + 2e-5 * scale
    assert np.isfinite(got).all(), name
    assert (err <= bound).all(), "%s: max excess %.3e (scale %.3e)" % (name, (err - bound).max(), scale)
    flips = np.mean(got != O.bf16_round(ref.astype(np.float32)).astype(np.float64))
    assert flips <= max_flip_frac, "%s: %.2f %% of elements differ from round_bf16(oracle)" % (name, 100 * flips)


def close_f32(got, ref, tol=2e-5, name=""):
    got = got.detach().float().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max() / scale
    assert err <= tol, "%s: max err %.3e of scale %.3e" % (name, err, scale)


@pytest.fixture(scope="module")
def bops(fdn):
    import importlib
    return importlib.import_module("4dflownet_amd.ops_bf16")


SHAPES = [(2, 8, 8, 8), (1, 5, 7, 9), (1, 10, 12, 16), (3, 4, 4, 2), (1, 1, 1, 1), (1, 16, 16, 16), (1, 3, 20, 11),
          (3, 24, 24, 24), (1, 17, 9, 12)]
# conv64 variants: 0 = planner, 4 / 8 = forced MT, +16 = full-depth tiles only (exercises the unrolled FAST kernel even
# where the planner would cut a small grid into thin tiles), +32 = never the two-slice kernel (MODE 2: 8 x 8 plane blocks, one LDS buffer
# of 64-B rows) -- i.e. the four-slice double-buffered FAST kernel on the grids where MODE 2 would otherwise take over
VARIANTS = [0, 4, 8, 20, 24, 52, 56]


def rb(a):
    return O.bf16_round(a.astype(np.float32)).astype(np.float64)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mt", VARIANTS)
def test_conv64_fwd_bf16(bops, fdn, shape, mt):
    rng = np.random.default_rng(11)
    N, D, H, W = shape
    x = rb(rng.normal(size=(N, D, H, W, 64)))
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    res = rb(rng.normal(size=(N, D, H, W, 64)))
    wf, _ = bops.pack_conv64_weights(dev(w))
    with variant_lib(fdn, mt) as lib:
        if lib is not None:
            lib.fdn_debug_set_conv64_bf16_mt(mt & 31)
            lib.fdn_debug_set_conv64_bf16_mode2(0 if mt & 32 else 1)
        try:
            for act, bias, r in [(O.ACT_RELU, b, None), (O.ACT_LEAKY, None, res), (O.ACT_NONE, None, None)]:
                ref = O.conv3d_fwd(x, rb(w), None if bias is None else bias.astype(np.float64), act, 0.2, r)
                got = bops.conv64_fwd(devb(x), wf, None if bias is None else dev(bias), act, 0.2,
                                      None if r is None else devb(r))
                close_bf16(got, ref, name="conv64 bf16 fwd act=%d" % act)
        finally:
            if lib is not None:
                lib.fdn_debug_set_conv64_bf16_mt(0)
                lib.fdn_debug_set_conv64_bf16_mode2(1)


@pytest.mark.parametrize("shape", SHAPES + [(1, 2, 3, 1)])
@pytest.mark.parametrize("mt", VARIANTS)
def test_conv64_dgrad_fused_bf16(bops, fdn, shape, mt):
    rng = np.random.default_rng(12)
    N, D, H, W = shape
    dz = rb(rng.normal(size=(N, D, H, W, 64)))
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    y = rb(rng.normal(size=(N, D, H, W, 64)))
    skip = rb(rng.normal(size=(N, D, H, W, 64)))
    dx = O.conv3d_dgrad(dz, rb(w), (N, D, H, W, 64))
    _, wd = bops.pack_conv64_weights(dev(w))
    with variant_lib(fdn, mt) as lib:
        if lib is not None:
            lib.fdn_debug_set_conv64_bf16_mt(mt & 31)
            lib.fdn_debug_set_conv64_bf16_mode2(0 if mt & 32 else 1)
        try:
            pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
            out = torch.full((N, D, H, W, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
            bops.conv64_dgrad_fused(devb(dz), wd, pad, out, skip=devb(skip), y_prev=devb(y), act=O.ACT_LEAKY)
            bops.fold_halo_border([pad], out, devb(skip), devb(y), O.ACT_LEAKY)
            close_bf16(out, O.act_bwd_from_output(dx + skip, y, O.ACT_LEAKY), name="bf16 fused dgrad+border")
        finally:
            if lib is not None:
                lib.fdn_debug_set_conv64_bf16_mt(0)
                lib.fdn_debug_set_conv64_bf16_mode2(1)


@pytest.mark.parametrize("shape", [(2, 8, 8, 8), (1, 5, 7, 9), (1, 10, 12, 16), (1, 16, 16, 16), (3, 24, 24, 24), (1, 3, 20, 11)])
@pytest.mark.parametrize("mt,nsrc,use_mask", [(0, 3, True), (0, 2, False), (8, 3, False), (20, 3, True), (52, 3, True), (4, 2, True), (0, 1, False)])
def test_multi_source_fused_dgrad_bf16(bops, fdn, shape, mt, nsrc, use_mask):
    """fdn_conv64_dgrad_fused_bf16_multi: dz_prev = fold(sum_s conv_T(dz_s, W_s)) as ONE launch, the sum kept in the fp32 accumulators of every
    kernel variant (two-slice MODE 2 + shell slabs in one launch, the four-slice FAST kernel, the general body).  Against the float64
    oracle within the bf16 bound; one source is bit-identical to the single-source entry point."""
    rng = np.random.default_rng(12)
    N, D, H, W = shape
    dzs = [rb(rng.normal(size=(N, D, H, W, 64))) for _ in range(nsrc)]
    ws = [(rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32) for _ in range(nsrc)]
    y = rb(rng.normal(size=(N, D, H, W, 64)))
    skip = rb(rng.normal(size=(N, D, H, W, 64)))
    dx = sum(O.conv3d_dgrad(dzs[s], rb(ws[s]), (N, D, H, W, 64)) for s in range(nsrc))
    packs = [bops.pack_conv64_weights(dev(w))[1] for w in ws]
    ydev = devb(y)
    mask = None
    if use_mask:
        bits = (ydev.float() > 0).view(N, D, H, W, 4, 16).to(torch.int32)
        mask = (bits << torch.arange(16, device="cuda", dtype=torch.int32)).sum(dim=-1)
        mask = torch.where(mask >= 32768, mask - 65536, mask).to(torch.int16).contiguous()
    with variant_lib(fdn, mt) as lib:
        if lib is not None:
            lib.fdn_debug_set_conv64_bf16_mt(mt & 31)
            lib.fdn_debug_set_conv64_bf16_mode2(0 if mt & 32 else 1)
        try:
            pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
            out = torch.full((N, D, H, W, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
            bops.conv3d_dgrad_fused_multi([devb(z) for z in dzs], packs, pad, out, skip=devb(skip), y_prev=None if use_mask else ydev,
                                          act=O.ACT_LEAKY, mask=mask)
            bops.fold_halo_border([pad], out, devb(skip), ydev, O.ACT_LEAKY)
            close_bf16(out, O.act_bwd_from_output(dx + skip, y, O.ACT_LEAKY), name="bf16 multi-source fused dgrad+border")
            if nsrc == 1:
                pad1 = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
                out1 = torch.full((N, D, H, W, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
                bops.conv64_dgrad_fused(devb(dzs[0]), packs[0], pad1, out1, skip=devb(skip), y_prev=ydev, act=O.ACT_LEAKY)
                bops.fold_halo_border([pad1], out1, devb(skip), ydev, O.ACT_LEAKY)
                assert torch.equal(out, out1)
        finally:
            if lib is not None:
                lib.fdn_debug_set_conv64_bf16_mt(0)
                lib.fdn_debug_set_conv64_bf16_mode2(1)


@pytest.mark.parametrize("variant", [0, 1])      # 0 = planner (the LDS-DMA kernel below 4 GB), 1 = the register-staged kernel (the >= 4 GB fallback)
@pytest.mark.parametrize("shape", [(2, 8, 8, 8), (1, 5, 7, 9), (1, 10, 12, 16), (3, 4, 4, 2), (2, 16, 16, 16), (1, 1, 1, 1),
                                   (1, 3, 20, 11), (1, 17, 9, 12), (2, 33, 8, 24)])
def test_conv64_wgrad_bf16(bops, fdn, shape, variant):
    """fp32 result from bf16 operands: products exact, accumulation fp32 -> fp32-level agreement with the oracle.
    (2, 33, 8, 24): depth segments of unequal length in the LDS-DMA kernel's unit walk."""
    rng = np.random.default_rng(13)
    N, D, H, W = shape
    x = rb(rng.normal(size=(N, D, H, W, 64)))
    dz = rb(rng.normal(size=(N, D, H, W, 64)))
    ref = O.conv3d_wgrad(x, dz, 3)
    with variant_lib(fdn, variant) as lib:
        if lib is not None:
            lib.fdn_debug_set_wgrad64_bf16_variant(1)
        try:
            dw, db = bops.conv3d_wgrad(devb(x), devb(dz), 3, 64, 64, want_bias=True)
        finally:
            if lib is not None:
                lib.fdn_debug_set_wgrad64_bf16_variant(0)
    close_f32(dw, ref, name="wgrad64 bf16")
    close_f32(db, O.bias_grad(dz), name="bias grad 64 bf16")


@pytest.mark.parametrize("shape", [(2, 6, 6, 6), (1, 5, 7, 9), (1, 1, 2, 3)])
def test_thin_layers_bf16(bops, shape):
    rng = np.random.default_rng(14)
    N, D, H, W = shape
    f64 = lambda a: a.astype(np.float64)
    # 3 -> 64
    x3 = rb(rng.normal(size=(N, D, H, W, 3)))
    w3 = (rng.normal(size=(3, 3, 3, 3, 64)) * 0.2).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    close_bf16(bops.conv3d_fwd(devb(x3), dev(w3), dev(b), O.ACT_RELU), O.conv3d_fwd(x3, f64(w3), f64(b), O.ACT_RELU), name="3->64 fwd")
    dz = rb(rng.normal(size=(N, D, H, W, 64)))
    dw, db = bops.conv3d_wgrad(devb(x3), devb(dz), 3, 3, 64, want_bias=True)
    close_f32(dw, O.conv3d_wgrad(x3, dz, 3), name="3->64 wgrad")
    close_f32(db, O.bias_grad(dz), name="3->64 bias grad")
    # 64 -> 1: bf16 input, fp32 prediction written into channel 1 of an (N,V,3) tensor
    x = rb(rng.normal(size=(N, D, H, W, 64)))
    w1 = (rng.normal(size=(3, 3, 3, 64, 1)) * 0.1).astype(np.float32)
    b1 = rng.normal(size=1).astype(np.float32)
    out = torch.zeros((N, D, H, W, 3), device="cuda")
    bops.conv3d_fwd(devb(x), dev(w1), dev(b1), O.ACT_NONE, out=out, ldy=3, y_coff=1)
    ref = np.zeros((N, D, H, W, 3)); ref[..., 1:2] = O.conv3d_fwd(x, f64(w1), f64(b1))
    close_f32(out, ref, name="64->1 fwd")
    dpred = rng.normal(size=(N, D, H, W, 3)).astype(np.float32)
    dzo = f64(dpred[..., 1:2])
    ymask = rb(rng.normal(size=(N, D, H, W, 64)))
    refd = O.conv3d_dgrad(dzo, f64(w1), x.shape) * (ymask > 0)
    dbp = torch.full((64,), float("nan"), device="cuda")
    got = bops.conv_cout1_dgrad_folded(dev(dpred), dev(w1), (N, D, H, W), devb(ymask), O.ACT_RELU, lddz=3, dz_coff=1, dbias_prev=dbp)
    close_bf16(got, refd, name="64->1 dgrad folded+relu")
    close_f32(dbp, O.bias_grad(refd), tol=3e-3, name="fused bias grad (sum of the unrounded fp32 values)")
    dw, db = bops.conv3d_wgrad(devb(x), dev(dpred), 3, 64, 1, want_bias=True, lddz=3, dz_coff=1)
    close_f32(dw, O.conv3d_wgrad(x, dzo, 3), name="64->1 wgrad")
    close_f32(db, O.bias_grad(dzo), name="64->1 bias grad")
    # 1x1x1 (64+64) -> 64
    xa = rb(rng.normal(size=(N, D, H, W, 64)))
    xb = rb(rng.normal(size=(N, D, H, W, 64)))
    wk = (rng.normal(size=(1, 1, 1, 128, 64)) * 0.1).astype(np.float32)
    cat = np.concatenate([xa, xb], -1)
    close_bf16(bops.conv3d_fwd(devb(xa), dev(wk), dev(b), O.ACT_RELU, x2=devb(xb)), O.conv3d_fwd(cat, f64(wk), f64(b), O.ACT_RELU), name="1x1 fwd")
    dcat = O.conv3d_dgrad(dz, f64(wk), cat.shape)
    da, dbb = bops.conv1x1_dgrad(devb(dz), dev(wk), devb(xa), devb(xb))
    close_bf16(da, dcat[..., :64] * (xa > 0), name="1x1 dgrad a")
    close_bf16(dbb, dcat[..., 64:] * (xb > 0), name="1x1 dgrad b")
    dw, db = bops.conv3d_wgrad(devb(xa), devb(dz), 1, 128, 64, x2=devb(xb), want_bias=True)
    close_f32(dw, O.conv3d_wgrad(cat, dz, 1), name="1x1 wgrad")


@pytest.mark.parametrize("R", [2, 4])
def test_upsample_and_features_bf16(bops, R):
    rng = np.random.default_rng(15)
    x = rb(rng.normal(size=(2, 5, 4, 6, 64)))
    ref = O.upsample_trilinear_fwd(x, R, f32_coeffs=True)
    close_bf16(bops.upsample_trilinear_fwd(devb(x), R), ref, name="upsample fwd")
    dy = rb(rng.normal(size=ref.shape))
    y = rb(rng.normal(size=x.shape))
    refb = O.upsample_trilinear_bwd(dy, x.shape[1:4], R, f32_coeffs=True)
    close_bf16(bops.upsample_trilinear_bwd(devb(dy), R), refb, name="upsample bwd")
    close_bf16(bops.upsample_trilinear_bwd(devb(dy), R, devb(y), O.ACT_LEAKY, 0.2), O.act_bwd_from_output(refb, y, O.ACT_LEAKY), name="upsample bwd+mask")
    batch = O.synthetic_batch(2, 6, 2, seed=11)
    ph, pc = bops.input_features(*[dev(a) for a in batch[:6]])
    rph, rpc = O.input_features(*[a.astype(np.float64) for a in batch[:6]])
    close_bf16(ph, rph, name="phase"); close_bf16(pc, rpc, name="pc")


def test_batched_bf16_pack_equals_per_layer_pack(bops):
    """fdn_pack_conv64_weights_bf16_batch (one launch after every optimizer step) writes the same streams as the per-layer entry point."""
    g = torch.Generator(device="cuda").manual_seed(5)
    n, sz = 5, 27 * 64 * 64
    flat = torch.randn(7 + n * (sz + 64), device="cuda", generator=g)
    offs = torch.tensor([7 + i * (sz + 64) for i in range(n)], device="cuda", dtype=torch.int64)
    packs = torch.zeros((n, 2, sz), device="cuda", dtype=torch.bfloat16)
    bops.pack_conv64_weights_batch(flat, offs, packs)
    for i in range(n):
        w = flat[7 + i * (sz + 64): 7 + i * (sz + 64) + sz].view(3, 3, 3, 64, 64)
        wf, wd = bops.pack_conv64_weights(w)
        assert torch.equal(packs[i, 0], wf) and torch.equal(packs[i, 1], wd)


@pytest.mark.parametrize("shape", [(2, 8, 8, 8), (1, 10, 12, 16), (1, 5, 7, 9), (1, 16, 16, 16)])
@pytest.mark.parametrize("act", [1, 2])
def test_sign_mask_replaces_y_in_the_fused_dgrad(bops, shape, act):
    """fdn_conv64_fwd_bf16_mask writes bit c of voxel v = (y[v][c] > 0); fdn_conv64_dgrad_fused_bf16_mask reads it for act' instead of y:
    same dz_prev and padded scratch bit for bit (MODE 2 tiles, ragged tiles and the shell slabs' general body)."""
    g = torch.Generator(device="cuda").manual_seed(17)
    N, D, H, W = shape
    x = torch.randn((N, D, H, W, 64), device="cuda", generator=g).to(torch.bfloat16)
    res = torch.randn((N, D, H, W, 64), device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn((3, 3, 3, 64, 64), device="cuda", generator=g) * 0.05
    b = torch.randn(64, device="cuda", generator=g) * 0.1
    wf, wd = bops.pack_conv64_weights(w)
    y0 = bops.conv64_fwd(x, wf, b, act, 0.2, res)
    mask = bops.new_sign_mask(y0)
    mask.fill_(-1)
    y1 = bops.conv64_fwd(x, wf, b, act, 0.2, res, mask=mask)
    assert torch.equal(y0, y1)
    bits = (mask.to(torch.int32) & 0xffff).unsqueeze(-1) >> torch.arange(16, device="cuda", dtype=torch.int32) & 1     # (N,D,H,W,4,16)
    assert torch.equal(bits.reshape(N, D, H, W, 64).bool(), y0.float() > 0)
    assert 0.05 < bits.float().mean().item() < 0.95
    dz = torch.randn((N, D, H, W, 64), device="cuda", generator=g).to(torch.bfloat16)
    outs = []
    for m in (None, mask):
        pad = torch.zeros((N, D + 2, H + 2, W + 2, 64), device="cuda")
        out = torch.zeros_like(dz)
        bops.conv3d_dgrad_fused(dz, wd, pad, out, skip=res, y_prev=y0, act=act, mask=m)
        bops.fold_halo_border([pad], out, res, y0, act)
        outs.append((out, pad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # the mask alone is enough for the conv launch (y_prev may be NULL there)
    pad = torch.zeros((N, D + 2, H + 2, W + 2, 64), device="cuda"); out = torch.zeros_like(dz)
    bops.conv3d_dgrad_fused(dz, wd, pad, out, skip=res, y_prev=None, act=act, mask=mask)
    bops.fold_halo_border([pad], out, res, y0, act)
    assert torch.equal(out, outs[0][0])


@pytest.mark.parametrize("shape,nl", [((4, 32, 32, 32), 9), ((4, 32, 32, 32), 2), ((1, 40, 24, 16), 8), ((2, 8, 8, 8), 3)])
def test_batched_bf16_wgrad_equals_per_layer_launches(bops, shape, nl):
    """fdn_conv3d_wgrad_bf16_batch: several 64->64 layers of one grid in chunks of up to seven per launch (9 = 7 + 2, 8 = 4 + 4); a different
    split of the voxel sum than the per-layer launch, so equal to fp32 rounding; a grid too small to batch loops over the per-layer path
    bit-identically."""
    g = torch.Generator(device="cuda").manual_seed(23)
    N, D, H, W = shape
    xs = [torch.randn((N, D, H, W, 64), device="cuda", generator=g).to(torch.bfloat16) for _ in range(nl)]
    dzs = [(torch.randn((N, D, H, W, 64), device="cuda", generator=g) * (0.3 + i)).to(torch.bfloat16) for i in range(nl)]
    dws = [torch.full((3, 3, 3, 64, 64), float("nan"), device="cuda") for _ in range(nl)]
    dbs = [torch.full((64,), float("nan"), device="cuda") if i % 2 else None for i in range(nl)]
    bops.conv3d_wgrad_batch(xs, dzs, dws, dbs)
    small = N * D * ((H + 7) // 8) * ((W + 7) // 8) < 320
    for i in range(nl):
        dw, db = bops.conv3d_wgrad(xs[i], dzs[i], 3, 64, 64, want_bias=dbs[i] is not None)
        scale = dw.abs().max().item()
        assert torch.isfinite(dws[i]).all()
        if small:
            assert torch.equal(dws[i], dw)
        else:
            assert (dws[i] - dw).abs().max().item() <= 2e-5 * scale, (i, (dws[i] - dw).abs().max().item() / scale)
        if dbs[i] is not None:
            assert torch.equal(dbs[i], db)


@pytest.mark.parametrize("shape", [(2, 8, 8, 8), (1, 5, 7, 9), (1, 12, 16, 24)])
def test_sign_mask_replaces_y_in_the_head_dgrad(bops, shape):
    """fdn_conv_cout1_dgrad_folded_bf16_mask: the 64->1 head's input gradient reads the sign mask of the head activation instead of its rows --
    same dz_prev and producer bias gradient bit for bit."""
    g = torch.Generator(device="cuda").manual_seed(29)
    N, D, H, W = shape
    x = torch.randn((N, D, H, W, 64), device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn((3, 3, 3, 64, 64), device="cuda", generator=g) * 0.05
    wf, _ = bops.pack_conv64_weights(w)
    mask = bops.new_sign_mask(x)
    y = bops.conv64_fwd(x, wf, None, 1, 0.2, None, mask=mask)                      # the head's 64->64 conv (ReLU)
    w1 = torch.randn((3, 3, 3, 64, 1), device="cuda", generator=g) * 0.1
    dpred = torch.randn((N, D, H, W, 3), device="cuda", generator=g)
    outs = []
    for m in (None, mask):
        db = torch.zeros(64, device="cuda")
        out = bops.conv_cout1_dgrad_folded(dpred, w1, (N, D, H, W), y, 1, lddz=3, dz_coff=2, dbias_prev=db, mask=m)
        outs.append((out, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert (outs[0][0] == 0).float().mean().item() > 0.2               # the ReLU mask really bites
