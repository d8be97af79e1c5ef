"""The hot kernels at the EXACT bench launch shapes -- (8,48^3,64) and (8,24^3,64) fp32 (BASELINE cfg2, the planner's
choices: Winograd tiles of 8x8x1 groups, 3456 / 432 workgroups) and (4,128^3,64) bf16 (cfg4) -- against a float64 evaluation
of SAMPLED outputs: the CPU oracle would need minutes per full tensor here, a sample of voxels (all 8 corners, edges, faces and
random interior points of several batch entries) needs seconds.  Same tolerances as the small-shape kernel tests."""
import importlib

import numpy as np
import pytest
import torch

from oracle import flownet_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 2e-5


def sample_voxels(N, D, H, W, n_random, rng):
    """(n, d, h, w) rows: every corner / some edge and face points of sample 0 and N-1, plus random voxels of all samples."""
    pts = []
    for n in sorted({0, N - 1}):
        for d in (0, D - 1):
            for h in (0, H - 1):
                for w in (0, W - 1):
                    pts.append((n, d, h, w))
        pts += [(n, 0, H // 2, W // 3), (n, D - 1, 1, W - 2), (n, D // 2, 0, 5), (n, 7, H - 1, W // 2), (n, D // 3, H // 2, 0),
                (n, 5, 6, W - 1), (n, 0, 0, W // 2), (n, D - 1, H // 2, W - 1), (n, 1, 1, 1), (n, D - 2, H - 2, W - 2)]
    r = np.stack([rng.integers(0, N, n_random), rng.integers(0, D, n_random), rng.integers(0, H, n_random), rng.integers(0, W, n_random)], 1)
    return np.concatenate([np.asarray(pts, dtype=np.int64), r.astype(np.int64)], 0)


def gather_rows(t, idx):
    """t (N,D,H,W,C) device tensor, idx (k,4) -> (k,C) float64 numpy."""
    i = torch.as_tensor(idx, device=t.device)
    return t[i[:, 0], i[:, 1], i[:, 2], i[:, 3]].double().cpu().numpy()


def ref_forward(x, w64, pts, dims):
    """sum_t x[clamp(p + t - 1)] @ w[t] for the sampled voxels (SYMMETRIC p=1 == edge clamp), float64."""
    N, D, H, W = dims
    out = np.zeros((len(pts), w64.shape[-1]))
    for a in range(3):
        for b in range(3):
            for c in range(3):
                q = pts.copy()
                q[:, 1] = np.clip(pts[:, 1] + a - 1, 0, D - 1); q[:, 2] = np.clip(pts[:, 2] + b - 1, 0, H - 1)
                q[:, 3] = np.clip(pts[:, 3] + c - 1, 0, W - 1)
                out += gather_rows(x, q) @ w64[a, b, c]
    return out


def ref_dgrad(dz, w64, pts, dims):
    """dx[i] = sum over (o, t) with clamp(o + t - 1) == i of dz[o] @ w[t]^T  (Conv3DBackpropInput + MirrorPadGrad), float64."""
    N, D, H, W = dims
    out = np.zeros((len(pts), 64))
    ext = (D, H, W)
    for k, (n, d, h, w_) in enumerate(pts):
        i = (d, h, w_)
        cand = [[(o, t) for o in range(max(i[ax] - 1, 0), min(i[ax] + 2, ext[ax])) for t in range(3)
                 if min(max(o + t - 1, 0), ext[ax] - 1) == i[ax]] for ax in range(3)]
        rows, taps = [], []
        for (od, ta) in cand[0]:
            for (oh, tb) in cand[1]:
                for (ow, tc) in cand[2]:
                    rows.append((n, od, oh, ow)); taps.append((ta, tb, tc))
        dzr = gather_rows(dz, np.asarray(rows, dtype=np.int64))
        for r, (ta, tb, tc) in zip(dzr, taps):
            out[k] += w64[ta, tb, tc] @ r
    return out


def ref_wgrad_rows(x, dz, picks):
    """dW[a,b,c][ci][:] = sum_vox x[clamp(vox + (a,b,c) - 1)][ci] * dz[vox][:] for the picked (a,b,c,ci), float64 on the device
    with plain torch indexing (no library kernel involved)."""
    N, D, H, W, _ = x.shape
    dzd = dz.double()
    out = []
    cl = lambda n, t: torch.clamp(torch.arange(n, device=x.device) + t - 1, 0, n - 1)
    for (a, b, c, ci) in picks:
        xs = x[:, cl(D, a)][:, :, cl(H, b)][:, :, :, cl(W, c)][..., ci].double()
        out.append(torch.einsum("ndhw,ndhwo->o", xs, dzd).cpu().numpy())
    return np.asarray(out)


@pytest.mark.parametrize("N,P", [(8, 48), (8, 24)])
def test_conv64_fp32_at_bench_shapes_sampled_float64(fdn, N, P):
    ops = fdn.ops
    rng = np.random.default_rng(5)
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((N, P, P, P, 64), device="cuda", generator=g)
    res = torch.randn((N, P, P, P, 64), device="cuda", generator=g)
    dz = torch.randn((N, P, P, P, 64), device="cuda", generator=g)
    w = torch.randn((3, 3, 3, 64, 64), device="cuda", generator=g) * 0.03
    bias = torch.randn((64,), device="cuda", generator=g)
    w64 = w.double().cpu().numpy()
    pts = sample_voxels(N, P, P, P, 300, rng)
    dims = (N, P, P, P)
    wf, wd = ops.pack_conv64_weights(w)

    # forward: bias + residual + LeakyReLU epilogue
    y = ops.conv3d_fwd(x, w, bias, ops.ACT_LEAKY, 0.2, res, wpack=wf)
    z = ref_forward(x, w64, pts, dims) + bias.double().cpu().numpy() + gather_rows(res, pts)
    ref = np.where(z > 0, z, 0.2 * z)
    got = gather_rows(y, pts)
    assert np.abs(got - ref).max() <= RTOL * np.abs(ref).max(), "fwd"

    # fused dgrad (Winograd inner box + shell slabs + border fold) with skip and act'
    pad = torch.empty((N, P + 2, P + 2, P + 2, 64), device="cuda")
    out = torch.empty_like(x)
    ops.conv3d_dgrad_fused(dz, wd, pad, out, skip=res, y_prev=y, act=ops.ACT_LEAKY)
    ops.fold_halo_border([pad], out, res, y, ops.ACT_LEAKY)
    refd = (ref_dgrad(dz, w64, pts, dims) + gather_rows(res, pts)) * np.where(gather_rows(y, pts) > 0, 1.0, 0.2)
    gotd = gather_rows(out, pts)
    assert np.abs(gotd - refd).max() <= RTOL * np.abs(refd).max(), "dgrad"

    # weight gradient: sampled (tap, ci) rows, all 64 cout
    ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1, device="cuda")
    dw, _ = ops.conv3d_wgrad(x, dz, 3, 64, 64, workspace=ws)
    picks = [(0, 0, 0, 0), (1, 1, 1, 17), (2, 2, 2, 63), (0, 2, 1, 31), (2, 0, 1, 40), (1, 0, 2, 5), (1, 2, 0, 58), (0, 1, 2, 22)]
    refw = ref_wgrad_rows(x, dz, picks)
    gotw = np.asarray([dw[a, b, c, ci].double().cpu().numpy() for (a, b, c, ci) in picks])
    full_scale = float(dw.abs().max())
    assert np.abs(gotw - refw).max() <= RTOL * full_scale, "wgrad"


def test_conv64_fp32_at_the_2d_kernels_size_limit_sampled_float64(fdn):
    """The 2-D Winograd kernel addresses a sample with 30-bit byte offsets (<= 2^22 voxels); a larger sample must take the 1-D
    kernels instead (fdn_conv64_wino2d_ok).  Forward and fused dgrad just below the limit (160 x 160 x 160 = 4 096 000 voxels:
    2-D kernels, the "reads zero" marker one step above the largest real offset) and just above it (164^3: fallback), sampled
    against float64 -- every corner, edge, face sample and random interior voxels, i.e. the largest offsets of the tensor."""
    ops = fdn.ops
    rng = np.random.default_rng(9)
    g = torch.Generator(device="cuda").manual_seed(13)
    w = torch.randn((3, 3, 3, 64, 64), device="cuda", generator=g) * 0.03
    w64 = w.double().cpu().numpy()
    wf, wd = ops.pack_conv64_weights(w)
    for P in (160, 164):
        x = torch.randn((1, P, P, P, 64), device="cuda", generator=g)
        pts = sample_voxels(1, P, P, P, 200, rng)
        dims = (1, P, P, P)
        y = ops.conv3d_fwd(x, w, None, ops.ACT_NONE, wpack=wf)
        ref = ref_forward(x, w64, pts, dims)
        assert np.abs(gather_rows(y, pts) - ref).max() <= RTOL * np.abs(ref).max(), ("fwd", P)
        pad = torch.empty((1, P + 2, P + 2, P + 2, 64), device="cuda")
        out = torch.empty_like(x)
        ops.conv3d_dgrad_fused(x, wd, pad, out)
        ops.fold_halo_border([pad], out)
        refd = ref_dgrad(x, w64, pts, dims)
        assert np.abs(gather_rows(out, pts) - refd).max() <= RTOL * np.abs(refd).max(), ("dgrad", P)
        del x, y, pad, out
        torch.cuda.empty_cache()


def test_conv64_bf16_at_cfg4_shape_sampled_float64(fdn):
    """cfg4 launch shape (4,128^3,64) in bf16 storage: forward and fused dgrad, sampled voxels vs float64 of the same bf16
    operands (products exact, fp32 accumulation, one bf16 rounding of the result: within one bf16 ulp)."""
    bops = importlib.import_module("4dflownet_amd.ops_bf16")
    N, P = 4, 128
    rng = np.random.default_rng(6)
    g = torch.Generator(device="cuda").manual_seed(12)
    x = torch.randn((N, P, P, P, 64), device="cuda", generator=g).to(torch.bfloat16)
    dz = torch.randn((N, P, P, P, 64), device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn((3, 3, 3, 64, 64), device="cuda", generator=g) * 0.03
    wb64 = w.to(torch.bfloat16).double().cpu().numpy()          # the kernels multiply with the bf16-rounded weights
    pts = sample_voxels(N, P, P, P, 200, rng)
    dims = (N, P, P, P)
    wf, wd = bops.pack_conv64_weights(w)
    y = bops.conv64_fwd(x, wf, None, bops.ACT_RELU, 0.2, None)
    ref = np.maximum(ref_forward(x, wb64, pts, dims), 0)
    got = gather_rows(y, pts)
    assert (np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 2e-5 * np.abs(ref).max()).all(), "bf16 fwd"
    del y
    pad = torch.empty((N, P + 2, P + 2, P + 2, 64), device="cuda")
    out = torch.empty_like(dz)
    bops.conv64_dgrad_fused(dz, wd, pad, out, skip=None, y_prev=None, act=bops.ACT_NONE)
    bops.fold_halo_border([pad], out, None, None, bops.ACT_NONE)
    refd = ref_dgrad(dz, wb64, pts, dims)
    gotd = gather_rows(out, pts)
    assert (np.abs(gotd - refd) <= 2.0 ** -8 * np.abs(refd) + 2e-5 * np.abs(refd).max()).all(), "bf16 dgrad"


def test_loss_metric_is_run_to_run_identical(fdn):
    """Reported loss / metric sums go through per-block partials added in a fixed order (no float atomics)."""
    ops = fdn.ops
    g = torch.Generator(device="cuda").manual_seed(3)
    N, V = 8, 48 ** 3
    pred = torch.randn((N, V, 3), device="cuda", generator=g)
    t = [torch.randn((N, V), device="cuda", generator=g) for _ in range(3)]
    mask = (torch.rand((N, V), device="cuda", generator=g) < 0.12).float()
    outs = [ops.loss_metrics(pred, t[0], t[1], t[2], mask)[0].clone() for _ in range(4)]
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    w = torch.randn(3_000_001, device="cuda", generator=g)
    isk = (torch.rand(3_000_001, device="cuda", generator=g) < 0.9).to(torch.uint8)
    s = [ops.l2_sumsq(w, isk).clone() for _ in range(3)]
    assert torch.equal(s[0], s[1]) and torch.equal(s[0], s[2])
    assert abs(float(s[0]) - float((w.double() ** 2 * isk.double()).sum())) <= 1e-5 * float(s[0])


@pytest.mark.parametrize("N,P", [(8, 48), (8, 24)])
def test_hot_kernels_are_run_to_run_identical_at_bench_shapes(fdn, N, P):
    """No atomics and no schedule-dependent summation order in the three hot kernels: forward, fused dgrad (+ border fold) and the weight
    gradient give bit-identical results launch after launch at the bench grids (a race in the LDS pipelines would show up here)."""
    ops = fdn.ops
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn((N, P, P, P, 64), device="cuda", generator=g)
    dz = torch.randn((N, P, P, P, 64), device="cuda", generator=g)
    w = torch.randn((3, 3, 3, 64, 64), device="cuda", generator=g) * 0.03
    wf, wd = ops.pack_conv64_weights(w)
    ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1, device="cuda")
    pad = torch.empty((N, P + 2, P + 2, P + 2, 64), device="cuda")
    ref = None
    for _ in range(4):
        y = ops.conv3d_fwd(x, w, None, ops.ACT_LEAKY, 0.2, dz, wpack=wf)
        out = torch.empty_like(x)
        pad.fill_(float("nan"))                      # the surface scratch must be fully rewritten by every launch
        ops.conv3d_dgrad_fused(dz, wd, pad, out, skip=x, y_prev=y, act=ops.ACT_LEAKY)
        ops.fold_halo_border([pad], out, x, y, ops.ACT_LEAKY)
        dw, _ = ops.conv3d_wgrad(x, dz, 3, 64, 64, workspace=ws)
        cur = (y, out, dw.clone())
        assert all(bool(torch.isfinite(t).all()) for t in cur)
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, cur))
