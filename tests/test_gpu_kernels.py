"""Parity of every HIP entry point against the CPU oracle, called through the C-ABI (ctypes) on a real MI355X.
Tolerance: north_star asks <= 1e-3 relative fp32; the kernels are exact-fp32 FMA chains, so we check 2e-5
of the output scale (max |ref|)."""
import contextlib

import numpy as np
import pytest
import torch

from oracle import flownet_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 2e-5


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")


def close(got, ref, tol=RTOL, name=""):
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max() / scale
    assert err <= tol, "%s: max err %.3e of scale %.3e" % (name, err, scale)


@pytest.fixture(scope="module")
def ops(fdn):
    return fdn.ops


SHAPES = [(2, 8, 8, 8), (1, 5, 7, 9), (1, 10, 12, 16), (3, 4, 4, 2)]
# W a multiple of 4 -> the planner (variant 0) takes the Winograd F(4,3) kernel; forced variants 1..6 stay on the direct
# kernel, 7 forces Winograd.  Shapes cover a single partial tile, ragged tile grids in every dimension, W = 4, and a
# multi-tile launch (24^3, the cfg2 low-res grid).
WINO_SHAPES = [(1, 5, 7, 12), (3, 4, 4, 4), (2, 9, 3, 24), (1, 1, 1, 4), (1, 17, 10, 8), (2, 24, 24, 24)]


@contextlib.contextmanager
def variant_lib(fdn, variant):
    """variant == 0: run on the product library (yields None).  Otherwise route the operator layer through the TEST build
    (lib4dflow_hip_test.so: same sources + fdn_debug_* hooks) so a kernel variant the planner would not pick at this size
    can be forced; the product library exports no such switches."""
    if not variant:
        yield None
    else:
        with fdn._lib.test_build() as lib:
            yield lib


@pytest.mark.parametrize("shape,mt", [(sh, mt) for sh in SHAPES for mt in (0, 1, 2, 3, 4, 5, 6)] +
                         [(sh, mt) for sh in WINO_SHAPES for mt in (0, 7, 3)])   # 0 = planner, 1..6 = forced <MT,NW,CS>, 7 = Winograd
def test_conv64_fwd(ops, fdn, shape, mt):
    rng = np.random.default_rng(1)
    N, D, H, W = shape
    x = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    res = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    with variant_lib(fdn, mt) as lib:              # mt == 0: the product library, planner's choice
        if lib is not None:
            lib.fdn_debug_set_conv64_mt(mt)
        try:
            for act, bias, r in [(O.ACT_RELU, b, None), (O.ACT_LEAKY, None, res), (O.ACT_NONE, None, None)]:
                ref = O.conv3d_fwd(x.astype(np.float64), w.astype(np.float64), None if bias is None else bias.astype(np.float64),
                                   act, 0.2, None if r is None else r.astype(np.float64))
                got = ops.conv3d_fwd(dev(x), dev(w), None if bias is None else dev(bias), act, 0.2,
                                     None if r is None else dev(r))
                close(got, ref, name="conv64 fwd act=%d" % act)
        finally:
            if lib is not None:
                lib.fdn_debug_set_conv64_mt(0)


# H even and W a multiple of 4 -> FDN_ALGO_AUTO takes the 2-D Winograd kernel (F(2,3) along H x F(4,3) along W, conv64_wino2d.hip).
# Shapes: one partial tile, ragged tile grids in d / h / w, the smallest grid (1,2,4), H = 2, a multi-tile launch at the cfg2 low-res grid.
# H a multiple of 4 as well -> F(4,3) along H (round 5; cells of 4 x 4 voxels): (1,5,8,12), (3,4,4,4), (2,24,24,24), a single cell (1,1,4,4),
# ragged cell-row / cell-column tile grids (1,7,12,20), (1,19,20,12), (2,3,28,4).
WINO2D_SHAPES = [(1, 5, 8, 12), (3, 4, 4, 4), (2, 9, 2, 24), (1, 1, 2, 4), (1, 17, 10, 8), (2, 24, 24, 24), (1, 3, 6, 20), (1, 11, 14, 28),
                 (1, 1, 4, 4), (1, 7, 12, 20), (1, 19, 20, 12), (2, 3, 28, 4)]


@pytest.mark.parametrize("shape,tile", [(sh, 0) for sh in WINO2D_SHAPES] +
                         [((2, 24, 24, 24), t) for t in (8 | 1 << 8 | 4 << 16, 8 | 4 << 8 | 1 << 16, 16 | 2 << 8 | 1 << 16, 5 | 2 << 8 | 2 << 16, 32 | 1 << 8 | 1 << 16)] +
                         [((1, 17, 10, 8), 3 | 1 << 8 | 2 << 16), ((1, 19, 20, 12), 3 | 2 << 8 | 1 << 16), ((1, 19, 20, 12), 30 | 1 << 8 | 1 << 16)])
def test_conv64_fwd_wino2d(ops, fdn, shape, tile):
    """2-D Winograd forward == oracle, with every epilogue variant; == the 1-D Winograd and direct kernels to fp32 rounding.
    tile != 0: a forced (td, ch, cw) tile through the test build (tiles the planner would not pick at this size)."""
    rng = np.random.default_rng(11)
    N, D, H, W = shape
    x = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    res = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    with variant_lib(fdn, tile) as lib:
        if lib is not None:
            lib.fdn_debug_set_conv64_wino2d_tile(tile)
        try:
            for act, bias, r in [(O.ACT_RELU, b, None), (O.ACT_LEAKY, None, res), (O.ACT_NONE, None, None)]:
                ref = O.conv3d_fwd(x.astype(np.float64), w.astype(np.float64), None if bias is None else bias.astype(np.float64),
                                   act, 0.2, None if r is None else r.astype(np.float64))
                got = ops.conv3d_fwd(dev(x), dev(w), None if bias is None else dev(bias), act, 0.2,
                                     None if r is None else dev(r), algo=ops.ALGO_AUTO)
                close(got, ref, name="conv64 fwd 2-D winograd act=%d" % act)
                for algo in (ops.ALGO_WINO_H2, ops.ALGO_WINO_W, ops.ALGO_DIRECT, ops.ALGO_WINO_BF16X3):
                    other = ops.conv3d_fwd(dev(x), dev(w), None if bias is None else dev(bias), act, 0.2,
                                           None if r is None else dev(r), algo=algo)
                    close(got, other.cpu().numpy(), name="2-D winograd vs algo %d" % algo)
                    if algo == ops.ALGO_WINO_BF16X3:         # the bf16 x 3 products against the oracle itself (F(4,3) x F(4,3) grids; else = AUTO)
                        close(other, ref, name="conv64 fwd 2-D winograd, bf16 x 3 products, act=%d" % act)
        finally:
            if lib is not None:
                lib.fdn_debug_set_conv64_wino2d_tile(0)


@pytest.mark.parametrize("shape", [(1, 5, 6, 6), (2, 7, 10, 14), (1, 1, 2, 2)])
def test_conv64_dgrad_padded_wino2d(ops, shape):
    """Zero-boundary mode of the 2-D kernel: the plain padded-grid dgrad (D+2, H+2, W+2 with H+2 even, W+2 a multiple of 4) + fold."""
    rng = np.random.default_rng(12)
    N, D, H, W = shape
    dz = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    dx = O.conv3d_dgrad(dz.astype(np.float64), w.astype(np.float64), (N, D, H, W, 64))
    pad = ops.conv3d_dgrad(dev(dz), dev(w))
    close(ops.fold_halo([pad]), dx, name="padded dgrad (2-D winograd) + fold")
    pad1 = ops.conv3d_dgrad(dev(dz), dev(w), algo=ops.ALGO_DIRECT)
    close(pad, pad1.cpu().numpy(), name="padded dgrad 2-D winograd vs direct")


@pytest.mark.parametrize("shape", SHAPES)
def test_conv64_dgrad_and_fold(ops, shape):
    rng = np.random.default_rng(2)
    N, D, H, W = shape
    dz = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    y = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    skip = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    dx = O.conv3d_dgrad(dz.astype(np.float64), w.astype(np.float64), (N, D, H, W, 64))
    pad = ops.conv3d_dgrad(dev(dz), dev(w))
    close(ops.fold_halo([pad]), dx, name="dgrad+fold")
    ref = O.act_bwd_from_output(dx + skip, y, O.ACT_LEAKY)
    close(ops.fold_halo([pad], dev(skip), dev(y), O.ACT_LEAKY, 0.2), ref, name="fold skip+leaky")
    ref3 = O.act_bwd_from_output(3 * dx, y, O.ACT_RELU)
    close(ops.fold_halo([pad, pad, pad], None, dev(y), O.ACT_RELU), ref3, name="fold 3 src+relu")


@pytest.mark.parametrize("shape,layout", [(sh, lo) for sh in SHAPES + [(1, 1, 1, 1), (1, 2, 3, 1)] for lo in (0, 1, 2, 3, 4, 5, 6)] +
                         [(sh, lo) for sh in WINO_SHAPES for lo in (0, 7, 5, 8)] +
                         [(sh, 0) for sh in [(1, 5, 8, 12), (2, 9, 2, 24), (1, 1, 2, 4), (1, 3, 6, 20), (1, 11, 14, 28), (1, 2, 4, 4)]])   # 2-D Winograd inner box
def test_conv64_dgrad_fused_fold(ops, fdn, shape, layout):
    """dgrad with the interior fold in the conv epilogue + border kernel == oracle dgrad (+ skip, * act').
    layout 0 = product library (H even, W % 4 == 0: inner box on the 2-D Winograd kernel + the shell faces as a 1-D Winograd launch;
    W % 4 == 0 only: one 1-D Winograd launch incl. the w-face region), 7 = forced 1-D Winograd (test build), 8 = Winograd with the
    w faces on the separate direct-kernel launch (the round-2 path, kept as a test-build switch), 1..6 = direct layouts."""
    wface_direct = layout == 8
    if wface_direct:
        layout = 7
    rng = np.random.default_rng(6)
    N, D, H, W = shape
    dz = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    y = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    skip = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    dx = O.conv3d_dgrad(dz.astype(np.float64), w.astype(np.float64), (N, D, H, W, 64))
    _, wd = ops.pack_conv64_weights(dev(w))
    with variant_lib(fdn, layout) as lib:          # layout == 0: the product library (planner, shell slabs)
        if lib is not None:
            lib.fdn_debug_set_conv64_mt(layout)
            lib.fdn_debug_set_conv64_wface_direct(1 if wface_direct else 0)
            # odd layouts also exercise the single padded-grid launch; the default is inner box + six 9-tap shell slabs
            lib.fdn_debug_set_conv64_shell_slabs(0 if layout in (1, 3, 5) else 1)
        try:
            pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
            out = torch.full((N, D, H, W, 64), float("nan"), device="cuda")
            ops.conv3d_dgrad_fused(dev(dz), wd, pad, out, skip=dev(skip), y_prev=dev(y), act=O.ACT_LEAKY)
            ops.fold_halo_border([pad], out, dev(skip), dev(y), O.ACT_LEAKY)
            close(out, O.act_bwd_from_output(dx + skip, y, O.ACT_LEAKY), name="fused dgrad+border")
            if layout == 0 and H % 4 == 0 and W % 4 == 0:    # FDN_ALGO_AUTO ran F(4,3) along H: the F(2,3)-along-H inner box (round 4) beside it
                pad.fill_(float("nan")); out.fill_(float("nan"))
                ops.conv3d_dgrad_fused(dev(dz), wd, pad, out, skip=dev(skip), y_prev=dev(y), act=O.ACT_LEAKY, algo=ops.ALGO_WINO_H2)
                ops.fold_halo_border([pad], out, dev(skip), dev(y), O.ACT_LEAKY)
                close(out, O.act_bwd_from_output(dx + skip, y, O.ACT_LEAKY), name="fused dgrad+border, F(2,3) along H")
                pad.fill_(float("nan")); out.fill_(float("nan"))  # and the F(4,3) x F(4,3) inner box with the bf16 x 3 products
                ops.conv3d_dgrad_fused(dev(dz), wd, pad, out, skip=dev(skip), y_prev=dev(y), act=O.ACT_LEAKY, algo=ops.ALGO_WINO_BF16X3)
                ops.fold_halo_border([pad], out, dev(skip), dev(y), O.ACT_LEAKY)
                close(out, O.act_bwd_from_output(dx + skip, y, O.ACT_LEAKY), name="fused dgrad+border, bf16 x 3 products")
            # fan-in of three consumers chained through the output buffer (skip aliases out), mask on the last
            acc = torch.full((N, D, H, W, 64), float("nan"), device="cuda")
            pads = [torch.empty_like(pad) for _ in range(3)]
            for k in range(3):
                ops.conv3d_dgrad_fused(dev(dz * (k + 1)), wd, pads[k], acc, skip=acc if k else None,
                                       y_prev=dev(y) if k == 2 else None, act=O.ACT_RELU if k == 2 else O.ACT_NONE)
            ops.fold_halo_border(pads, acc, None, dev(y), O.ACT_RELU)
            close(acc, O.act_bwd_from_output(6 * dx, y, O.ACT_RELU), name="fused fan-in of 3")
        finally:
            if lib is not None:
                lib.fdn_debug_set_conv64_mt(0)
                lib.fdn_debug_set_conv64_shell_slabs(1)
                lib.fdn_debug_set_conv64_wface_direct(0)


@pytest.mark.parametrize("shape,nl", [((2, 8, 12, 16), 3), ((8, 24, 24, 24), 8), ((1, 2, 5, 4), 2), ((2, 6, 6, 8), 11), ((1, 7, 6, 8), 3), ((1, 4, 6, 10), 2), ((4, 24, 24, 24), 11), ((2, 12, 12, 24), 14)])
def test_conv64_wgrad_batch(ops, shape, nl):
    """fdn_conv3d_wgrad_batch: the weight (and bias) gradients of nl layers of one grid in ONE launch == the float64 oracle, and == nl
    calls of fdn_conv3d_wgrad (different split of the voxel sum: equal to fp32 rounding; the fall-back shapes -- odd D, W % 4 != 0 --
    loop over the single-layer path and are bit-identical)."""
    N, D, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(nl * 1000 + D)
    xs = [torch.randn((N, D, H, W, 64), device="cuda", generator=g) for _ in range(nl)]
    dzs = [torch.randn((N, D, H, W, 64), device="cuda", generator=g) for _ in range(nl)]
    dws = [torch.full((3, 3, 3, 64, 64), float("nan"), device="cuda") for _ in range(nl)]
    dbs = [torch.full((64,), float("nan"), device="cuda") if i % 2 == 0 else None for i in range(nl)]
    ops.conv3d_wgrad_batch(xs, dzs, dws, dbs)
    batched = D % 2 == 0 and W % 4 == 0
    for i in range(nl):
        one_w, one_b = ops.conv3d_wgrad(xs[i], dzs[i], 3, 64, 64, want_bias=True)
        scale = one_w.abs().max().item()
        if batched:
            assert (dws[i] - one_w).abs().max().item() <= 2e-5 * scale, i      # two splits of an fp32 sum over N D H W voxels (measured 4e-6 at 8 x 24^3)
        else:
            assert torch.equal(dws[i], one_w), i
        if dbs[i] is not None:
            assert torch.equal(dbs[i], one_b), i
    if N * D * H * W <= 4096:                      # the float64 oracle on the small cases (first and last layer)
        for i in (0, nl - 1):
            ref = O.conv3d_wgrad(xs[i].cpu().numpy().astype(np.float64), dzs[i].cpu().numpy().astype(np.float64), 3)
            close(dws[i], ref, name="batched wgrad layer %d" % i)
    with pytest.raises(Exception):
        ops.conv3d_wgrad_batch(xs, dzs[:-1], dws)                      # ragged lists
    with pytest.raises(Exception):
        ops.conv3d_wgrad_batch(xs, dzs, dws, workspace=torch.empty(16, device="cuda"))      # workspace too small: loud


@pytest.mark.parametrize("shape,algo", [(sh, 0) for sh in [(1, 5, 8, 12), (2, 9, 2, 24), (1, 1, 2, 4), (3, 24, 24, 24), (1, 11, 14, 28), (1, 7, 12, 20), (1, 1, 4, 4)]] +
                         [(sh, 3) for sh in [(1, 5, 8, 12), (3, 24, 24, 24)]] +      # algo 3 = FDN_ALGO_WINO_H2 where AUTO takes F(4,3) along H
                         [(sh, 4) for sh in [(1, 5, 8, 12), (3, 24, 24, 24), (1, 7, 12, 20)]])   # algo 4 = FDN_ALGO_WINO_BF16X3: the same inner box with the bf16 x 3 products
def test_conv64_dgrad_fused_one_launch_equals_two(ops, fdn, shape, algo):
    """The fused dgrad of the 2-D Winograd path is ONE launch (conv64_wino2d_shell_kernel: inner box on the 2-D body, shell faces behind
    it on the 1-D body).  It must be bit-identical to the same two bodies as two launches (test-build switch), and to the two `parts` a
    caller may issue on its own."""
    rng = np.random.default_rng(61)
    N, D, H, W = shape
    dz = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    y = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    skip = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    _, wd = ops.pack_conv64_weights(dev(w))

    def run(**kw):
        pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
        out = torch.full((N, D, H, W, 64), float("nan"), device="cuda")
        if kw.get("two_parts"):
            ops.conv3d_dgrad_fused(dz, wd, pad, out, skip=skip, y_prev=y, act=O.ACT_LEAKY, parts=ops.DGRAD_SHELL, algo=algo)
            ops.conv3d_dgrad_fused(dz, wd, pad, out, skip=skip, y_prev=y, act=O.ACT_LEAKY, parts=ops.DGRAD_INNER, algo=algo)
        else:
            ops.conv3d_dgrad_fused(dz, wd, pad, out, skip=skip, y_prev=y, act=O.ACT_LEAKY, algo=algo)
        return pad.clone(), out.clone()          # (before the border fold: the raw products of the launch(es))
    pad1, out1 = run()
    pad2, out2 = run(two_parts=True)
    with fdn._lib.test_build() as lib:
        lib.fdn_debug_set_conv64_split_dgrad(1)
        try:
            pad3, out3 = run()
        finally:
            lib.fdn_debug_set_conv64_split_dgrad(0)
    for name, (p_, o_) in (("parts", (pad2, out2)), ("two launches", (pad3, out3))):
        assert torch.equal(torch.nan_to_num(pad1, nan=-7.0), torch.nan_to_num(p_, nan=-7.0)), name + ": padded scratch differs"
        assert torch.equal(torch.nan_to_num(out1, nan=-7.0), torch.nan_to_num(o_, nan=-7.0)), name + ": interior differs"


@pytest.mark.parametrize("shape", SHAPES + [(2, 16, 16, 16), (1, 1, 1, 1), (1, 2, 3, 1), (1, 3, 20, 33), (5, 9, 8, 24), (2, 24, 24, 24),
                                            (1, 7, 6, 8), (1, 2, 13, 7)])
@pytest.mark.parametrize("direct", [0, 1, 2])    # 0 = product library, FDN_ALGO_AUTO: Winograd F(3,4) along W, + F(3,2) along D when D is even;
def test_conv64_wgrad(ops, fdn, shape, direct):  # 1 = the direct kernel (test build); 2 = FDN_ALGO_WINO_W: the W-only Winograd kernel
    rng = np.random.default_rng(3)
    N, D, H, W = shape
    x = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    dz = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    ref = O.conv3d_wgrad(x.astype(np.float64), dz.astype(np.float64), 3)
    with variant_lib(fdn, direct == 1) as lib:
        if lib is not None:
            lib.fdn_debug_set_wgrad64_direct(1)
        try:
            dw, db = ops.conv3d_wgrad(dev(x), dev(dz), 3, 64, 64, want_bias=True, algo=ops.ALGO_WINO_W if direct == 2 else ops.ALGO_AUTO)
        finally:
            if lib is not None:
                lib.fdn_debug_set_wgrad64_direct(0)
    close(dw, ref, name="wgrad64")
    close(db, O.bias_grad(dz.astype(np.float64)), name="bias grad 64")


@pytest.mark.parametrize("shape,valu", [((2, 6, 6, 6), 0), ((1, 5, 7, 9), 0), ((1, 1, 2, 3), 0), ((1, 4, 5, 8), 0), ((2, 12, 10, 24), 0),
                                        ((2, 6, 6, 6), 1), ((1, 5, 7, 9), 1)])
def test_thin_layers(ops, fdn, shape, valu):
    """valu = 1: the VALU 3 -> 64 kernels (test build); 0: the product library (im2col MFMA kernels)."""
    with variant_lib(fdn, valu) as lib:
        if lib is not None:
            lib.fdn_debug_set_cin3_mfma(0)
        try:
            _thin_layers(ops, shape)
        finally:
            if lib is not None:
                lib.fdn_debug_set_cin3_mfma(1)


def _thin_layers(ops, shape):
    rng = np.random.default_rng(4)
    N, D, H, W = shape
    f64 = lambda a: a.astype(np.float64)
    # 3 -> 64
    x3 = rng.normal(size=(N, D, H, W, 3)).astype(np.float32)
    w3 = (rng.normal(size=(3, 3, 3, 3, 64)) * 0.2).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    close(ops.conv3d_fwd(dev(x3), dev(w3), dev(b), O.ACT_RELU), O.conv3d_fwd(f64(x3), f64(w3), f64(b), O.ACT_RELU), name="3->64 fwd")
    dz = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    dw, db = ops.conv3d_wgrad(dev(x3), dev(dz), 3, 3, 64, want_bias=True)
    close(dw, O.conv3d_wgrad(f64(x3), f64(dz), 3), name="3->64 wgrad")
    close(db, O.bias_grad(f64(dz)), name="3->64 bias grad")
    # 64 -> 1 writing into channel 1 of an (N,V,3) tensor
    x = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    w1 = (rng.normal(size=(3, 3, 3, 64, 1)) * 0.1).astype(np.float32)
    b1 = rng.normal(size=1).astype(np.float32)
    out = torch.zeros((N, D, H, W, 3), device="cuda")
    ops.conv3d_fwd(dev(x), dev(w1), dev(b1), O.ACT_NONE, out=out, ldy=3, y_coff=1)
    ref = np.zeros((N, D, H, W, 3)); ref[..., 1:2] = O.conv3d_fwd(f64(x), f64(w1), f64(b1))
    close(out, ref, name="64->1 fwd")
    dpred = rng.normal(size=(N, D, H, W, 3)).astype(np.float32)
    dzo = f64(dpred[..., 1:2])
    pad = ops.conv3d_dgrad(dev(dpred), dev(w1), lddz=3, dz_coff=1)
    close(ops.fold_halo([pad]), O.conv3d_dgrad(dzo, f64(w1), x.shape), name="64->1 dgrad")
    ymask = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    close(ops.conv_cout1_dgrad_folded(dev(dpred), dev(w1), (N, D, H, W), dev(ymask), O.ACT_RELU, lddz=3, dz_coff=1),
          O.conv3d_dgrad(dzo, f64(w1), x.shape) * (ymask > 0), name="64->1 dgrad folded+relu")
    close(ops.conv_cout1_dgrad_folded(dev(dpred), dev(w1), (N, D, H, W), lddz=3, dz_coff=1),
          O.conv3d_dgrad(dzo, f64(w1), x.shape), name="64->1 dgrad folded")
    dbp = torch.full((64,), float("nan"), device="cuda")
    ops.conv_cout1_dgrad_folded(dev(dpred), dev(w1), (N, D, H, W), dev(ymask), O.ACT_RELU, lddz=3, dz_coff=1, dbias_prev=dbp)
    close(dbp, O.bias_grad(O.conv3d_dgrad(dzo, f64(w1), x.shape) * (ymask > 0)), name="64->1 dgrad folded: fused bias grad")
    dw, db = ops.conv3d_wgrad(dev(x), dev(dpred), 3, 64, 1, want_bias=True, lddz=3, dz_coff=1)
    close(dw, O.conv3d_wgrad(f64(x), dzo, 3), name="64->1 wgrad")
    close(db, O.bias_grad(dzo), name="64->1 bias grad")
    # 1x1x1 (64+64) -> 64
    xa = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    xb = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    wk = (rng.normal(size=(1, 1, 1, 128, 64)) * 0.1).astype(np.float32)
    cat = np.concatenate([xa, xb], -1)
    close(ops.conv3d_fwd(dev(xa), dev(wk), dev(b), O.ACT_RELU, x2=dev(xb)), O.conv3d_fwd(f64(cat), f64(wk), f64(b), O.ACT_RELU), name="1x1 fwd")
    dcat = O.conv3d_dgrad(f64(dz), f64(wk), cat.shape)
    da, dbb = ops.conv1x1_dgrad(dev(dz), dev(wk), dev(xa), dev(xb))
    close(da, dcat[..., :64] * (xa > 0), name="1x1 dgrad a")
    close(dbb, dcat[..., 64:] * (xb > 0), name="1x1 dgrad b")
    dw, db = ops.conv3d_wgrad(dev(xa), dev(dz), 1, 128, 64, x2=dev(xb), want_bias=True)
    close(dw, O.conv3d_wgrad(f64(cat), f64(dz), 1), name="1x1 wgrad")


@pytest.mark.parametrize("R", [2, 3, 4])
def test_upsample(ops, R):
    rng = np.random.default_rng(5)
    x = rng.normal(size=(2, 5, 4, 6, 64)).astype(np.float32)
    ref = O.upsample_trilinear_fwd(x.astype(np.float64), R, f32_coeffs=True)
    close(ops.upsample_trilinear_fwd(dev(x), R), ref, name="upsample fwd")
    # against exact (float64-coefficient) trilinear the fp32 coefficient rule differs by ~1e-6
    close(ops.upsample_trilinear_fwd(dev(x), R), O.upsample_trilinear_fwd(x.astype(np.float64), R), tol=1e-5, name="upsample fwd exact")
    dy = rng.normal(size=ref.shape).astype(np.float32)
    y = rng.normal(size=x.shape).astype(np.float32)
    refb = O.upsample_trilinear_bwd(dy.astype(np.float64), x.shape[1:4], R, f32_coeffs=True)
    close(ops.upsample_trilinear_bwd(dev(dy), R), refb, name="upsample bwd")
    close(ops.upsample_trilinear_bwd(dev(dy), R, dev(y), O.ACT_LEAKY, 0.2), O.act_bwd_from_output(refb, y, O.ACT_LEAKY), name="upsample bwd+mask")


@pytest.mark.parametrize("shape,R", [((1, 3, 7, 5), 2), ((2, 1, 5, 4), 3), ((1, 4, 1, 3), 2), ((1, 6, 9, 8), 4), ((1, 3, 3, 2), 5)])
def test_upsample_bwd_row_blocks(ops, fdn, shape, R):
    """upsample_bwd_kernel folds the high-res rows of TWO low-res rows per block (round 6): odd H leaves a one-row block, an axis of
    length 1 takes every high-res row; against the oracle, and one / two rows per block bit for bit (the same terms in the same order)."""
    rng = np.random.default_rng(15)
    N, D, H, W = shape
    dy = rng.normal(size=(N, D * R, H * R, W * R, 64)).astype(np.float32)
    y = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    refb = O.upsample_trilinear_bwd(dy.astype(np.float64), (D, H, W), R, f32_coeffs=True)
    got = ops.upsample_trilinear_bwd(dev(dy), R, dev(y), O.ACT_LEAKY, 0.2)
    close(got, O.act_bwd_from_output(refb, y, O.ACT_LEAKY), name="upsample bwd+mask")
    with fdn._lib.test_build() as lib:
        outs = []
        try:
            for hb in (1, 2):
                lib.fdn_debug_set_upsample_bwd_hb(hb)
                outs.append(ops.upsample_trilinear_bwd(dev(dy), R, dev(y), O.ACT_LEAKY, 0.2))
        finally:
            lib.fdn_debug_set_upsample_bwd_hb(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], got)


def test_input_features_loss_metric(ops):
    B, P, R = 2, 6, 2
    batch = O.synthetic_batch(B, P, R, seed=11)
    u, v, w, mu, mv, mw, uh, vh, wh, venc, mask = batch
    ph, pc = ops.input_features(*[dev(a) for a in batch[:6]])
    rph, rpc = O.input_features(*[a.astype(np.float64) for a in batch[:6]])
    close(ph, rph, name="phase"); close(pc, rpc, name="pc")
    rng = np.random.default_rng(12)
    H = P * R
    pred = rng.uniform(-0.5, 0.5, size=(B, H, H, H, 3)).astype(np.float32)
    # exercise the metric's branches: exact zeros in target and exact hits
    uh[0, 0, 0, :4] = 0; vh[0, 0, 0, :4] = 0; wh[0, 0, 0, :4] = 0; mask[0, 0, 0, :4] = 1
    hires = np.concatenate([uh, vh, wh], -1).astype(np.float64)
    loss, dpred = O.masked_mse_loss_fwd_bwd(pred.astype(np.float64), hires, mask.astype(np.float64))
    rel = O.relative_error(pred.astype(np.float32), hires.astype(np.float32), mask)
    out, dp = ops.loss_metrics(dev(pred), dev(uh), dev(vh), dev(wh), dev(mask))
    close(out[:, 0], loss, tol=1e-5, name="loss")
    close(out[:, 1], rel, tol=2e-3, name="rel err (rounding to 1e-4 steps)")
    close(out[:, 2], mask.sum((1, 2, 3)), name="sum mask")
    close(dp, dpred, name="dpred")
    out2, none = ops.loss_metrics(dev(pred), dev(uh), dev(vh), dev(wh), dev(mask), want_grad=False)
    assert none is None
    close(out2[:, 0], loss, tol=1e-5, name="loss (no grad)")


def test_adam_and_l2(ops):
    rng = np.random.default_rng(13)
    n = 10007
    w = rng.normal(size=n).astype(np.float32); g = rng.normal(size=n).astype(np.float32)
    isk = (rng.uniform(size=n) < 0.8)
    m = np.zeros(n); v = np.zeros(n); wr = w.astype(np.float64).copy()
    dw, dm, dv, dk = dev(w), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.tensor(isk.astype(np.uint8), device="cuda")
    close(ops.l2_sumsq(dw, dk), [float((w[isk].astype(np.float64) ** 2).sum())], tol=1e-5, name="l2 sumsq")
    l2s = 8 * 2 * O.L2_LAMBDA
    for t in range(1, 4):
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        O.adam_step_tf(wr, g.astype(np.float64) + l2s * wr * isk, m, v, t, 1e-3)
        if t == 2:      # the device-side scale (global batch carried through the all-reduce) is equivalent
            ops.adam_step(dw, dev(g), dm, dv, dk, lr_t, 0.9, 0.999, 1e-7, 2 * O.L2_LAMBDA, torch.tensor([8.0], device="cuda"))
        else:
            ops.adam_step(dw, dev(g), dm, dv, dk, lr_t, 0.9, 0.999, 1e-7, l2s)
    # the step can leave sum(w_new^2) of the kernel parameters behind (per-block partials, summed in a fixed order): equal to a
    # fresh fdn_l2_sumsq pass over the updated parameters, and run-to-run identical
    part = torch.full((ops.ADAM_PARTIALS,), float("nan"), device="cuda")
    w_before = dw.clone()
    ops.adam_step(dw, dev(g), dm.clone(), dv.clone(), dk, 1e-3, 0.9, 0.999, 1e-7, l2s, sumsq_partials=part)
    s1 = ops.sum_partials(part).clone()
    close(s1, ops.l2_sumsq(dw, dk).cpu().numpy(), tol=1e-6, name="adam sumsq partials")
    part2 = torch.zeros_like(part)
    dw2 = w_before.clone()
    ops.adam_step(dw2, dev(g), dm.clone(), dv.clone(), dk, 1e-3, 0.9, 0.999, 1e-7, l2s, sumsq_partials=part2)
    assert torch.equal(ops.sum_partials(part2), s1) and torch.equal(dw2, dw)
    dw = w_before
    # (1-b2) evaluated in fp32 (as Keras does for fp32 variables) is off by 1.3e-5 relative from the float64 oracle
    close(dw, wr, tol=5e-5, name="adam w"); close(dm, m, tol=1e-5, name="adam m"); close(dv, v, tol=5e-5, name="adam v")


@pytest.mark.parametrize("shape", [(2, 8, 8, 8), (1, 5, 7, 9), (1, 6, 4, 12)])
def test_conv64_dgrad_fused_parts_equal_whole(ops, shape):
    """FDN_DGRAD_INNER and FDN_DGRAD_SHELL issued separately (on two streams) write exactly what the single call writes."""
    rng = np.random.default_rng(8)
    N, D, H, W = shape
    dz = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    y = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    skip = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    _, wd = ops.pack_conv64_weights(dev((rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)))
    pad0 = torch.zeros((N, D + 2, H + 2, W + 2, 64), device="cuda"); out0 = torch.zeros((N, D, H, W, 64), device="cuda")
    ops.conv3d_dgrad_fused(dz, wd, pad0, out0, skip=skip, y_prev=y, act=O.ACT_LEAKY)
    pad1 = torch.zeros_like(pad0); out1 = torch.zeros_like(out0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.conv3d_dgrad_fused(dz, wd, pad1, out1, parts=ops.DGRAD_SHELL)
    ops.conv3d_dgrad_fused(dz, wd, pad1, out1, skip=skip, y_prev=y, act=O.ACT_LEAKY, parts=ops.DGRAD_INNER)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(pad0, pad1) and torch.equal(out0, out1)


def test_pack_batch_equals_per_layer_pack(ops):
    """fdn_pack_conv64_weights_batch (all 64->64 layers of the flat parameter buffer in one launch) == per-layer packing."""
    rng = np.random.default_rng(17)
    nl, sz = 3, 27 * 64 * 64
    gaps = [5, 64, 0]                                   # biases / other layers between the kernels in the flat buffer
    offs, pos = [], 7
    for gp in gaps:
        offs.append(pos); pos += sz + gp
    flat = dev(rng.normal(size=pos).astype(np.float32))
    packs = torch.full((nl, 2, ops.CONV64_PACK_FLOATS), float("nan"), device="cuda")
    ops.pack_conv64_weights_batch(flat, torch.tensor(offs, device="cuda", dtype=torch.int64), packs)
    for i, o in enumerate(offs):
        wf, wd = ops.pack_conv64_weights(flat[o:o + sz].view(3, 3, 3, 64, 64))
        assert torch.equal(packs[i, 0], wf) and torch.equal(packs[i, 1], wd)


def test_errors_are_loud(ops, fdn):
    x = torch.zeros((1, 4, 4, 4, 5), device="cuda")
    w = torch.zeros((3, 3, 3, 5, 7), device="cuda")
    with pytest.raises(fdn.FdnError):
        ops.conv3d_fwd(x, w)                        # unsupported shape
    with pytest.raises(fdn.FdnError):
        ops.conv3d_fwd(torch.zeros((1, 4, 4, 4, 64)), torch.zeros((3, 3, 3, 64, 64)))   # CPU tensors: no fallback


def test_winograd_kernels_equal_direct_kernels_on_random_shapes(ops, fdn):
    """Tile / region logic sweep: on 24 random shapes (W a multiple of 4 for the conv, any W for the weight gradient) the
    Winograd kernels of the product library must reproduce the direct MFMA kernels (test build) to fp32 accuracy -- forward
    with bias + residual + LeakyReLU, fused dgrad + border fold with skip and act', weight gradient."""
    rng = np.random.default_rng(2024)
    shapes = [(int(rng.integers(1, 4)), int(rng.integers(1, 21)), int(rng.integers(1, 21)), 4 * int(rng.integers(1, 9))) for _ in range(20)]
    shapes += [(1, 1, 1, 4), (2, 2, 19, 4), (1, 20, 1, 32), (1, 9, 9, 28)]
    # round 6: grids off the multiple-of-4 raster -- the aligned box on F(4,3) x F(4,3), the remainder strips on the direct kernel (W % 4 != 0
    # with any H; a w strip only, an h strip as well, odd H), and grids too small for the split (all direct)
    shapes += [(30, 8, 10, 10), (8, 10, 18, 18), (6, 9, 22, 22), (40, 6, 9, 13), (200, 5, 5, 5), (2, 6, 8, 10), (1, 2, 12, 7), (1, 3, 6, 6), (5, 18, 18, 18)]
    for (N, D, H, W) in shapes:
        g = torch.Generator(device="cuda").manual_seed(N * 1000003 + D * 1009 + H * 31 + W)
        x = torch.randn((N, D, H, W, 64), device="cuda", generator=g)
        res = torch.randn((N, D, H, W, 64), device="cuda", generator=g)
        dz = torch.randn((N, D, H, W, 64), device="cuda", generator=g)
        w = torch.randn((3, 3, 3, 64, 64), device="cuda", generator=g) * 0.05
        b = torch.randn((64,), device="cuda", generator=g)
        yfix = torch.randn((N, D, H, W, 64), device="cuda", generator=g)   # act' mask: fixed, so a sign flip of a near-zero forward
        wf, wd = ops.pack_conv64_weights(w)                               # output cannot leak into the dgrad comparison
        Wg = W + 3 - int(rng.integers(0, 4))                       # weight gradient: an arbitrary W around it
        xg = torch.randn((N, D, H, Wg, 64), device="cuda", generator=g)
        dzg = torch.randn((N, D, H, Wg, 64), device="cuda", generator=g)

        def run(algo=ops.ALGO_AUTO):
            y = ops.conv3d_fwd(x, w, b, ops.ACT_LEAKY, 0.2, res, wpack=wf, algo=algo)
            pad = torch.zeros((N, D + 2, H + 2, W + 2, 64), device="cuda")
            out = torch.zeros_like(x)
            ops.conv3d_dgrad_fused(dz, wd, pad, out, skip=res, y_prev=yfix, act=ops.ACT_LEAKY, algo=algo)
            ops.fold_halo_border([pad], out, res, yfix, ops.ACT_LEAKY)
            dw, _ = ops.conv3d_wgrad(xg, dzg, 3, 64, 64, algo=algo)
            return y, out, dw

        got = run()
        got_h2 = run(ops.ALGO_WINO_H2)
        with fdn._lib.test_build() as lib:
            lib.fdn_debug_set_conv64_mt(5); lib.fdn_debug_set_wgrad64_direct(1)
            try:
                ref = run()
            finally:
                lib.fdn_debug_set_conv64_mt(0); lib.fdn_debug_set_wgrad64_direct(0)
        # max error over max |reference|, round 4's bound for both: F(4,3) x F(4,3) on the points 0, +-3/4, +-3/2, inf is as accurate as
        # F(2,3) x F(4,3) on the classic points (on 0, +-1, +-2, inf it measured up to 1.1e-5 here)
        for name, a, a2, r in zip(("fwd", "dgrad", "wgrad"), got, got_h2, ref):
            scale = max(r.abs().max().item(), 1e-30)
            err, err2 = (a - r).abs().max().item() / scale, (a2 - r).abs().max().item() / scale
            assert err2 <= 1e-5, (name, "F(2,3) along H", (N, D, H, W, Wg), err2)
            assert err <= 1e-5, (name, "auto", (N, D, H, W, Wg), err)


@pytest.mark.parametrize("shape", [(1, 5, 8, 12), (2, 24, 24, 24), (1, 7, 12, 20), (3, 4, 4, 4), (1, 19, 20, 12)])
@pytest.mark.parametrize("mb", [1, 2])
def test_conv64_wino2d_half_and_full_tiles(ops, fdn, shape, mb):
    """F(4,3) x F(4,3) with half-size tiles (MB = 1: one 16-cell M-block per wave, the planner's choice below 400 full tiles) and with full
    tiles (MB = 2), each forced through the test build: forward (all epilogues) and fused dgrad + border fold against the oracle."""
    rng = np.random.default_rng(23)
    N, D, H, W = shape
    x = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    res = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    y = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    dx = O.conv3d_dgrad(x.astype(np.float64), w.astype(np.float64), (N, D, H, W, 64))
    with fdn._lib.test_build() as lib:
        lib.fdn_debug_set_conv64_wino2d_mb(mb)
        try:
            for act, bias, r in [(O.ACT_RELU, b, None), (O.ACT_LEAKY, None, res), (O.ACT_NONE, None, None)]:
                ref = O.conv3d_fwd(x.astype(np.float64), w.astype(np.float64), None if bias is None else bias.astype(np.float64),
                                   act, 0.2, None if r is None else r.astype(np.float64))
                got = ops.conv3d_fwd(dev(x), dev(w), None if bias is None else dev(bias), act, 0.2, None if r is None else dev(r))
                close(got, ref, name="conv64 fwd MB=%d act=%d" % (mb, act))
            _, wd = ops.pack_conv64_weights(dev(w))
            pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
            out = torch.full((N, D, H, W, 64), float("nan"), device="cuda")
            ops.conv3d_dgrad_fused(dev(x), wd, pad, out, skip=dev(res), y_prev=dev(y), act=O.ACT_LEAKY)
            ops.fold_halo_border([pad], out, dev(res), dev(y), O.ACT_LEAKY)
            close(out, O.act_bwd_from_output(dx + res, y, O.ACT_LEAKY), name="fused dgrad+border MB=%d" % mb)
        finally:
            lib.fdn_debug_set_conv64_wino2d_mb(0)


@pytest.mark.parametrize("shape", [(1, 5, 8, 12), (2, 24, 24, 24), (1, 7, 12, 20), (3, 4, 4, 4)])
@pytest.mark.parametrize("mb", [0, 1, 2])
def test_fp32_sign_masks_replace_y_in_the_fused_dgrad(ops, fdn, shape, mb):
    """Training forward of a 64->64 layer writes the sign mask of its output beside it (planar [cout / 16][voxel] int16 words); the fused
    dgrad of its consumer reads the mask instead of y_prev.  The mask holds exactly (y > 0), the forward output is unchanged and the
    fused dgrad is bit-identical to the y_prev form -- with the planner's tiles (mb = 0) and with half-size / full tiles forced."""
    rng = np.random.default_rng(31)
    N, D, H, W = shape
    x = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    w = dev((rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32))
    res = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    dz = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    skip = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    assert ops.conv64_mask_ok(N, D, H, W) and not ops.conv64_mask_ok(N, D, H, W + 2) and not ops.conv64_mask_ok(N, D, H, W, ops.ALGO_DIRECT)
    wf, wd = ops.pack_conv64_weights(w)
    with (fdn._lib.test_build() if mb else contextlib.nullcontext()) as lib:
        if lib is not None:
            lib.fdn_debug_set_conv64_wino2d_mb(mb)
        try:
            for act, r in ((O.ACT_LEAKY, res), (O.ACT_RELU, None)):
                y0 = ops.conv3d_fwd(x, w, None, act, 0.2, r, wpack=wf)
                mask = ops.new_sign_mask(y0)
                mask.fill_(0x5a5a)
                y1 = ops.conv3d_fwd(x, w, None, act, 0.2, r, wpack=wf, mask=mask)
                assert torch.equal(y0, y1)
                bits = (y0 > 0).view(N * D * H * W, 4, 16).to(torch.int32)
                words = (bits << torch.arange(16, device="cuda", dtype=torch.int32)).sum(dim=2)          # (voxel, plane)
                assert torch.equal(words.t().contiguous(), mask.to(torch.int32) & 0xffff)
                pads, outs = [], []
                for use_mask in (False, True):
                    pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
                    out = torch.full((N, D, H, W, 64), float("nan"), device="cuda")
                    ops.conv3d_dgrad_fused(dz, wd, pad, out, skip=skip, y_prev=None if use_mask else y0, act=act, mask=mask if use_mask else None)
                    pads.append(torch.nan_to_num(pad, nan=-7.0)); outs.append(torch.nan_to_num(out, nan=-7.0))
                assert torch.equal(pads[0], pads[1]) and torch.equal(outs[0], outs[1])
        finally:
            if lib is not None:
                lib.fdn_debug_set_conv64_wino2d_mb(0)
    with pytest.raises(fdn.FdnError):                       # a grid off the F(4,3) x F(4,3) kernels refuses masks loudly
        xs = torch.zeros((1, 4, 6, 6, 64), device="cuda")
        ops.conv3d_fwd(xs, w, None, O.ACT_RELU, 0.2, None, wpack=wf, mask=ops.new_sign_mask(xs))


@pytest.mark.parametrize("shape", [(1, 5, 8, 12), (2, 24, 24, 24), (1, 6, 4, 4), (3, 3, 12, 20)])
@pytest.mark.parametrize("nsrc,use_mask,order", [(3, True, (0, 1, 2)), (3, False, (2, 0, 1)), (2, True, (1, 0)), (1, False, (0,))])
def test_multi_source_fused_dgrad_equals_the_chained_launches(ops, fdn, shape, nsrc, use_mask, order):
    """fdn_conv64_dgrad_fused_multi: dz_prev = fold(sum_s conv_T(dz_s, W_s)) in ONE launch (the three heads' 64->64 convs share their input).
    Against the chained single-source launches (skip = the running sum) to fp32 rounding, and against the float64 oracle; the packs are
    views of one buffer handed over in any order (the entry point orders them by address)."""
    rng = np.random.default_rng(41)
    N, D, H, W = shape
    ws = [(rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32) for _ in range(nsrc)]
    dzs = [rng.normal(size=(N, D, H, W, 64)).astype(np.float32) for _ in range(nsrc)]
    y = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    skip = rng.normal(size=(N, D, H, W, 64)).astype(np.float32)
    buf = torch.zeros((nsrc, 2, ops.CONV64_PACK_FLOATS), device="cuda")
    for s in range(nsrc):
        ops.pack_conv64_weights(dev(ws[s]), buf[order[s], 0], buf[order[s], 1])
    packs = [buf[order[s], 1] for s in range(nsrc)]
    ddz = [dev(z) for z in dzs]
    ydev, sdev = dev(y), dev(skip)
    mask = None
    if use_mask:                                             # the mask of y, as the forward would have written it
        bits = (ydev > 0).view(N * D * H * W, 4, 16).to(torch.int32)
        mask = (bits << torch.arange(16, device="cuda", dtype=torch.int32)).sum(dim=2).t().contiguous()
        mask = torch.where(mask >= 32768, mask - 65536, mask).to(torch.int16)
    # chained reference: sources 0 .. n-2 linear with skip = running sum, the last applies act'
    out_c = torch.zeros((N, D, H, W, 64), device="cuda")
    pads_c = []
    for s in range(nsrc):
        pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
        last = s == nsrc - 1
        ops.conv3d_dgrad_fused(ddz[s], packs[s], pad, out_c, skip=(sdev if s == 0 else out_c), y_prev=ydev if last else None,
                               act=O.ACT_LEAKY if last else O.ACT_NONE)
        pads_c.append(pad)
    ops.fold_halo_border(pads_c, out_c, sdev, ydev, O.ACT_LEAKY)
    pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
    out_m = torch.full((N, D, H, W, 64), float("nan"), device="cuda")
    ops.conv3d_dgrad_fused_multi(ddz, packs, pad, out_m, skip=sdev, y_prev=None if use_mask else ydev, act=O.ACT_LEAKY, mask=mask)
    ops.fold_halo_border([pad], out_m, sdev, ydev, O.ACT_LEAKY)
    assert torch.isfinite(out_m).all()
    scale = out_c.abs().max().item()
    assert (out_m - out_c).abs().max().item() <= 2e-5 * scale
    ref = sum(O.conv3d_dgrad(dzs[s].astype(np.float64), ws[s].astype(np.float64), (N, D, H, W, 64)) for s in range(nsrc))
    ref = O.act_bwd_from_output(ref + skip, y, O.ACT_LEAKY)
    close(out_m, ref, name="multi-source fused dgrad")
    if nsrc == 1:                                             # one source: the single-source kernel's own arithmetic, bit for bit
        assert torch.equal(out_m, out_c)
    with pytest.raises(fdn.FdnError):                        # a grid off the F(4,3) x F(4,3) kernels refuses loudly
        z = torch.zeros((1, 4, 6, 6, 64), device="cuda")
        ops.conv3d_dgrad_fused_multi([z, z], packs[:1] * 2, torch.zeros((1, 6, 8, 8, 64), device="cuda"), torch.zeros_like(z))


@pytest.mark.parametrize("shape", [(1, 5, 8, 12), (2, 24, 24, 24), (1, 3, 7, 4), (2, 9, 5, 20)])
def test_fp32_sign_mask_replaces_y_in_the_head_dgrad(ops, shape):
    """fdn_conv_cout1_dgrad_folded_mask: the 64->1 head dgrad reads the producer's planar sign mask (eight 8-B loads per 16 voxels and lane)
    instead of y's rows: dz_prev and the producer's bias gradient bit for bit (ragged tiles in d and h included; W % 4 == 0)."""
    rng = np.random.default_rng(77)
    N, D, H, W = shape
    y = dev(rng.normal(size=(N, D, H, W, 64)).astype(np.float32))
    w = dev((rng.normal(size=(3, 3, 3, 64, 1)) * 0.1).astype(np.float32))
    dpred = dev(rng.normal(size=(N, D, H, W, 3)).astype(np.float32))
    bits = (y > 0).view(N * D * H * W, 4, 16).to(torch.int32)
    mask = (bits << torch.arange(16, device="cuda", dtype=torch.int32)).sum(dim=2).t().contiguous()
    mask = torch.where(mask >= 32768, mask - 65536, mask).to(torch.int16)
    for coff in (0, 2):
        db0, db1 = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
        a = ops.conv_cout1_dgrad_folded(dpred, w, (N, D, H, W), y, O.ACT_RELU, lddz=3, dz_coff=coff, dbias_prev=db0)
        b = ops.conv_cout1_dgrad_folded(dpred, w, (N, D, H, W), None, O.ACT_RELU, lddz=3, dz_coff=coff, dbias_prev=db1, mask=mask)
        assert torch.equal(a, b) and torch.equal(db0, db1)
        assert 0.2 < (a == 0).float().mean().item() < 0.8          # (the ReLU mask really bit)
