"""Parity of the bf16 activation path (BASELINE.json configs[3]) against the CPU oracle, through the C-ABI.

Model of the arithmetic: operands are bf16 values (inputs rounded with oracle.bf16_round), products are exact, the
accumulation is fp32, the stored result is rounded to bf16 once.  The oracle accumulates in float64, so a GPU result may
differ from round_bf16(oracle) only where fp32 accumulation noise crosses a rounding boundary: every element must be
within one bf16 ulp (2^-8 relative) + fp32 noise, and all but a small fraction must be bit-identical after rounding."""
import numpy as np
import pytest
import torch

from oracle import flownet_oracle as O

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


SHAPES = [(2, 8, 8, 8), (1, 5, 7, 9), (1, 10, 12, 16), (3, 4, 4, 2), (1, 1, 1, 1), (1, 16, 16, 16), (1, 3, 20, 11)]


def rb(a):
    return O.bf16_round(a.astype(np.float32)).astype(np.float64)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mt", [0, 4, 8])
def test_conv64_fwd_bf16(bops, fdn, shape, mt):
    rng = np.random.default_rng(11)
    N, D, H, W = shape
    x = rb(rng.normal(size=(N, D, H, W, 64)))
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    res = rb(rng.normal(size=(N, D, H, W, 64)))
    wf, _ = bops.pack_conv64_weights(dev(w))
    lib = fdn._lib.load()
    lib.fdn_debug_set_conv64_bf16_mt(mt)
    try:
        for act, bias, r in [(O.ACT_RELU, b, None), (O.ACT_LEAKY, None, res), (O.ACT_NONE, None, None)]:
            ref = O.conv3d_fwd(x, rb(w), None if bias is None else bias.astype(np.float64), act, 0.2, r)
            got = bops.conv64_fwd(devb(x), wf, None if bias is None else dev(bias), act, 0.2,
                                  None if r is None else devb(r))
            close_bf16(got, ref, name="conv64 bf16 fwd act=%d" % act)
    finally:
        lib.fdn_debug_set_conv64_bf16_mt(0)


@pytest.mark.parametrize("shape", SHAPES + [(1, 2, 3, 1)])
@pytest.mark.parametrize("mt", [0, 4, 8])
def test_conv64_dgrad_fused_bf16(bops, fdn, shape, mt):
    rng = np.random.default_rng(12)
    N, D, H, W = shape
    dz = rb(rng.normal(size=(N, D, H, W, 64)))
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    y = rb(rng.normal(size=(N, D, H, W, 64)))
    skip = rb(rng.normal(size=(N, D, H, W, 64)))
    dx = O.conv3d_dgrad(dz, rb(w), (N, D, H, W, 64))
    _, wd = bops.pack_conv64_weights(dev(w))
    lib = fdn._lib.load()
    lib.fdn_debug_set_conv64_bf16_mt(mt)
    try:
        pad = torch.full((N, D + 2, H + 2, W + 2, 64), float("nan"), device="cuda")
        out = torch.full((N, D, H, W, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
        bops.conv64_dgrad_fused(devb(dz), wd, pad, out, skip=devb(skip), y_prev=devb(y), act=O.ACT_LEAKY)
        bops.fold_halo_border([pad], out, devb(skip), devb(y), O.ACT_LEAKY)
        close_bf16(out, O.act_bwd_from_output(dx + skip, y, O.ACT_LEAKY), name="bf16 fused dgrad+border")
    finally:
        lib.fdn_debug_set_conv64_bf16_mt(0)
