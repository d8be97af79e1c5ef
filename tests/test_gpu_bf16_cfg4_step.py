"""One FULL cfg4 train step (BASELINE.json configs[3]: patch 32, res x4, 8 + 4 ResBlocks, bf16 activations; batch 1 here)
checked layer by layer against the bf16-emulating float64 reference on SAMPLED voxels (last bullet).

The numpy oracle needs ~15 minutes for one cfg4 patch, so the whole-network comparison of tests/test_gpu_bf16_train.py is
not available at this size.  Instead the step's own tensors are recorded -- for sampled ResBlocks of both resolutions the
layer inputs, outputs and gradients exactly as the product path produced them inside ONE forward + backward -- and every
kernel launch they came from is re-derived in float64 from its bf16 operands at sampled voxels (corners, edges, faces,
interior): forward conv + residual + LeakyReLU, fused dgrad + skip + act', weight gradient.  Same rounding points as the HIP
path (bf16 activations / gradients / MFMA weight copies, fp32 accumulation): a result may differ from the rounded reference
by one bf16 ulp.  Semantics: src/Network/SR4DFlowNet.py:111-120 (resnet_block) under tape.gradient (TrainerController.py:223)."""
import importlib

import numpy as np
import pytest
import torch

from oracle import flownet_oracle as O
from test_gpu_fullsize import gather_rows, ref_dgrad, ref_forward, ref_wgrad_rows, sample_voxels

pytestmark = pytest.mark.gpu

ULP = 2.0 ** -8
PICKS = [(0, 0, 0, 3), (1, 1, 1, 17), (2, 2, 2, 63), (0, 2, 1, 31), (2, 0, 1, 40), (1, 0, 2, 5)]


def within_ulp(got, ref, name):
    tol = ULP * np.abs(ref) + 3e-5 * np.abs(ref).max()
    bad = np.abs(got - ref) > tol
    assert not bad.any(), "%s: %d of %d sampled elements off by more than one bf16 ulp (worst %.3e of scale %.3e)" % (
        name, int(bad.sum()), bad.size, float(np.abs(got - ref).max()), float(np.abs(ref).max()))


def test_full_cfg4_train_step_layers_match_bf16_reference(fdn):
    trainer = importlib.import_module("4dflownet_amd.trainer")
    bops = importlib.import_module("4dflownet_amd.ops_bf16")
    P, R, LB, HB = 32, 4, 8, 4
    tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB,
                                   seed=0, dtype="bfloat16")
    m = tc.model
    batch = O.synthetic_batch(1, P, R, seed=77)
    rec_w, rec_f = {}, {}
    orig_wgrad, orig_conv = m._wgrad, m._conv

    def rec_wgrad(x, dz, L, **k):
        if (L.k, L.cin, L.cout) == (3, 64, 64):
            rec_w[L.name] = (x, dz)
        return orig_wgrad(x, dz, L, **k)

    def rec_conv(x, L, act, residual=None, **k):
        y = orig_conv(x, L, act, residual=residual, **k)
        if (L.k, L.cin, L.cout) == (3, 64, 64):
            rec_f[L.name] = (x, residual, act, y)
        return y
    m._wgrad, m._conv = rec_wgrad, rec_conv
    inputs, hires, venc, mask = tc._unpack(batch)
    pred = m.forward(inputs, training=True)
    out, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask)
    g_before = m.flat_g.clone()
    m.backward(dpred)
    m._wgrad, m._conv = orig_wgrad, orig_conv
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all() and torch.isfinite(m.flat_g).all() and not torch.equal(m.flat_g, g_before)

    rng = np.random.default_rng(12)
    # ResBlock k = layers (6 + 2k, 7 + 2k); low-res blocks 0 and 7 (32^3), hi-res blocks 8 and 11 (128^3)
    for blk in (0, LB - 1, LB, LB + HB - 1):
        La, Lb = m.layers[6 + 2 * blk], m.layers[7 + 2 * blk]
        xa, res_a, act_a, h = rec_f[La.name]
        xb, res_b, act_b, y = rec_f[Lb.name]
        assert res_a is None and res_b is xa and xb is h and act_a == act_b == bops.ACT_LEAKY
        N, D, H, W = xa.shape[:4]
        assert (D, H, W) == ((P,) * 3 if blk < LB else (P * R,) * 3)
        dims = (N, D, H, W)
        pts = sample_voxels(N, D, H, W, 60, rng)
        wa = La.w.to(torch.bfloat16).double().cpu().numpy()          # the MFMA streams hold the bf16-rounded kernels
        wb = Lb.w.to(torch.bfloat16).double().cpu().numpy()
        leaky = lambda z: np.where(z > 0, z, 0.2 * z)
        # forward: h = leaky(conv(x, Wa)), y = leaky(conv(h, Wb) + x)      (SR4DFlowNet.py:111-120)
        within_ulp(gather_rows(h, pts), leaky(ref_forward(xa, wa, pts, dims)), "%s forward" % La.name)
        within_ulp(gather_rows(y, pts), leaky(ref_forward(h, wb, pts, dims) + gather_rows(xa, pts)), "%s forward + residual" % Lb.name)
        # backward: dz_h = dgrad(dz; Wb) * leaky'(h)      (the dz the first conv's weight gradient consumed)
        (hb, dz), (xa2, dz_h) = rec_w[Lb.name], rec_w[La.name]
        assert hb is h and xa2 is xa
        refd = ref_dgrad(dz, wb, pts, dims) * np.where(gather_rows(h, pts) > 0, 1.0, 0.2)
        within_ulp(gather_rows(dz_h, pts), refd, "%s fused dgrad" % Lb.name)
        # the block's input gradient = (dgrad(dz_h; Wa) + dz) * act'(x): it is the dz of the block below, when that is a
        # ResBlock of the same resolution (leaky producer)
        if blk not in (0, LB):
            _, dz_prev = rec_w[m.layers[7 + 2 * (blk - 1)].name]
            refp = (ref_dgrad(dz_h, wa, pts, dims) + gather_rows(dz, pts)) * np.where(gather_rows(xa, pts) > 0, 1.0, 0.2)
            within_ulp(gather_rows(dz_prev, pts), refp, "%s fused dgrad + skip" % La.name)
        # weight gradients (fp32 results): sampled (tap, cin) rows against float64 sums of the bf16 operands
        for L, (xx, dd) in ((La, (xa, dz_h)), (Lb, (h, dz))):
            refw = ref_wgrad_rows(xx, dd, PICKS)
            cond = ref_wgrad_rows(xx.abs(), dd.abs(), PICKS)           # sum |x||dz|: the bound any fp32 summation order obeys
            gotw = np.asarray([L.gw[a, b, c, ci].double().cpu().numpy() for (a, b, c, ci) in PICKS])
            assert (np.abs(gotw - refw) <= 1e-5 * cond + 1e-30).all(), "%s wgrad: %.3e of bound" % (
                L.name, float((np.abs(gotw - refw) / np.maximum(cond, 1e-300)).max()))
    # and the step as a whole updates the parameters
    w0 = m.flat_w.clone()
    loss = tc.train_step(batch)
    assert torch.isfinite(loss).all() and 0 < float((m.flat_w - w0).abs().max()) <= 1.05e-4
