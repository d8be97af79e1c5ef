"""Oracle self-consistency: the numpy restatement vs torch-CPU (float64) as an
independent second opinion on the math (TensorFlow is not
available, so this is NOT a check against the reference -- parity unpinned)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import flownet_oracle as O
from oracle.torch_cpu import t_forward, t_loss


def _rand_params(LB, HB, seed=0):
    params = O.init_params(seed, LB, HB, np.float64)
    rng = np.random.default_rng(seed + 1)
    for p in params:      # non-zero biases so bias grads are exercised
        if p["b"] is not None:
            p["b"] = rng.normal(0, 0.05, p["b"].shape)
        p["w"] = p["w"] * 3.0   # keep activations alive through the stack
    return params


@pytest.mark.parametrize("P,R,LB,HB", [(6, 2, 1, 1), (5, 1, 2, 0), (4, 3, 0, 1)])
def test_loss_and_grads_match_torch_autograd(P, R, LB, HB):
    B = 2
    params = _rand_params(LB, HB)
    batch = O.synthetic_batch(B, P, R, seed=7, dtype=np.float64)
    out = O.loss_and_grads(params, batch, R, LB, HB)

    tp = []
    for p in params:
        w = torch.tensor(p["w"], dtype=torch.float64, requires_grad=True)
        b = None if p["b"] is None else torch.tensor(p["b"], dtype=torch.float64, requires_grad=True)
        tp.append((w, b))
    tb = [torch.tensor(a, dtype=torch.float64) for a in batch]
    pred = t_forward(tp, tb[:6], R, LB, HB)
    hires = torch.cat(tb[6:9], -1)
    mse = t_loss(pred, hires, tb[10])
    l2 = sum(O.L2_LAMBDA * (w ** 2).sum() for w, _ in tp)
    loss = mse + l2                      # shape (B,), like TrainerController.py:249
    loss.sum().backward()                # tape.gradient of a vector target sums it

    np.testing.assert_allclose(out["pred"], pred.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(out["loss"], loss.detach().numpy(), rtol=1e-9)
    for g, (w, b), p in zip(out["grads"], tp, params):
        np.testing.assert_allclose(g["w"], w.grad.numpy(), rtol=1e-7, atol=1e-11, err_msg=p["name"])
        if b is not None:
            np.testing.assert_allclose(g["b"], b.grad.numpy(), rtol=1e-7, atol=1e-11, err_msg=p["name"])


def test_upsample_matches_two_pass_bilinear_composition():
    """SR4DFlowNet.py:77-89 composes two 2-D align_corners resizes; the oracle uses separable trilinear."""
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 5, 4, 3, 2))
    for R in (2, 3, 4):
        y = O.upsample_trilinear_fwd(x, R)
        xt = torch.tensor(x).permute(0, 4, 1, 2, 3)
        yt = F.interpolate(xt, scale_factor=R, mode="trilinear", align_corners=True).permute(0, 2, 3, 4, 1)
        np.testing.assert_allclose(y, yt.numpy(), rtol=1e-12, atol=1e-13)
        # adjoint identity <Ux, r> == <x, U^T r>
        r = rng.normal(size=y.shape)
        np.testing.assert_allclose((y * r).sum(), (x * O.upsample_trilinear_bwd(r, x.shape[1:4], R)).sum(), rtol=1e-10)


def test_dgrad_wgrad_are_adjoints():
    rng = np.random.default_rng(5)
    x = rng.normal(size=(2, 4, 5, 3, 6))
    w = rng.normal(size=(3, 3, 3, 6, 7))
    dz = rng.normal(size=(2, 4, 5, 3, 7))
    y = O.conv3d_linear(x, w)
    np.testing.assert_allclose((y * dz).sum(), (x * O.conv3d_dgrad(dz, w, x.shape)).sum(), rtol=1e-10)
    np.testing.assert_allclose((y * dz).sum(), (w * O.conv3d_wgrad(x, dz, 3)).sum(), rtol=1e-10)


def test_relative_error_rounding_and_mask():
    pred = np.zeros((1, 1, 1, 4, 3)); tgt = np.zeros((1, 1, 1, 4, 3))
    tgt[0, 0, 0, :, 0] = [1.0, 2.0, 0.0, 1.0]
    pred[0, 0, 0, :, 0] = [1.00005, 2.5, 0.3, 5.0]     # rel 0.00005 -> rounds to 0.0 (half-to-even of 0.5), 0.25, diff 0.3, clip 1
    mask = np.array([[[[1.0, 1.0, 1.0, 0.0]]]])
    r = O.relative_error(pred, tgt, mask)
    exp = (np.round(0.00005 / (1 + 1e-5) * 1e4) / 1e4 + np.round(0.5 / (2 + 1e-5) * 1e4) / 1e4 + 0.3) / 4 * 100
    np.testing.assert_allclose(r, [exp], rtol=1e-12)


def test_adam_matches_keras_formula_against_torch_with_scaled_eps():
    """Keras puts epsilon outside the bias correction; torch.optim.Adam matches when its eps is
    scaled by 1/sqrt(1-b2^t) per step."""
    rng = np.random.default_rng(9)
    w = rng.normal(size=50); w0 = w.copy()
    m = np.zeros(50); v = np.zeros(50)
    wt = torch.tensor(w0.copy(), requires_grad=True)
    for t in range(1, 5):
        g = rng.normal(size=50)
        O.adam_step_tf(w, g, m, v, t, 1e-3)
    # closed form check of step 1: w1 = w0 - lr*sqrt(1-b2)/(1-b1) * (1-b1) g / (sqrt((1-b2) g^2) + eps)
    rng = np.random.default_rng(9); rng.normal(size=50); g1 = rng.normal(size=50)
    w1 = w0 - 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * g1) / (np.sqrt(0.001 * g1 * g1) + 1e-7)
    w = w0.copy(); m[:] = 0; v[:] = 0
    O.adam_step_tf(w, g1, m, v, 1, 1e-3)
    np.testing.assert_allclose(w, w1, rtol=1e-12)


def test_param_count_matches_survey():
    assert O.count_params(O.init_params(0, 8, 4)) == 3342083
    assert O.count_params(O.init_params(0, 2, 1)) == 1351427       # cfg1
    names = [s[0] for s in O.layer_specs(8, 4)]
    assert names[0] == "conv3d" and names[-1] == "conv3d_35" and len(names) == 36
