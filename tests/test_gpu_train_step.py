"""End-to-end parity on the GPU: forward, loss, full backward and Adam updates of the HIP path vs the CPU oracle
(float64) from identical weights and inputs.  Tolerance 1e-3 relative (north_star); observed ~1e-5."""
import numpy as np
import pytest
import torch

from oracle import flownet_oracle as O
from _kink import kink_sides

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


def make(fdn, P, R, LB, HB, seed=0, wscale=3.0):
    trainer_mod = __import__("importlib").import_module("4dflownet_amd.trainer")
    tc = trainer_mod.TrainerController(P, R, initial_learning_rate=1e-3, quicksave_enable=False, low_resblock=LB,
                                       hi_resblock=HB, seed=seed)
    params = O.init_params(seed, LB, HB, np.float64)
    rng = np.random.default_rng(seed + 1)
    for p in params:
        p["w"] = p["w"] * wscale
        if p["b"] is not None:
            p["b"] = rng.normal(0, 0.05, p["b"].shape)
    arrays = []
    for p in params:
        arrays.append(p["w"].astype(np.float32))
        if p["b"] is not None:
            arrays.append(p["b"].astype(np.float32))
    tc.model.set_weights(arrays)
    # oracle starts from the fp32-rounded values the GPU holds
    it = iter(arrays)
    for p in params:
        p["w"] = next(it).astype(np.float64)
        if p["b"] is not None:
            p["b"] = next(it).astype(np.float64)
    return tc, params


def test_glorot_init_matches_oracle_seeded_init(fdn):
    net = __import__("importlib").import_module("4dflownet_amd.network")
    m = net.SR4DFlowNet(2).build_network(*[net.Input((8, 8, 8, 1))] * 6, low_resblock=1, hi_resblock=1, seed=5)
    ref = O.flatten(O.init_params(5, 1, 1, np.float32))
    np.testing.assert_array_equal(m.flat_w.cpu().numpy(), ref)
    assert m.n_params == O.count_params(O.init_params(5, 1, 1))


def test_auto_algo_warns_once_when_a_grid_falls_off_the_winograd_kernels(fdn):
    """FDN_ALGO_AUTO picks the 64->64 kernel by the extents and the model says so once per grid that lands on slower kernels -- and only
    then: P = 18 (a legal reference patch size, W % 4 != 0) runs its aligned 16 x 16 box on F(4,3) x F(4,3) and two strips on the direct
    kernel (round 6): silent, like P = 16; a 14 x 6 x 6 grid is too small for the split (all direct): warned."""
    import warnings
    net = __import__("importlib").import_module("4dflownet_amd.network")
    g = torch.Generator(device="cuda").manual_seed(0)
    for shp, expect in (((6, 18, 18, 18), False), ((6, 16, 16, 16), False), ((70, 14, 6, 6), True)):
        m = net.FlowNetModel(1, low_resblock=1, hi_resblock=0, seed=0)
        x = [torch.rand(shp + (1,), device="cuda", generator=g) for _ in range(6)]
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            m.forward(x); m.forward(x)
        hits = [i for i in w if issubclass(i.category, RuntimeWarning) and "W % 4 == 0" in str(i.message)]
        assert len(hits) == (1 if expect else 0), [str(i.message) for i in w]
        if expect:
            assert "direct kernels" in str(hits[0].message) and "14x6x6" in str(hits[0].message)
        # the library agrees with the python rule: which pack streams does this grid read?
        need = fdn.ops.conv64_pack_streams(*shp, fdn.ops.ALGO_AUTO, fdn.ops.ROLE_FWD)
        assert bool(need & fdn.ops.PACK_STREAM_WINO_H4) == (not expect), (shp, need)


@pytest.mark.parametrize("P,R,LB,HB,B", [(6, 2, 1, 1, 2), (8, 1, 2, 1, 2), (4, 3, 0, 1, 1), (6, 2, 2, 0, 3)])
def test_train_step_matches_oracle(fdn, P, R, LB, HB, B):
    for seed in range(3):
        tc, params = make(fdn, P, R, LB, HB, seed=seed)
        batch = O.synthetic_batch(B, P, R, seed=21 + seed)
        b64 = tuple(a.astype(np.float64) for a in batch)
        state = {}
        ad_state = {}
        for step in range(2):
            # every step starts from identical parameters: the oracle adopts the GPU's fp32 weights (the +-lr
            # ambiguity of Adam on noise-level gradients, see below, must not leak into the next step's check)
            it = iter(tc.model.get_weights())
            for p in params:
                p["w"] = next(it).astype(np.float64)
                if p["b"] is not None:
                    p["b"] = next(it).astype(np.float64)
            # GPU: forward/backward pieces individually first so they can be compared
            inputs, hires, venc, mask = tc._unpack(batch)
            pred = tc.model.forward(inputs, training=True)
            _, rc = O.network_forward(params, b64[:6], R, LB, HB, f32_coeffs=True)
            sides, flips, worst = kink_sides(tc.model._cache, rc)
            n_units = sum(int(np.size(v)) for v in sides.values() if isinstance(v, np.ndarray)) + \
                sum(h.size + o.size for h, o in sides["blocks"]) + sum(g.size for g in sides["heads"])
            assert flips <= max(2, 1e-5 * n_units) and worst <= 2e-5, (flips, n_units, worst)
            # the oracle differentiates in the linear region the GPU forward landed in: EVERY instance is held to the tight bound
            ref = O.loss_and_grads(params, b64, R, LB, HB, f32_coeffs=True, sides=sides)
            tol_g = 1e-4
            out, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask)
            g = tc.model.backward(dpred).cpu().numpy().astype(np.float64)
            assert rel_err(pred.cpu().numpy(), ref["pred"]) < 1e-4
            assert rel_err(out[:, 0].cpu().numpy(), ref["mse"]) < 1e-4
            isk = tc.model.is_kernel.cpu().numpy().astype(np.float64)
            g_total = g + B * 2 * O.L2_LAMBDA * tc.model.flat_w.cpu().numpy().astype(np.float64) * isk
            gref = O.flatten(ref["grads"])
            # per-layer comparison so a broken thin layer cannot hide behind the big ones
            for L in tc.model.layers:
                sl = slice(L.w_off, L.w_off + L.w.numel())
                assert rel_err(g_total[sl], gref[sl]) < tol_g, (L.name, "kernel grad", rel_err(g_total[sl], gref[sl]), flips)
                if L.b is not None:
                    sb = slice(L.b_off, L.b_off + L.cout)
                    assert rel_err(g_total[sb], gref[sb]) < tol_g, (L.name, "bias grad", flips)
            # now the real step on both sides.  Adam's update is ~ lr*sign(g) on the first steps, so an element whose
            # gradient is rounding noise may move by +-lr in either direction: bound those by 2.5*lr*steps.
            w_before = tc.model.flat_w.cpu().numpy().astype(np.float64)
            loss = tc.train_step(batch)
            O.train_step(params, state, b64, 1e-3, R, LB, HB, f32_coeffs=True)
            assert rel_err(loss.cpu().numpy(), ref["loss"]) < 1e-4
            w_gpu = tc.model.flat_w.cpu().numpy().astype(np.float64)
            w_ref = O.flatten(params)
            assert np.abs(w_gpu - w_ref).max() <= 2.5e-3
            assert np.abs(w_gpu - w_before).max() <= 1.05e-3          # |Adam update| <= lr while m/sqrt(v) <= 1
            # The optimizer arithmetic itself, decoupled from gradient noise: Keras-Adam in float64 driven by the GPU's OWN gradient
            # (g_total above; train_step recomputes the identical, deterministic gradient) must land on the GPU's weights.  (Comparing
            # against the oracle's gradient instead amplifies its ~1e-6 rounding noise by lr/eps' = 316 wherever |g| ~ eps' = 3e-6.)
            ad_m = ad_state.setdefault("m", np.zeros_like(w_before)); ad_v = ad_state.setdefault("v", np.zeros_like(w_before))
            w_exp = w_before.copy()
            O.adam_step_tf(w_exp, g_total, ad_m, ad_v, step + 1, 1e-3)
            assert np.abs(w_gpu - w_exp).max() <= 2e-6, "adam update vs float64 Keras-Adam on the same gradient"
        assert tc.loss_metrics["train_loss"].result() > 0
        assert abs(tc.loss_metrics["l2_reg_loss"].result() - O.l2_regularizer(params)) / O.l2_regularizer(params) < 1e-2


def test_test_step_and_predict(fdn):
    tc, params = make(fdn, 6, 2, 1, 1)
    batch = O.synthetic_batch(2, 6, 2, seed=22)
    pred = tc.test_step(batch)
    ref, _ = O.network_forward(params, tuple(a.astype(np.float64) for a in batch[:6]), 2, 1, 1, f32_coeffs=True)
    assert rel_err(pred.cpu().numpy(), ref) < 1e-4
    out = tc.model.predict(list(batch[:6]), batch_size=1)
    assert out.shape == ref.shape and rel_err(out, ref) < 1e-4
    w_before = tc.model.flat_w.clone()
    tc.test_step(batch)
    assert torch.equal(w_before, tc.model.flat_w)          # test_step never updates


def test_wgrad_side_stream_gives_identical_gradients(fdn):
    """overlap_wgrad=True runs the weight-gradient launches on a second HIP stream; same kernels, same order within each
    layer, so the gradient buffer must be bit-identical to the single-stream schedule."""
    tc, _ = make(fdn, 8, 2, 2, 1, seed=3)
    batch = O.synthetic_batch(2, 8, 2, seed=9)
    grads = []
    for ov in (False, True, True):
        tc.model.overlap_wgrad = ov
        inputs, hires, venc, mask = tc._unpack(batch)
        pred = tc.model.forward(inputs, training=True)
        out, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask)
        grads.append(tc.model.backward(dpred).clone())
        torch.cuda.synchronize()
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])


@pytest.mark.parametrize("P,R,LB,HB,B", [(8, 2, 2, 1, 2), (12, 2, 1, 2, 1), (10, 2, 1, 1, 2)])
def test_fp32_sign_masks_give_identical_gradients(fdn, P, R, LB, HB, B):
    """fp32 training on grids the F(4,3) x F(4,3) kernels serve: the forward of a 64->64 layer writes the sign mask of its output and the
    fused dgrad reads it instead of y (network._conv_m / _dgrad_fold).  Same gradient buffer bit for bit as reading y; on a grid off those
    kernels (P = 10 low-res: W % 4 != 0) no mask is produced and nothing changes."""
    batch = O.synthetic_batch(B, P, R, seed=19)
    grads = []
    for use_masks in (True, False):
        tc, _ = make(fdn, P, R, LB, HB, seed=5)
        tc.model.sign_masks = use_masks
        inputs, hires, venc, mask = tc._unpack(batch)
        pred = tc.model.forward(inputs, training=True)
        if use_masks:
            hm = tc.model._cache["hmasks"]
            assert tc.model._cache["rb"].mask is not None                     # the high-res grid (P * R) is a multiple of 4 in all three
            assert all(m is not None for m in hm) if P % 4 == 0 else (any(m is None for m in hm) and any(m is not None for m in hm))
        out, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask)
        grads.append(tc.model.backward(dpred).clone())
        torch.cuda.synchronize()
    assert torch.isfinite(grads[0]).all() and torch.equal(grads[0], grads[1])


def test_batched_wgrad_gives_the_same_gradients(fdn):
    """batch_wgrad (default on): the 64->64 weight gradients of a gradient bucket's small-grid layers go out as ONE batched launch at
    the end of the bucket instead of one launch per layer.  Same products, a different split of the voxel sum: every layer's gradient
    equals the per-layer schedule's to fp32 rounding, everything else (biases, thin layers, hi-res layers) bit for bit -- with and without
    the side stream, and the bucket callbacks still arrive in order with every gradient of the bucket enqueued."""
    tc, _ = make(fdn, 8, 2, 2, 1, seed=3)
    m = tc.model
    batch = O.synthetic_batch(2, 8, 2, seed=9)
    grads = {}
    for key, (bw, ov) in (("ref", (False, False)), ("batched", (True, False)), ("batched+side", (True, True))):
        m.batch_wgrad, m.overlap_wgrad = bw, ov
        inputs, hires, venc, mask = tc._unpack(batch)
        pred = m.forward(inputs, training=True)
        out, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask)
        seen = []
        g = m.backward(dpred, grad_ready=lambda lo, hi: seen.append((lo, hi)))
        grads[key] = g.clone()
        torch.cuda.synchronize()
        assert seen == list(m.grad_buckets) and not m._wg_pending
    m.batch_wgrad, m.overlap_wgrad = True, False
    ref = grads["ref"]
    assert torch.equal(grads["batched"], grads["batched+side"])
    n64 = 0
    for L in m.layers:
        sl = slice(L.w_off, L.w_off + L.w.numel())
        a, b = grads["batched"][sl], ref[sl]
        if (L.k, L.cin, L.cout) == (3, 64, 64):
            n64 += 1
            assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item(), L.name
        else:
            assert torch.equal(a, b), L.name
        if L.b is not None:
            sb = slice(L.b_off, L.b_off + L.cout)
            assert torch.equal(grads["batched"][sb], ref[sb]), L.name
    assert n64 >= 9


def test_full_size_cfg2_patch_matches_oracle(fdn):
    """BASELINE cfg2 network (patch 24, res x2, 8 LR + 4 HR ResBlocks, Glorot init as in bench.py) on ONE synthetic patch:
    prediction, loss and every layer's gradient against the float32 CPU oracle (one ~1 TFLOP CPU train step, 20-60 s).
    north_star tolerance: 1e-3 relative fp32."""
    P, R, LB, HB = 24, 2, 8, 4
    trainer_mod = __import__("importlib").import_module("4dflownet_amd.trainer")
    tc = trainer_mod.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB,
                                       hi_resblock=HB, seed=0)
    params = O.init_params(0, LB, HB, np.float32)
    batch = O.synthetic_batch(1, P, R, seed=1234)
    ref = O.loss_and_grads(params, batch, R, LB, HB, f32_coeffs=True)
    inputs, hires, venc, mask = tc._unpack(batch)
    pred = tc.model.forward(inputs, training=True)
    out, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask)
    g = tc.model.backward(dpred).cpu().numpy().astype(np.float64)
    assert rel_err(pred.cpu().numpy(), ref["pred"]) < 1e-3
    assert rel_err(out[:, 0].cpu().numpy(), ref["mse"]) < 1e-3
    # metric in percent: mean over ~13 k fluid voxels of values rounded to 1e-4 steps; an fp32-vs-float32-oracle difference can move a
    # voxel across a rounding step (1e-2 percentage points / 13 k each) -- hold the mean to 2e-3 percentage points
    assert abs(float(out[0, 1]) - float(ref["rel_err"][0])) < 2e-3
    isk = tc.model.is_kernel.cpu().numpy().astype(np.float64)
    g_total = g + 2 * O.L2_LAMBDA * tc.model.flat_w.cpu().numpy().astype(np.float64) * isk
    gref = O.flatten(ref["grads"]).astype(np.float64)
    worst = 0.0
    for L in tc.model.layers:
        sl = slice(L.w_off, L.w_off + L.w.numel())
        worst = max(worst, rel_err(g_total[sl], gref[sl]))
    assert worst < 1e-3, worst
    # cosine of the whole 3.3 M-element gradient
    cos = float(g_total @ gref / (np.linalg.norm(g_total) * np.linalg.norm(gref)))
    assert cos > 1 - 1e-6


def test_linearity_of_conv_at_full_size(fdn):
    """Size-independent property at the BASELINE shape (8,24^3,64): conv(ax+by) == a conv(x) + b conv(y)."""
    ops = fdn.ops
    torch.manual_seed(0)
    x = torch.randn((8, 24, 24, 24, 64), device="cuda"); y = torch.randn_like(x)
    w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.03
    lhs = ops.conv3d_fwd(2.0 * x - 0.5 * y, w)
    rhs = 2.0 * ops.conv3d_fwd(x, w) - 0.5 * ops.conv3d_fwd(y, w)
    assert (lhs - rhs).abs().max().item() < 1e-4 * rhs.abs().max().item()
    # adjoint identities <conv(x), g> == <x, fold(dgrad(g))> == <w, wgrad(x, g)> at the same size
    g = torch.randn_like(x)
    yx = ops.conv3d_fwd(x, w)
    lhs = (yx.double() * g.double()).sum().item()
    dx = ops.fold_halo([ops.conv3d_dgrad(g, w)])
    dw, _ = ops.conv3d_wgrad(x, g, 3, 64, 64)
    assert abs((x.double() * dx.double()).sum().item() - lhs) < 1e-4 * abs(lhs) + 1e-2
    assert abs((w.double() * dw.double()).sum().item() - lhs) < 1e-4 * abs(lhs) + 1e-2


def test_winograd_and_direct_kernels_train_alike(fdn):
    """The product path (2-D Winograd forward / dgrad, depth-transformed Winograd wgrad), the W-only Winograd kernels (FDN_ALGO_WINO_W)
    and the round-1 direct MFMA kernels (forced in the test build) run the same 6 training steps from the same weights and batch.
    First-step gradient: the 1-D path keeps round 3's bound (1e-5 of the scale); the product path is held to the same bound whenever
    its forward put every activation unit on the same side of its kink as the direct forward did -- only a run with counted kink flips
    (each moves single gradient elements by ~1e-4 of the scale) gets the wider one.  Per-step losses agree to fp32 rounding,
    and after 6 Adam steps the bulk of the weights has moved identically."""
    trainer_mod = __import__("importlib").import_module("4dflownet_amd.trainer")
    P, R, LB, HB, B = 8, 2, 2, 1, 2
    batch = O.synthetic_batch(B, P, R, seed=77)

    def run(algo):
        tc = trainer_mod.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=3,
                                           conv_algo=algo)
        inputs = tc._unpack(batch)[0]
        tc.model.forward(inputs, training=True)
        c = tc.model._cache
        masks = [(t > 0).cpu() for t in [c["a0"], c["a1"], c["p0"], c["p1"], c["c0"], c["c1"]] + [t for blk in c["blocks"] for t in blk[1:]] + list(c["heads"])]
        losses, g0 = [], None
        for step in range(6):
            losses.append(tc.train_step(batch).cpu().numpy().astype(np.float64))
            if step == 0:
                g0 = tc.model.flat_g.cpu().numpy().astype(np.float64)
        return np.asarray(losses), g0, tc.model.flat_w.cpu().numpy().astype(np.float64), masks

    l_w, g_w, w_w, m_w = run("auto")
    l_1, g_1, w_1, m_1 = run("winograd_w")
    with fdn._lib.test_build() as lib:
        lib.fdn_debug_set_conv64_mt(5)            # direct <1,2,cs2> forward / dgrad kernel
        lib.fdn_debug_set_wgrad64_direct(1)
        try:
            l_d, g_d, w_d, m_d = run("auto")
        finally:
            lib.fdn_debug_set_conv64_mt(0)
            lib.fdn_debug_set_wgrad64_direct(0)
    # the first step sees identical weights: the losses agree to fp32 rounding; after that the two runs are two summation orders under
    # Adam's +-lr*sign(g) updates of noise-level gradients, which separates the losses by a few 1e-6 per step
    assert np.abs(l_w[0] - l_d[0]).max() <= 1e-6 * np.abs(l_d[0]).max()
    assert np.abs(l_w - l_d).max() <= 3e-5 * np.abs(l_d).max()
    for name, g, m in (("W-only Winograd", g_1, m_1), ("product path (2-D Winograd)", g_w, m_w)):
        flips = sum(int((a != b).sum()) for a, b in zip(m, m_d))
        e_l2 = np.linalg.norm(g - g_d) / np.linalg.norm(g_d)
        e_max = np.abs(g - g_d).max() / np.abs(g_d).max()
        print("\n[train-alike] first-step gradient, %s vs direct kernels: rel L2 %.2e, max %.2e of the scale, %d activation units on the "
              "other side of the kink" % (name, e_l2, e_max, flips))
        if flips == 0:
            assert e_l2 <= 1e-5 and e_max <= 1e-5, (name, e_l2, e_max)
        else:
            assert flips <= 8 and e_l2 <= 1.5e-4 and e_max <= 5e-4, (name, flips, e_l2, e_max)
    # Adam moves noise-level gradients by +-lr either way; everything else must coincide
    dw = np.abs(w_w - w_d)
    # (the share of weights that moved differently at all grows with every gradient element a kink flip touches: 0.19 with one flip)
    assert dw.max() <= 6 * 2.1e-4 and np.quantile(dw, 0.99) <= 2e-5 and np.mean(dw > 1e-6) < 0.30


def test_per_step_repack_writes_only_the_streams_in_use_and_widens_on_a_new_grid(fdn):
    """The model re-packs, after every optimizer step, only the pack streams its grids read (fdn_conv64_pack_streams); a grid or
    algorithm that needs another stream widens the set and re-packs at once.  Checked against a model whose packs are all current:
    same weights, grids visited in an order that narrows first (P = 8: F(4,3)xF(4,3) + 1-D), then needs the direct stream (P = 10),
    then F(2,3)xF(4,3) by algorithm."""
    net = __import__("importlib").import_module("4dflownet_amd.network")
    ops = __import__("importlib").import_module("4dflownet_amd.ops")
    g = torch.Generator(device="cuda").manual_seed(3)
    m = net.FlowNetModel(2, low_resblock=1, hi_resblock=1, seed=0)
    assert m._pack_streams == [0, 0]
    full = net.FlowNetModel(2, low_resblock=1, hi_resblock=1, seed=0)
    full._pack_streams = [ops.PACK_STREAM_ALL, ops.PACK_STREAM_ALL]
    full.weights_changed()

    def step(P, algo=None):
        x = [torch.rand((2, P, P, P, 1), device="cuda", generator=g) for _ in range(6)]
        dp = torch.randn((2, 2 * P, 2 * P, 2 * P, 3), device="cuda", generator=g)
        out = []
        for mm in (m, full):
            if algo is not None:
                mm.set_conv_algo(algo)
            pred = mm.forward(x, training=True).clone()
            grads = mm.backward(dp).clone()
            mm.flat_w.add_(grads / grads.abs().max(), alpha=-1e-3)            # a (bounded) parameter update, then the per-step re-pack
            mm.weights_changed()
            out.append((pred, grads))
        assert torch.isfinite(out[1][0]).all() and torch.isfinite(out[1][1]).all()
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])

    step(8)
    assert m._pack_streams == [8, 8 | 2]
    step(8)
    pred_only = m.forward([torch.rand((2, 12, 12, 12, 1), device="cuda", generator=g) for _ in range(6)])   # inference: no dgrad streams asked
    assert m._pack_streams == [8, 8 | 2] and torch.isfinite(pred_only).all()
    step(10)                                              # 10^3 -> direct; 20^3 -> F(4,3)xF(4,3)
    assert m._pack_streams == [8 | 1, 8 | 2 | 1]
    step(8, algo="winograd_h2")
    assert m._pack_streams == [8 | 4 | 1, 8 | 4 | 2 | 1]
    step(10)
    step(8, algo="auto")
    # the untouched streams of the narrowed model really are stale: only what the mask names is written
    stale = m._packs[:, 0, 27 * 4096:81 * 4096]           # forward packs, 1-D Winograd stream: never asked for
    assert not torch.equal(stale, full._packs[:, 0, 27 * 4096:81 * 4096])


def test_multi_source_head_dgrad_gives_the_same_gradients(fdn):
    """The input gradients of the three heads' 64->64 convs as ONE multi-source launch (default) vs three chained launches
    (multi_dgrad = False): the same gradient buffer to fp32 rounding (the sum over the heads is formed in another order)."""
    batch = O.synthetic_batch(2, 8, 2, seed=23)
    grads = []
    for multi in (True, False):
        tc, _ = make(fdn, 8, 2, 2, 1, seed=7)
        tc.model.multi_dgrad = multi
        inputs, hires, venc, mask = tc._unpack(batch)
        pred = tc.model.forward(inputs, training=True)
        out, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask)
        grads.append(tc.model.backward(dpred).clone())
        torch.cuda.synchronize()
    assert torch.isfinite(grads[0]).all() and not torch.equal(grads[0], grads[1])      # (the multi-source path really ran)
    assert (grads[0] - grads[1]).abs().max().item() <= 2e-5 * grads[1].abs().max().item()
