"""Winograd parity on REALISTIC operand distributions.

The fp32 product path runs every 64->64 3x3x3 layer through 1-D Winograd kernels (F(4,3) along W for forward / dgrad,
F(3,4) for wgrad).  Their transforms cancel (B^T rows like 4x0 - 5x2 + x4), so their error depends on the operands'
offset and dynamic range, which N(0,1) test data does not exercise.  Here:

  (a) synthetic operands with DC offset and wide range: x = |N(0,1)|*s + m for m/s in {0, 10, 100}, gradients dz with
      magnitudes log-uniform over 1e-6..1, kernels with a non-zero mean / exactly centred;
  (b) the REAL operands of the cfg2 network (paper default: patch 24, res x2, 8 + 4 ResBlocks, batch 8) after >= 300 product
      train steps on patches of the reference's example_data*.h5: every 64->64 layer's input activation, output gradient
      and trained kernel.

Every result is compared PER ELEMENT against a float64 evaluation of sampled outputs (corners, edges, faces, interior),
normalised two ways:  e_cond = |err| / sum|x||w|  (the conditioning-aware bound any fp32 summation obeys) and
e_max = max|err| / max|ref| per tensor.  The direct kernels (FDN_ALGO_DIRECT) run beside the Winograd ones and their errors
are printed in the same table.  north_star tolerance: 1e-3 relative fp32 -> asserted with a >= 10x margin (1e-4).
Semantics: src/Network/SR4DFlowNet.py:93-120 (conv3d with SYMMETRIC padding, resnet_block)."""
import contextlib
import importlib
import io
import os

import numpy as np
import pytest
import torch

from test_gpu_fullsize import gather_rows, ref_dgrad, ref_forward, ref_wgrad_rows, sample_voxels

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")

TOL = 1e-4            # north_star 1e-3 with a 10x margin
PICKS = [(0, 0, 0, 0), (1, 1, 1, 17), (2, 2, 2, 63), (0, 2, 1, 31), (2, 0, 1, 40), (1, 0, 2, 5), (1, 2, 0, 58), (0, 1, 2, 22)]


def _rel_cond(e, cond):
    """max |err| / sum|x||w| over the elements whose bound is non-zero.  Where the bound of ONE tap is zero (a dead post-ReLU
    channel; or a channel that is non-zero only in the last plane, which tap 0 never reads) the direct kernel returns an exact
    zero; the Winograd kernel forms the three W taps jointly from transformed data, so its zero is a cancellation of the other
    taps' products, exact to rounding only (measured 2e-16 next to bounds of 1e-3): such elements must stay below 1e-6 of the
    tensor's largest bound."""
    live = cond > 0
    assert (e[~live] <= 1e-6 * cond.max()).all(), (e[~live].max(), cond.max())
    return float((e[live] / cond[live]).max()) if live.any() else 0.0


def conv_errors(ops, x, w, dz, rng, n_random=160, algos=("auto", "direct")):
    """Forward, fused dgrad (+ border fold) and wgrad of one 64->64 layer with both algorithms; sampled float64 references.
    Returns {kind: {algo: (e_cond, e_max)}}."""
    N, D, H, W = x.shape[:4]
    dims = (N, D, H, W)
    pts = sample_voxels(N, D, H, W, n_random, rng)
    w64 = w.double().cpu().numpy()
    wf, wd = ops.pack_conv64_weights(w)
    xa, dza, wa = x.abs(), dz.abs(), np.abs(w64)
    ref_f, cond_f = ref_forward(x, w64, pts, dims), ref_forward(xa, wa, pts, dims)
    ref_d, cond_d = ref_dgrad(dz, w64, pts, dims), ref_dgrad(dza, wa, pts, dims)
    ref_w, cond_w = ref_wgrad_rows(x, dz, PICKS), ref_wgrad_rows(xa, dza, PICKS)
    ws = torch.empty(ops.wgrad_workspace_bytes(N, D, H, W, 64, 64, 3) // 4 + 1, device=x.device)
    out = {"fwd": {}, "dgrad": {}, "wgrad": {}}
    for name in algos:
        algo = {"auto": ops.ALGO_AUTO, "direct": ops.ALGO_DIRECT, "bf16x3": ops.ALGO_WINO_BF16X3}[name]
        y = ops.conv3d_fwd(x, w, None, ops.ACT_NONE, 0.2, None, wpack=wf, algo=algo)
        e = np.abs(gather_rows(y, pts) - ref_f)
        out["fwd"][name] = (_rel_cond(e, cond_f), float(e.max() / np.abs(ref_f).max()))
        del y
        pad = torch.empty((N, D + 2, H + 2, W + 2, 64), device=x.device)
        dx = torch.empty_like(dz)
        ops.conv3d_dgrad_fused(dz, wd, pad, dx, algo=algo)
        ops.fold_halo_border([pad], dx)
        e = np.abs(gather_rows(dx, pts) - ref_d)
        out["dgrad"][name] = (_rel_cond(e, cond_d), float(e.max() / np.abs(ref_d).max()))
        del pad, dx
        dw, _ = ops.conv3d_wgrad(x, dz, 3, 64, 64, workspace=ws, algo=algo)
        got = np.asarray([dw[a, b, c, ci].double().cpu().numpy() for (a, b, c, ci) in PICKS])
        e = np.abs(got - ref_w)
        out["wgrad"][name] = (_rel_cond(e, cond_w), float(e.max() / np.abs(ref_w).max()))
    return out


def fmt(tag, res):
    cells = []
    for kind in ("fwd", "dgrad", "wgrad"):
        (wc, wm), (dc, dm) = res[kind]["auto"], res[kind]["direct"]
        cells.append("%s wino %.1e/%.1e direct %.1e/%.1e" % (kind, wc, wm, dc, dm))
        if "bf16x3" in res[kind] and kind != "wgrad":          # (the weight gradient of FDN_ALGO_WINO_BF16X3 is FDN_ALGO_AUTO's)
            cells[-1] += " bf16x3 %.1e/%.1e" % res[kind]["bf16x3"]
    return "%-34s | %s" % (tag, " | ".join(cells))


@pytest.mark.parametrize("m_over_s", [0, 10, 100])
@pytest.mark.parametrize("wmode", ["shifted", "centred"])
def test_wino_kernels_on_offset_and_wide_range_operands(fdn, m_over_s, wmode, capsys):
    """(a): error normalised by sum|x||w| must stay <= 1e-4 whatever the offset; the output-scale error is printed beside the
    direct kernel's (with a DC offset and zero-mean kernels the RESULT is a small difference of large terms for any fp32
    summation order -- that loss is the problem's conditioning, visible in the direct column as well)."""
    ops = fdn.ops
    rng = np.random.default_rng(100 + m_over_s)
    g = torch.Generator(device="cuda").manual_seed(7 + m_over_s)
    N, P = 2, 24
    s = 0.7
    x = torch.randn((N, P, P, P, 64), device="cuda", generator=g).abs() * s + m_over_s * s
    mag = 10.0 ** (-6.0 * torch.rand((N, P, P, P, 64), device="cuda", generator=g))
    dz = mag * torch.sign(torch.randn((N, P, P, P, 64), device="cuda", generator=g))
    w = torch.randn((3, 3, 3, 64, 64), device="cuda", generator=g) * 0.03
    if wmode == "shifted":
        w = w + 0.015                                   # trained kernels drift away from zero mean
    else:
        w = w - w.mean(dim=(0, 1, 2, 3), keepdim=True)  # exactly zero mean per output channel: the DC part of x cancels
    res = conv_errors(ops, x, w, dz, rng, algos=("auto", "bf16x3", "direct"))
    with capsys.disabled():
        print("\n[wino-parity a] " + fmt("m/s=%d kernels %s" % (m_over_s, wmode), res))
    for kind in ("fwd", "dgrad", "wgrad"):
        for name in ("auto", "bf16x3"):
            e_cond, e_max = res[kind][name]
            assert e_cond <= TOL, (kind, name, res[kind])
            # forward sees the offset operand x; dgrad / wgrad results are sums over dz (no offset): those must also meet the
            # output-scale tolerance.  wgrad contracts x with dz, so the DC offset of x scales signal and error alike.
            if kind != "fwd" or m_over_s == 0 or wmode == "shifted":
                assert e_max <= TOL, (kind, name, res[kind])
        # the bf16 x 3 products (FDN_ALGO_WINO_BF16X3) are the same transforms with exact-split operands: no worse than the fp32-MFMA kernel
        # (a maximum over sampled elements: the same kernel moves by +-40 % between runs -- rounds 4, 5 saw 4.1e-7 .. 5.8e-7 for one table row)
        assert res[kind]["bf16x3"][0] <= 2.0 * res[kind]["auto"][0] + 1e-7, (kind, res[kind])


@pytest.fixture(scope="module")
def trained(fdn):
    """The cfg2 network after 320 product train steps (batch 8, lr 1e-4) on 64 patches of example_data*.h5, plus one recorded
    step: for every 64->64 layer its input activation x and output gradient dz as the backward pass saw them."""
    data = importlib.import_module("4dflownet_amd.data")
    data_device = importlib.import_module("4dflownet_amd.data_device")
    patch_index = importlib.import_module("4dflownet_amd.patch_index")
    trainer = importlib.import_module("4dflownet_amd.trainer")
    import tempfile
    with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(io.StringIO()):
        csv = os.path.join(td, "t.csv")
        patch_index.generate_patch_index(DATA, "example_data.h5", "example_data_HR.h5", csv, patch_size=24, n_patch=64,
                                         minimum_coverage=0.05, seed=0)
        rows = data.load_indexes(csv)
        ds = data_device.DevicePatchHandler3D(DATA, 24, 2, 8, 0.6).initialize_dataset(rows, shuffle=True, shard=(0, 1))
    tc = trainer.TrainerController(24, 2, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=8, hi_resblock=4, seed=0)
    steps = 0
    first = None
    target = int(os.environ.get("FDN_WINO_TRAIN_STEPS", "320"))        # longer runs by hand: FDN_WINO_TRAIN_STEPS=1600 pytest ... -s
    while steps < target:
        tc.reset_metrics()
        for batch in ds:
            tc.train_step(batch)
            steps += 1
        if first is None:
            first = tc.loss_metrics["train_loss"].result()
    last = tc.loss_metrics["train_loss"].result()
    assert np.isfinite(last) and last < 0.5 * first, (first, last)      # it did train
    batch = next(iter(ds))
    rec = {}
    m = tc.model
    orig = m._wgrad

    def recording_wgrad(x, dz, L, **k):
        if (L.k, L.cin, L.cout) == (3, 64, 64):
            rec[L.name] = (x, dz)
        return orig(x, dz, L, **k)
    m._wgrad = recording_wgrad
    inputs, hires, venc, mask = tc._unpack(batch)
    pred = m.forward(inputs, training=True)
    _, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask, want_grad=True)
    m.backward(dpred)
    m._wgrad = orig
    torch.cuda.synchronize()
    return {"tc": tc, "rec": rec, "batch": batch, "steps": steps, "loss": (first, last)}


def test_wino_kernels_on_trained_cfg2_operands(fdn, trained, capsys):
    """(b): every 64->64 layer of the trained cfg2 network with its real activation, gradient and kernel, Winograd vs float64,
    direct beside it.  Both normalisations <= 1e-4 on every layer and kernel."""
    ops = fdn.ops
    tc, rec = trained["tc"], trained["rec"]
    layers = [L for L in tc.model.layers if (L.k, L.cin, L.cout) == (3, 64, 64)]
    assert len(rec) == len(layers) == 30
    rng = np.random.default_rng(9)
    worst = {}
    lines = []
    for L in layers:
        x, dz = rec[L.name]
        xs = x.float()
        stats = "x mean/std %.2g/%.2g |dz| max %.1e w mean/std %.1e/%.1e" % (
            float(xs.mean()), float(xs.std()), float(dz.abs().max()), float(L.w.mean()), float(L.w.std()))
        res = conv_errors(ops, x, L.w.contiguous(), dz, rng, n_random=96, algos=("auto", "bf16x3", "direct"))
        lines.append(fmt("%s %s" % (L.name, "x".join(str(d) for d in x.shape[1:4])), res) + " | " + stats)
        for kind in res:
            for algo in res[kind]:
                k = (kind, algo)
                worst[k] = tuple(max(a, b) for a, b in zip(worst.get(k, (0, 0)), res[kind][algo]))
    with capsys.disabled():
        print("\n[wino-parity b] cfg2 after %d train steps (loss %.4f -> %.4f); columns: e_cond/e_max" % (
            trained["steps"], trained["loss"][0], trained["loss"][1]))
        for l in lines:
            print("[wino-parity b] " + l)
        print("[wino-parity b] WORST " + " | ".join("%s %s %.1e/%.1e" % (k[0], k[1], v[0], v[1]) for k, v in sorted(worst.items())))
    for (kind, algo), (e_cond, e_max) in worst.items():
        assert e_cond <= TOL and e_max <= TOL, (kind, algo, e_cond, e_max)
    # VERDICT r5 item 1's go criterion: the bf16 x 3 kernel's worst trained-layer e_cond no worse than the fp32-MFMA kernel's (same samples)
    for kind in ("fwd", "dgrad"):
        assert worst[(kind, "bf16x3")][0] <= 2.0 * worst[(kind, "auto")][0] + 1e-7, (kind, worst)     # (sampled maxima: +-40 % run to run; measured 2.7e-7 / 3.1e-7 and 3.4e-7 / 2.7e-7 forward)


def test_trained_cfg2_step_winograd_vs_direct_vs_oracle(fdn, trained, oracle, capsys):
    """End to end at the trained state: one cfg2 train step's prediction, loss and all 48 parameter gradients with the Winograd
    product path vs the same step with FDN_ALGO_DIRECT on every layer (conv_algo='direct'), and -- for one patch -- vs the
    float64 oracle.  Tolerance 1e-4 (norm) between the algorithms, north_star 1e-3 against the oracle."""
    trainer = importlib.import_module("4dflownet_amd.trainer")
    tc = trained["tc"]
    batch = trained["batch"]
    w_trained = tc.model.get_weights()

    def grads_of(conv_algo, rows, keep_acts=False):
        t = trainer.TrainerController(24, 2, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=8, hi_resblock=4,
                                      seed=0, conv_algo=conv_algo)
        t.model.set_weights(w_trained)
        sub = tuple(a[rows] for a in batch)
        inputs, hires, venc, mask = t._unpack(sub)
        pred = t.model.forward(inputs, training=True)
        c = t.model._cache                                 # (backward releases it) host copies of the activations: the kink sides
        t.kept = None if not keep_acts else {**{k: c[k].cpu() for k in ("a0", "a1", "p0", "p1", "c0", "c1")},
                                             "blocks": [(None, h.cpu(), o.cpu()) for _, h, o in c["blocks"]], "heads": [g.cpu() for g in c["heads"]]}
        out, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask, want_grad=True)
        t.model.backward(dpred)
        torch.cuda.synchronize()
        return pred.double().cpu().numpy(), out[:, 0].double().cpu().numpy(), t.model.flat_g.double().cpu().numpy(), t

    rows = slice(0, 8)
    pw, lw, gw, tw = grads_of("auto", rows)
    pd, ld, gd, td = grads_of("direct", rows)
    assert set(tw.model.conv_algo.values()) == {fdn.ops.ALGO_AUTO} and set(td.model.conv_algo.values()) == {fdn.ops.ALGO_DIRECT}
    rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
    e_pred, e_loss, e_grad = rel(pw, pd), float(np.abs(lw - ld).max() / np.abs(ld).max()), rel(gw, gd)
    per_layer = []
    for L in tw.model.layers:
        n = L.w.numel()
        per_layer.append(rel(gw[L.w_off:L.w_off + n], gd[L.w_off:L.w_off + n]))
    with capsys.disabled():
        print("\n[wino-parity c] trained cfg2 step, Winograd vs direct product path (B=8): pred %.2e loss %.2e grad %.2e, worst layer grad %.2e"
              % (e_pred, e_loss, e_grad, max(per_layer)))
    assert e_pred <= TOL and e_loss <= TOL and e_grad <= TOL and max(per_layer) <= 10 * TOL
    # one patch against the float64 oracle (CPU restatement of SR4DFlowNet.py:7-120, TrainerController.py:84-156)
    O = oracle
    p1, l1, g1, t1 = grads_of("auto", slice(0, 1), keep_acts=True)
    params = O.init_params(0, 8, 4, np.float64)          # list of {"w", "b"} in creation order: fill in the trained values
    it = iter(w_trained)
    for layer in params:
        layer["w"] = np.asarray(next(it), np.float64)
        if layer["b"] is not None:
            layer["b"] = np.asarray(next(it), np.float64)
    assert next(it, None) is None
    sub = tuple(np.asarray(a[0:1].cpu() if isinstance(a, torch.Tensor) else a[0:1], np.float64) for a in batch)
    # the oracle differentiates in the linear region the GPU forward landed in (tests/_kink.py); the units that changed side are counted
    from _kink import kink_sides
    seen = {}

    def sides_of(rc):
        seen["sides"], seen["flips"], seen["worst"] = kink_sides(t1.kept, rc)
        return seen["sides"]
    ref = O.loss_and_grads(params, sub, 2, 8, 4, f32_coeffs=True, sides=sides_of)
    gref = O.flatten(ref["grads"])
    e_l, e_g = float(np.abs(l1 - ref["mse"]).max() / np.abs(ref["mse"]).max()), rel(g1, gref)
    with capsys.disabled():
        print("[wino-parity c] one patch vs float64 oracle at the trained weights: loss %.2e grad(norm) %.2e in the GPU's linear region "
              "(%d activation units on the other side of the kink, largest %.1e of its tensor's scale)" % (e_l, e_g, seen["flips"], seen["worst"]))
    # (at the trained state the gradient is a small difference of large per-voxel terms: 9e-5 of its norm is arithmetic, measured with
    # the flips taken out; round 4 asserted 1e-3 here with the flips left in)
    assert e_l <= TOL and e_g <= 3e-4 and seen["flips"] <= 200 and seen["worst"] <= 2e-5


def test_training_with_winograd_tracks_training_with_direct_kernels(fdn, capsys):
    """160 product train steps of the cfg2 network on example-data patches, once with the Winograd kernels and once with
    conv_algo='direct' (same seed, same batches in the same order): the two optimisation runs must stay on the same loss curve.
    They cannot stay bit-close -- Adam's first updates are +-lr * sign(g), so rounding-level gradient differences flip single
    weights and the trajectories separate slowly -- but a systematic error of the Winograd path would show as a different curve."""
    data = importlib.import_module("4dflownet_amd.data")
    data_device = importlib.import_module("4dflownet_amd.data_device")
    patch_index = importlib.import_module("4dflownet_amd.patch_index")
    trainer = importlib.import_module("4dflownet_amd.trainer")
    import tempfile
    with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(io.StringIO()):
        csv = os.path.join(td, "t.csv")
        patch_index.generate_patch_index(DATA, "example_data.h5", "example_data_HR.h5", csv, patch_size=24, n_patch=64,
                                         minimum_coverage=0.05, seed=0)
        rows = data.load_indexes(csv)
    curves = {}
    for algo in ("auto", "direct"):
        ds = data_device.DevicePatchHandler3D(DATA, 24, 2, 8, 0.6).initialize_dataset(rows, shuffle=True, seed=3, shard=(0, 1))
        tc = trainer.TrainerController(24, 2, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=8, hi_resblock=4, seed=0,
                                       conv_algo=algo)
        losses = []
        for ep in range(20):
            tc.reset_metrics()
            for batch in ds:
                tc.train_step(batch)
            losses.append((tc.loss_metrics["train_loss"].result(), tc.loss_metrics["train_accuracy"].result()))
        curves[algo] = np.asarray(losses)
        del tc, ds
        torch.cuda.empty_cache()
    a, d = curves["auto"], curves["direct"]
    rel = np.abs(a[:, 0] - d[:, 0]) / d[:, 0]
    with capsys.disabled():
        print("\n[wino-parity d] 20 epochs x 8 steps, train loss winograd vs direct: first %.5f / %.5f, last %.5f / %.5f, max rel diff %.2e; "
              "rel-error metric last %.2f %% / %.2f %%" % (a[0, 0], d[0, 0], a[-1, 0], d[-1, 0], rel.max(), a[-1, 1], d[-1, 1]))
    signed = (a[:, 0] - d[:, 0]) / d[:, 0]
    with capsys.disabled():
        print("[wino-parity d] per-epoch relative loss difference: " + " ".join("%+.1e" % x for x in signed))
    assert a[-1, 0] < 0.2 * a[0, 0] and d[-1, 0] < 0.2 * d[0, 0]            # both trained
    # the first epochs agree to rounding; afterwards the runs separate the way two runs of ANY two summation orders do (measured:
    # 3e-7, 3e-5, 1e-5, 1e-4, 2e-3 ... then the size of the epoch-to-epoch noise of the loss itself, ~1e-1, with changing sign)
    assert rel[:4].max() <= 1e-3
    # Two fp32 summation orders under Adam's +-lr*sign(g) updates are two chaotic trajectories: they agree while the differences are
    # still rounding noise (first epochs, asserted above) and then wander around each other -- with changing sign, no systematic
    # offset, both reaching the same loss level (round 3, 1-D Winograd: final 0.0092 vs 0.0084; round 4, 2-D forward / dgrad:
    # 0.0053 vs 0.0084 while the loss fell 28x).  Asserted: within a factor of two of each other at every epoch, mean signed offset small.
    assert rel.max() <= 0.6 and abs(signed[8:].mean()) <= 0.12
