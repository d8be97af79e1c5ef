"""Loader -> TrainerController.train_network -> checkpoint -> predictor on the reference's example data (GPU)."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import flownet_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")

data = importlib.import_module("4dflownet_amd.data")
trainer = importlib.import_module("4dflownet_amd.trainer")
predictor = importlib.import_module("4dflownet_amd.predictor")
h5io = importlib.import_module("4dflownet_amd.h5io")


def test_cfg1_train_network_checkpoint_restore_and_quicksave(tmp_path):
    """BASELINE cfg1 (patch 16, res 1, batch 2, 2 LR + 1 HR blocks) on data/train.csv rows: plumbing end to end."""
    P, R, B, LB, HB = 16, 1, 2, 2, 1
    idx = data.load_indexes(os.path.join(DATA, "train.csv"))[:6]
    val = data.load_indexes(os.path.join(DATA, "validate.csv"))[:4]
    bench = data.load_indexes(os.path.join(DATA, "benchmark.csv"))[:2]
    mk = lambda rows, sh: data.PatchHandler3D(DATA, P, R, B, 0.6).initialize_dataset(rows, shuffle=sh, shard=(0, 1))
    tc = trainer.TrainerController(P, R, initial_learning_rate=2e-4, quicksave_enable=True, network_name="t4d",
                                   low_resblock=LB, hi_resblock=HB)
    tc.init_model_dir(base_dir=str(tmp_path / "models"))
    tc.train_network(mk(idx, True), mk(val, True), n_epoch=2, testset=mk(bench, False), verbose=False)
    md = tc.model_dir
    lines = open(os.path.join(md, "loss.csv")).read().splitlines()
    header = [l for l in lines if l.startswith("epoch")][0]
    assert "train_loss,val_loss,train_accuracy,val_accuracy,train_mse,val_mse,train_div,val_div,l2_reg_loss" in header
    rows = [l for l in lines if l[:2] in ("1,", "2,")]
    assert len(rows) == 2 and rows[0].split(",")[-5 if "%" in rows[0] else -1] is not None
    assert os.path.exists(os.path.join(md, "t4d-best.h5")) and os.path.exists(os.path.join(md, "optimizer.pkl"))
    # TensorBoard epoch scalars (TrainerController.py:181-182,396-412): two writers, tags '<name>/<metric minus prefix>', step = epoch
    import glob
    tfevents = importlib.import_module("4dflownet_amd.tfevents")
    csv_vals = dict(zip(header.replace("epoch, ", "epoch,").split(",")[1:10], rows[1].split(",")[1:10]))
    for sub, prefix, extra in (("train", "train_", ["t4d/learning_rate"]), ("validate", "val_", [])):
        files = glob.glob(os.path.join(md, "tensorboard", sub, "events.out.tfevents.*"))
        assert len(files) == 1
        ev = tfevents.read_events(files[0])
        assert ev[0]["file_version"] == "brain.Event:2"
        by_step = {}
        for e in ev[1:]:
            by_step.setdefault(e["step"], {}).update(e["scalars"])
        assert sorted(by_step) == [0, 1]
        want = sorted(extra + ["t4d/" + k for k in ("loss", "accuracy", "mse", "div")])
        assert sorted(by_step[1]) == want
        for k in ("loss", "accuracy", "mse", "div"):
            assert abs(by_step[1]["t4d/" + k] - float(csv_vals[prefix + k])) <= 1e-5 + 1e-5 * abs(float(csv_vals[prefix + k]))
    q = h5io.read_all(os.path.join(md, "quicksave_t4d.h5"))
    assert q["u"].shape[1:] == (2, 16, 16, 16) and q["lr_u"].shape == (2, 16, 16, 16, 1) and q["mask"].shape == (2, 16, 16, 16)
    # the first step of a fresh controller equals the oracle on the same loader batch
    tc2 = trainer.TrainerController(P, R, initial_learning_rate=2e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB)
    batch = next(iter(mk(idx, False)))
    params = O.init_params(0, LB, HB, np.float64)
    ref = O.loss_and_grads(params, tuple(np.asarray(a, np.float64) for a in batch), R, LB, HB)
    loss = tc2.train_step(batch)
    assert np.abs(loss.cpu().numpy() - ref["loss"]).max() / np.abs(ref["loss"]).max() < 1e-4
    # restore: weights + Adam slots come back exactly
    tc3 = trainer.TrainerController(P, R, initial_learning_rate=2e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=9)
    tc.save_best_model()                      # the checkpoint now holds the controller's CURRENT state (epoch 2 need not have been the best)
    tc3.restore_model(md, "t4d-best.h5")
    best = h5io.read_keras_weights(os.path.join(md, "t4d-best.h5"))
    assert np.array_equal(tc3.model.layers[7].w.cpu().numpy(), best["conv3d_7"][0])
    assert tc3.optimizer.iterations > 0 and float(tc3.optimizer.v.abs().sum()) > 0
    # optimizer.pkl is written in Keras trainable_variables order (network.keras_layer_order) and read back through the inverse
    # permutation: the restored slots equal the saved controller's, and the file's first kernel slot is conv3d_2's (phase branch)
    assert torch.equal(tc3.optimizer.m, tc.optimizer.m) and torch.equal(tc3.optimizer.v, tc.optimizer.v)
    import pickle
    slots = pickle.load(open(os.path.join(md, "optimizer.pkl"), "rb"))
    ntv = len(tc.model.trainable_variables)
    assert len(slots) == 1 + 2 * ntv
    L2_, L0_ = tc.model.layers[2], tc.model.layers[0]
    m_flat = tc.optimizer.m.cpu().numpy()
    assert np.array_equal(slots[1].reshape(-1), m_flat[L2_.w_off:L2_.w_off + L2_.w.numel()])          # conv3d_2/kernel first
    assert np.array_equal(slots[3].reshape(-1), m_flat[L0_.w_off:L0_.w_off + L0_.w.numel()])          # then conv3d/kernel
    # the sidecar names the slots: restore maps by NAME, so a creation-order file + its sidecar restores correctly,
    # a file without a sidecar is taken as Keras order (warning), FDN_OPTIMIZER_PKL_ORDER=creation reads a pre-round-3 file
    side = os.path.join(md, "optimizer_order.txt")
    listed = [l.strip() for l in open(side) if l.strip() and not l.startswith("#")]
    assert listed[0] == "conv3d_2/kernel:0" and listed[2] == "conv3d/kernel:0" and len(listed) == ntv
    names = tc.model.trainable_variable_names()
    inv = [None] * ntv
    for slot, nm in enumerate(listed):
        inv[names.index(nm)] = slot
    legacy = [slots[0]] + [slots[1 + inv[i]] for i in range(ntv)] + [slots[1 + ntv + inv[i]] for i in range(ntv)]     # creation order
    ld = str(tmp_path / "legacy")
    os.makedirs(ld)
    import shutil
    shutil.copy(os.path.join(md, "t4d-best.h5"), ld)
    pickle.dump(legacy, open(os.path.join(ld, "optimizer.pkl"), "wb"))
    open(os.path.join(ld, "optimizer_order.txt"), "w").write("\n".join(names) + "\n")
    tc4 = trainer.TrainerController(P, R, initial_learning_rate=2e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=11)
    tc4.restore_model(ld, "t4d-best.h5")
    assert torch.equal(tc4.optimizer.m, tc.optimizer.m) and torch.equal(tc4.optimizer.v, tc.optimizer.v)
    os.remove(os.path.join(ld, "optimizer_order.txt"))
    os.environ["FDN_OPTIMIZER_PKL_ORDER"] = "creation"
    try:
        tc4.optimizer.m.zero_()
        tc4.restore_model(ld, "t4d-best.h5")
        assert torch.equal(tc4.optimizer.m, tc.optimizer.m)
    finally:
        del os.environ["FDN_OPTIMIZER_PKL_ORDER"]
    # no sidecar, no override: Keras order is assumed (with a warning).  For THIS creation-order file that is the wrong guess; here
    # the shape check catches it (conv3d's 3->64 kernel where a 64->64 one is expected) -- layers of equal shape could not be told apart
    with pytest.warns(UserWarning, match="optimizer_order.txt"), pytest.raises(ValueError, match="slot"):
        tc4.restore_model(ld, "t4d-best.h5")


def test_predictor_end_to_end_on_example_volume(tmp_path):
    """predictor.py at its defaults (patch 24, res 2, batch 8) on data/example_data.h5: 12 patches -> (84,76,72)."""
    net = predictor.prepare_network(24, 2, 2, 1)
    wpath = str(tmp_path / "w.h5")
    net.save(wpath)
    net2 = predictor.prepare_network(24, 2, 2, 1)
    net2.glorot_uniform_init(seed=123)
    net2.load_weights(wpath)
    assert torch.equal(net.flat_w, net2.flat_w)
    out = str(tmp_path / "result" / "example_result.h5")
    vols = predictor.predict_file(net2, os.path.join(DATA, "example_data.h5"), out, 24, 2, batch_size=8, verbose=False)
    back = h5io.read_all(out)
    assert back["u"].shape == (1, 84, 76, 72) and back["u"].dtype == np.float32 and back["dx"].shape == (1, 3)
    np.testing.assert_allclose(back["dx"], [[1.1875 / 2] * 3])
    assert np.array_equal(back["w"], vols[0][2].astype(np.float32))
    # one patch against the oracle forward
    ds = data.ImageDataset(); ds.load_vectorfield(os.path.join(DATA, "example_data.h5"), 0)
    pg = importlib.import_module("4dflownet_amd.tiler").PatchGenerator(24, 2)
    vel, mag = pg.patchify(ds)
    ins = [v[5:6] for v in vel] + [m[5:6] for m in mag]
    params = O.init_params(0, 2, 1, np.float64)
    ref, _ = O.network_forward(params, tuple(np.asarray(a, np.float64) for a in ins), 2, 2, 1, f32_coeffs=True)
    got = net.predict(ins)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-4
    # small velocities are zeroed: nothing in (0, venc/2048)
    nz = np.abs(back["u"][back["u"] != 0])
    assert nz.size == 0 or nz.min() >= float(ds.velocity_per_px)


def test_cfg4_grid_sizes_run_in_fp32():
    """BASELINE cfg4 geometry (patch 32, res x4 -> 128^3 HR grid, 130^3 padded) through a shortened network in fp32:
    exercises the large-grid index paths of every fp32 kernel.  (cfg4 proper -- bf16 storage -- is covered by tests/test_gpu_bf16*.py
    and, at its full launch shape, tests/test_gpu_fullsize.py.)"""
    torch.manual_seed(0)
    tc = trainer.TrainerController(32, 4, initial_learning_rate=1e-3, quicksave_enable=False, low_resblock=1, hi_resblock=1)
    batch = O.synthetic_batch(1, 32, 4, seed=3)
    losses = [float(tc.train_step(batch).cpu().numpy()[0]) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    g = tc.model.flat_g
    assert torch.isfinite(g).all() and float(g.abs().sum()) > 0
    # linearity of the HR conv at 128^3 (size-independent property; the oracle would take minutes here)
    ops = importlib.import_module("4dflownet_amd.ops")
    x = torch.randn((1, 128, 128, 128, 64), device="cuda"); w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.03
    y1 = ops.conv3d_fwd(x, w); y2 = ops.conv3d_fwd(2 * x, w)
    assert (y2 - 2 * y1).abs().max().item() <= 1e-4 * y1.abs().max().item()
    # corner voxel against a direct evaluation of the clamped stencil
    xs = x[0, :2, :2, :2].double(); idx = [0, 0, 1]
    ref = sum(xs[idx[a], idx[b], idx[c]] @ w[a, b, c].double() for a in range(3) for b in range(3) for c in range(3))
    assert (y1[0, 0, 0, 0].double() - ref).abs().max().item() < 1e-3


def test_integration_md_ctypes_stub_runs_as_written():
    """The reference-side binding shown in INTEGRATION.md section 2 is executed verbatim (a maintainer's first contact with the
    C-ABI) and must reproduce SR4DFlowNet.conv3d (pad SYMMETRIC + Conv3D + bias + ReLU) of the oracle."""
    import re
    root = os.path.dirname(HERE)
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch\n.*?)```", md, re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)                      # the stub opens the library by its repo-relative path
    try:
        exec(code, ns)
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(4)
    x = rng.normal(size=(1, 6, 5, 8, 64)).astype(np.float32)
    w = (rng.normal(size=(3, 3, 3, 64, 64)) * 0.05).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    y = ns["conv3d"](torch.tensor(x, device="cuda"), torch.tensor(w, device="cuda"), torch.tensor(b, device="cuda"), 1)
    torch.cuda.synchronize()
    ref = O.conv3d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), O.ACT_RELU, 0.2, None)
    assert np.abs(y.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()


def test_full_depth_network_on_example_patches_matches_float64_reference():
    """north_star parity statement, end to end: the paper-default architecture (8 low-res + 4 hi-res ResBlocks, 36 convolutions) with
    identical Glorot weights on real patches of data/example_data*.h5 (loader output, two rows of train.csv incl. a rotated one) --
    prediction, per-sample loss and the gradient of sum_b loss_b against an independent float64 evaluation of the same graph with stock
    torch-CPU operators (oracle/torch_cpu.py: replicate pad + conv3d, trilinear align_corners, autograd).  Tolerance of the task:
    1e-3 relative; asserted an order of magnitude tighter."""
    TC = importlib.import_module("oracle.torch_cpu")
    P, R, B, LB, HB = 16, 2, 2, 8, 4
    rows = data.load_indexes(os.path.join(DATA, "train.csv"))[:2]
    batch = next(iter(data.PatchHandler3D(DATA, P, R, B, 0.6).initialize_dataset(rows, shuffle=False, shard=(0, 1))))
    tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=0)
    m = tc.model
    inputs, hires, venc, mask = tc._unpack(batch)
    pred = m.forward(inputs, training=True)
    c = m._cache                                      # (backward releases it) the side of the kink every activation unit landed on
    acts = [c["a0"], c["a1"], c["p0"], c["p1"], c["c0"], c["c1"]] + [t for blk in c["blocks"] for t in blk[1:]] + list(c["heads"])
    masks = [(a > 0).cpu() for a in acts]
    del c, acts
    loss, dpred = tc.calculate_and_update_metrics(hires, pred, mask, 'train', True)
    g = m.backward(dpred).double().cpu().numpy()
    # float64 reference with the same parameters
    params = O.init_params(0, LB, HB, np.float64)
    assert np.array_equal(O.flatten(params).astype(np.float32), m.flat_w.cpu().numpy())
    tp = TC.to_torch_params(params, torch.float64)
    tb = [torch.tensor(np.asarray(a, np.float64)) for a in batch]
    # The network is piecewise linear; a unit whose pre-activation is within fp32 rounding of the kink may land on either side, and ONE
    # such unit moves single gradient elements by 1e-4..1e-3 of the gradient's scale.  So the float64 backward differentiates in the
    # linear region the GPU forward landed in (oracle/torch_cpu._ActWithMask, masks = sign of the GPU's stored activations) -- and the
    # units that changed side must be few and within rounding of zero in float64.  What is left is arithmetic error: asserted at 1e-4.
    zs = []
    tpred = TC.t_forward(tp, tb[:6], R, LB, HB, masks=masks, zs=zs)
    flipped = 0
    for z, mk in zip(zs, masks):
        f = (z > 0) != mk
        flipped += int(f.sum())
        if f.any():
            assert float(z[f].abs().max()) <= 2e-5 * float(z.abs().max()), "a unit far from the kink changed side"
    assert flipped <= 1e-5 * sum(mk.numel() for mk in masks), flipped
    print("[full-depth parity] %d of %d activation units on the other side of the kink than float64" % (flipped, sum(mk.numel() for mk in masks)))
    tloss = TC.t_loss(tpred, torch.cat(tb[6:9], -1), tb[10])
    leaves = [t for wb in tp for t in wb if t is not None]
    tg = torch.cat([x.reshape(-1) for x in torch.autograd.grad(tloss.sum(), leaves)]).numpy()
    rp = tpred.detach().numpy()
    assert np.abs(pred.double().cpu().numpy() - rp).max() <= 1e-4 * np.abs(rp).max()
    l2 = float(tc.calculate_regularizer_loss())
    assert np.abs(loss.double().cpu().numpy() - l2 - tloss.detach().numpy()).max() <= 1e-5 * np.abs(tloss.detach().numpy()).max()
    assert np.linalg.norm(g - tg) <= 1e-4 * np.linalg.norm(tg)
    assert np.abs(g - tg).max() <= 1e-4 * np.abs(tg).max()
