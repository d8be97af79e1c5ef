"""tests/test_gpu_pipeline.py::test_full_depth_network_on_example_patches_matches_float64_reference, per conv algorithm and per layer:
gradient of the paper-default network on two example patches against the float64 torch-CPU evaluation (max error, where it sits, and
how many activation units changed side against the float64 forward -- kink flips move gradient elements by 1e-4..1e-3 of scale)."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fdn = importlib.import_module("4dflownet_amd")
trainer = importlib.import_module("4dflownet_amd.trainer")
data = importlib.import_module("4dflownet_amd.data")
O = importlib.import_module("oracle.flownet_oracle")
TC = importlib.import_module("oracle.torch_cpu")
DATA = os.path.join(ROOT, "tests", "golden", "data")
P, R, B, LB, HB = 16, 2, 2, 8, 4
rows = data.load_indexes(os.path.join(DATA, "train.csv"))[:2]
batch = next(iter(data.PatchHandler3D(DATA, P, R, B, 0.6).initialize_dataset(rows, shuffle=False, shard=(0, 1))))
params = O.init_params(0, LB, HB, np.float64)
tp = TC.to_torch_params(params, torch.float64)
tb = [torch.tensor(np.asarray(a, np.float64)) for a in batch]
tpred = TC.t_forward(tp, tb[:6], R, LB, HB)
tloss = TC.t_loss(tpred, torch.cat(tb[6:9], -1), tb[10])
leaves = [t for wb in tp for t in wb if t is not None]
tg = torch.cat([x.reshape(-1) for x in torch.autograd.grad(tloss.sum(), leaves)]).numpy()
res = {}
for algo in ("auto", "winograd_h2", "winograd_w", "direct"):
    tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=0, conv_algo=algo)
    m = tc.model
    inputs, hires, venc, mask = tc._unpack(batch)
    pred = m.forward(inputs, training=True)
    loss, dpred = tc.calculate_and_update_metrics(hires, pred, mask, 'train', True)
    g = m.backward(dpred).double().cpu().numpy()
    res[algo] = g
    d = np.abs(g - tg)
    i = int(d.argmax())
    lay = [(L.name, L.w_off, L.w_off + L.k ** 3 * L.cin * L.cout) for L in m.layers]
    where = [n for n, lo, hi in lay if lo <= i < hi]
    print("%-12s pred max err %.2e   grad rel L2 %.2e   max %.2e (of max |g| %.3f) at %d in %s, g there %.4e" % (
        algo, np.abs(pred.double().cpu().numpy() - tpred.detach().numpy()).max() / np.abs(tpred.detach().numpy()).max(),
        np.linalg.norm(g - tg) / np.linalg.norm(tg), d[i] / np.abs(tg).max(), np.abs(tg).max(), i, where, tg[i]), flush=True)
    worst = sorted(((np.abs(g[lo:hi] - tg[lo:hi]).max() / np.abs(tg).max(), np.linalg.norm(g[lo:hi] - tg[lo:hi]) / max(np.linalg.norm(tg[lo:hi]), 1e-30), n) for n, lo, hi in lay if hi > lo), reverse=True)[:5]
    print("     worst layers (max err / global max |g|, rel L2 in the layer):", ["%s %.1e %.1e" % (n, a, b) for a, b, n in worst])
for a, b in (("auto", "direct"), ("winograd_h2", "direct"), ("auto", "winograd_h2")):
    print("%s vs %s: rel L2 %.2e max %.2e" % (a, b, np.linalg.norm(res[a] - res[b]) / np.linalg.norm(res[b]), np.abs(res[a] - res[b]).max() / np.abs(res[b]).max()))

# which direction carries the F(4,3)xF(4,3) error: forward and fused dgrad forced independently
ops = fdn.ops
f0, d0 = ops.conv3d_fwd, ops.conv3d_dgrad_fused
for fa, da in ((3, 3), (0, 3), (3, 0), (0, 0)):
    ops.conv3d_fwd = lambda *a, _f=fa, **k: f0(*a, **{**k, "algo": _f})
    ops.conv3d_dgrad_fused = lambda *a, _d=da, **k: d0(*a, **{**k, "algo": _d})
    try:
        tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=0)
        m = tc.model
        inputs, hires, venc, mask = tc._unpack(batch)
        pred = m.forward(inputs, training=True)
        loss, dpred = tc.calculate_and_update_metrics(hires, pred, mask, 'train', True)
        g = m.backward(dpred).double().cpu().numpy()
    finally:
        ops.conv3d_fwd, ops.conv3d_dgrad_fused = f0, d0
    d = np.abs(g - tg)
    print("forward algo %d, dgrad algo %d (0 = auto F(4,3)^2, 3 = F(2,3)xF(4,3)): grad rel L2 %.2e  max %.2e" % (fa, da, np.linalg.norm(g - tg) / np.linalg.norm(tg), d.max() / np.abs(tg).max()))

# are the differing act' masks exact zeros of the direct forward (background of the masked example data) that come out as +-rounding
# noise of the Winograd output transform?
caches = {}
for algo in ("auto", "winograd_h2", "direct"):
    tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=0, conv_algo=algo)
    inputs, hires, venc, mask = tc._unpack(batch)
    tc.model.forward(inputs, training=True)
    c = tc.model._cache
    caches[algo] = {"c0": c["c0"], "c1": c["c1"], "blk0_h": c["blocks"][0][1], "blk0_out": c["blocks"][0][2], "blk7_out": c["blocks"][7][2],
                    "blk11_out": c["blocks"][11][2], "head0": c["heads"][0]}
for algo in ("auto", "winograd_h2"):
    for k, a in caches[algo].items():
        d = caches["direct"][k]
        flips = (a > 0) != (d > 0)
        n = int(flips.sum())
        nz = int((flips & (d == 0)).sum())
        tiny = int((flips & (a.abs() < 1e-5) & (d.abs() < 1e-5)).sum())
        print("%-12s %-10s: %8d of %9d units on the other side of the kink than the direct forward; %8d where direct is exactly 0; %8d with both |y| < 1e-5; exact zeros in direct %d, in this %d"
              % (algo, k, n, a.numel(), nz, tiny, int((d == 0).sum()), int((a == 0).sum())))
