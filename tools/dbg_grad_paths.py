"""First-step gradient of a small network on the three conv paths (2-D Winograd, 1-D Winograd, direct): pairwise relative L2 / max,
overall and per layer.  Separates kernel error from ReLU-kink flips (debugging aid; the oracle comparison lives in tests/)."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
trainer = importlib.import_module("4dflownet_amd.trainer")
P, R, LB, HB, B = 8, 2, 2, 1, 2

def synthetic_batch(B, P, R, seed):
    r = np.random.default_rng(seed)
    lr = [r.uniform(-1, 1, (B, P, P, P)).astype(np.float32) for _ in range(3)]
    mg = [r.uniform(0, 1, (B, P, P, P)).astype(np.float32) for _ in range(3)]
    hr = [r.uniform(-1, 1, (B, P * R, P * R, P * R)).astype(np.float32) for _ in range(3)]
    mask = (r.uniform(0, 1, (B, P * R, P * R, P * R)) > 0.3).astype(np.float32)
    venc = np.ones((B,), np.float32)
    return tuple(lr + mg + hr + [venc, mask])
batch = synthetic_batch(B, P, R, 77)
g = {}
for name in ("auto", "winograd_w", "direct"):
    tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=3, conv_algo=name)
    tc.train_step(batch)
    torch.cuda.synchronize()
    g[name] = tc.model.flat_g.cpu().numpy().astype(np.float64)[:-1]
    layers = [(L.name, L.w_off, L.b_off, L.b_off + L.cout, L.k, L.cin, L.cout) for L in tc.model.layers]
def cmp(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b), np.abs(a - b).max() / np.abs(b).max()
for a, b in (("auto", "direct"), ("winograd_w", "direct"), ("auto", "winograd_w")):
    print("%-11s vs %-11s: rel L2 %.2e  max %.2e" % ((a, b) + cmp(g[a], g[b])))
d = np.abs(g["auto"] - g["direct"]); i = int(d.argmax())
print("largest |diff| %.3e at flat index %d of %d (value %.4e)" % (d[i], i, d.size, g["direct"][i]))
order = np.argsort(-d)[:10]
print("top-10 diff indices", order.tolist(), "values", np.round(g["direct"][order], 5).tolist(), "diffs", d[order].tolist())

print("per layer (kernel / bias gradient): 2-D vs direct rel L2 | 1-D vs direct rel L2")
for name, wo, bo, be, k, ci, co in layers:
    if (k, ci, co) != (3, 64, 64) and name not in ("conv3d", "conv3d_2"):
        continue
    r = lambda a, b, lo, hi: np.linalg.norm(g[a][lo:hi] - g[b][lo:hi]) / max(np.linalg.norm(g[b][lo:hi]), 1e-30)
    print(" %-10s k%d %3d->%-3d  w %.2e | %.2e   b %.2e | %.2e" % (name, k, ci, co, r("auto", "direct", wo, bo), r("winograd_w", "direct", wo, bo),
                                                                   r("auto", "direct", bo, be), r("winograd_w", "direct", bo, be)))
