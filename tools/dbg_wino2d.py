"""Probe the 2-D Winograd forward against the direct kernel with structured operands (debugging aid)."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
ops = fdn.ops
torch.manual_seed(0)
np.set_printoptions(linewidth=250, precision=3, suppress=True)

def run(x, w, algo):
    return ops.conv3d_fwd(x, w, None, ops.ACT_NONE, algo=algo).cpu().numpy()

def report(name, x, w):
    a = run(x, w, ops.ALGO_AUTO); d = run(x, w, ops.ALGO_DIRECT)
    err = np.abs(a - d)
    print("==", name, "max err %.3e of %.3e" % (err.max(), np.abs(d).max()))
    if err.max() > 1e-4 * max(np.abs(d).max(), 1e-9):
        e = err.max(axis=-1)[0]                  # (D,H,W)
        print(" err by (d): ", e.max(axis=(1, 2)))
        print(" err by (h): ", e.max(axis=(0, 2)))
        print(" err by (w): ", e.max(axis=(0, 1)))
        print(" err by cout:", err.max(axis=(0, 1, 2, 3)))
        return a, d
    return None

N, D, H, W = 1, 2, 4, 8
x = torch.randn(N, D, H, W, 64, device="cuda")
# 1. centre tap identity
w = torch.zeros(3, 3, 3, 64, 64, device="cuda"); w[1, 1, 1] = torch.eye(64, device="cuda")
r = report("centre identity", x, w)
if r is not None:
    a, d = r
    print(" direct[0,0,0,0,:8]", d[0, 0, 0, 0, :8]); print(" wino2d[0,0,0,0,:8]", a[0, 0, 0, 0, :8])
    # where does each output come from?  correlate
    xa = x.cpu().numpy()
    for (dd, hh, ww) in ((0, 0, 0), (0, 1, 0), (0, 0, 1), (1, 2, 5)):
        v = a[0, dd, hh, ww]
        best = None
        for d2 in range(D):
            for h2 in range(H):
                for w2 in range(W):
                    for perm in ("id",):
                        c = np.abs(v - xa[0, d2, h2, w2]).max()
                        if best is None or c < best[0]: best = (c, d2, h2, w2)
        print(" out(%d,%d,%d) closest to x%s (dist %.2e)" % (dd, hh, ww, best[1:], best[0]))
# 2. single cin -> single cout at centre
for ci, co in ((0, 0), (5, 0), (0, 5), (17, 40)):
    w = torch.zeros(3, 3, 3, 64, 64, device="cuda"); w[1, 1, 1, ci, co] = 1.0
    report("centre cin %d -> cout %d" % (ci, co), x, w)
# 3. single taps
for t in ((0, 1, 1), (2, 1, 1), (1, 0, 1), (1, 2, 1), (1, 1, 0), (1, 1, 2)):
    w = torch.zeros(3, 3, 3, 64, 64, device="cuda"); w[t] = torch.eye(64, device="cuda")
    report("tap %s identity" % (t,), x, w)
w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
report("random", x, w)
for shp in ((1, 1, 2, 4), (1, 3, 2, 4), (1, 8, 8, 8), (2, 8, 8, 8)):
    report("random %s" % (shp,), torch.randn(*shp, 64, device="cuda"), w)
