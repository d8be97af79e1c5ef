"""Probe the depth-Winograd wgrad (FDN_ALGO_AUTO, D even) against the W-only kernel (debugging aid)."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
ops = fdn.ops
torch.manual_seed(0)
np.set_printoptions(linewidth=220, precision=3, suppress=True)
for shp in ((1, 2, 6, 8), (1, 2, 1, 4), (1, 4, 6, 8), (1, 8, 8, 8), (2, 8, 8, 8), (1, 2, 13, 7)):
    x = torch.randn(*shp, 64, device="cuda"); dz = torch.randn(*shp, 64, device="cuda")
    a, _ = ops.conv3d_wgrad(x, dz, 3, 64, 64, algo=ops.ALGO_AUTO)
    w, _ = ops.conv3d_wgrad(x, dz, 3, 64, 64, algo=ops.ALGO_WINO_W)
    a = a.cpu().numpy(); w = w.cpu().numpy()
    err = np.abs(a - w).reshape(3, 3, 3, -1).max(-1)
    print("== shape", shp, "max err %.3e of %.3e" % (err.max(), np.abs(w).max()))
    if err.max() > 1e-3:
        print(" err by (kd,kh,kw):\n", err)
        r = (a.reshape(27, -1) * w.reshape(27, -1)).sum(-1) / (w.reshape(27, -1) ** 2).sum(-1)
        print(" projection coefficient per tap:\n", r.reshape(3, 3, 3))
