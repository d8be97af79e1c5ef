"""64->64 layer times on the low-res grids of patch sizes off the multiple-of-4 grid (P = 10, 18, 22: legal per the reference's README)
beside their neighbours: FDN_ALGO_AUTO falls to the direct kernels where W % 4 != 0.  VERDICT r5 item 7: how much is lost?
   python tools/bench_small_grids.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
ops = fdn.ops
from bench_wino2d import timeit  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
w = torch.randn(3, 3, 3, 64, 64, device="cuda", generator=g) * 0.05
wp, wd = ops.pack_conv64_weights(w)
N = 8
for P in (8, 10, 12, 16, 18, 20, 22, 24):
    x = torch.randn(N, P, P, P, 64, device="cuda", generator=g)
    res = torch.randn(N, P, P, P, 64, device="cuda", generator=g)
    out = torch.empty_like(x)
    pad = torch.empty(N, P + 2, P + 2, P + 2, 64, device="cuda")
    dxo = torch.empty_like(x)
    ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1, device="cuda")
    dw = torch.empty(3, 3, 3, 64, 64, device="cuda")
    row = []
    for algo in (ops.ALGO_AUTO, ops.ALGO_DIRECT):
        tf = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_LEAKY, residual=res, wpack=wp, out=out, algo=algo))
        td = timeit(lambda: (ops.conv3d_dgrad_fused(x, wd, pad, dxo, skip=res, y_prev=out, act=ops.ACT_LEAKY, algo=algo),
                             ops.fold_halo_border([pad], dxo, res, out, ops.ACT_LEAKY)))
        tw = timeit(lambda: ops.conv3d_wgrad(x, res, 3, 64, 64, dw=dw, workspace=ws, algo=algo))
        row.append((tf, td, tw))
    vox = N * P ** 3
    print("N=%d P=%2d (%7d voxels)  auto: fwd %.4f dgrad %.4f wgrad %.4f = %.4f ms   direct: %.4f %.4f %.4f = %.4f ms   ns/voxel auto %.2f direct %.2f"
          % (N, P, vox, *row[0], sum(row[0]), *row[1], sum(row[1]), sum(row[0]) / vox * 1e6, sum(row[1]) / vox * 1e6), flush=True)
