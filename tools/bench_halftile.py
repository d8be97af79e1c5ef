"""Half-size tiles of the 2-D Winograd kernel (MB = 1: one 16-cell M-block per wave) against the full tiles at the grids whose full
tiles do not fill the chip (measured, not argued).  Test build (fdn_debug_set_conv64_wino2d_mb).
   python tools/bench_halftile.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
ops = fdn.ops
from bench_wino2d import timeit  # noqa: E402


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    w = torch.randn(3, 3, 3, 64, 64, device="cuda", generator=g) * 0.05
    wp, wd = ops.pack_conv64_weights(w)
    with fdn._lib.test_build() as lib:
        for N, P in ((8, 24), (4, 24), (2, 24), (16, 24), (12, 24), (1, 48), (2, 48), (8, 48)):
            x = torch.randn(N, P, P, P, 64, device="cuda", generator=g)
            res = torch.randn(N, P, P, P, 64, device="cuda", generator=g)
            out = torch.empty_like(x)
            pad = torch.empty(N, P + 2, P + 2, P + 2, 64, device="cuda")
            dxo = torch.empty_like(x)
            ref = {}
            for mb in (2, 1):
                lib.fdn_debug_set_conv64_wino2d_mb(mb)
                for tile in ((0, 0, 0),) if mb == 2 else ((0, 0, 0), (8, 2, 1), (8, 1, 2), (4, 2, 2), (4, 4, 1), (4, 1, 4)):
                    lib.fdn_debug_set_conv64_wino2d_tile(tile[0] | tile[1] << 8 | tile[2] << 16)
                    t = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wp, out=out))
                    y1 = out.clone()
                    t2 = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_LEAKY, residual=res, wpack=wp, out=out))
                    td = timeit(lambda: (ops.conv3d_dgrad_fused(x, wd, pad, dxo, skip=res, y_prev=out, act=ops.ACT_LEAKY),
                                         ops.fold_halo_border([pad], dxo, res, out, ops.ACT_LEAKY)))
                    d1 = dxo.clone()
                    if mb == 2:
                        ref = {"y": y1, "d": d1}
                        err = ""
                    else:
                        err = "  max|diff| vs full tiles: fwd %.2e dgrad %.2e (of max|ref| %.2e / %.2e)" % (
                            (y1 - ref["y"]).abs().max().item(), (d1 - ref["d"]).abs().max().item(), ref["y"].abs().max().item(), ref["d"].abs().max().item())
                    print("N=%2d P=%d  MB=%d tile %s: fwd %.4f ms  +res+leaky %.4f  fused dgrad + border %.4f%s"
                          % (N, P, mb, "planner" if not tile[0] else "%dx%dx%d" % tile, t, t2, td, err), flush=True)
            lib.fdn_debug_set_conv64_wino2d_tile(0)
            lib.fdn_debug_set_conv64_wino2d_mb(0)


if __name__ == "__main__":
    main()
