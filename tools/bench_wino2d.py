"""conv64 forward at the cfg2 grids: 2-D Winograd (FDN_ALGO_AUTO) beside the 1-D kernel (FDN_ALGO_WINO_W) and the direct one,
same box, same operands; optional ablation bits of the 2-D kernel (test build).   python tools/bench_wino2d.py [--ablate]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
ops = fdn.ops


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n)
    return min(ts)


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    w = torch.randn(3, 3, 3, 64, 64, device="cuda", generator=g) * 0.05
    wp, wd = ops.pack_conv64_weights(w)
    for N, P in ((8, 48), (8, 24)):
        x = torch.randn(N, P, P, P, 64, device="cuda", generator=g)
        res = torch.randn(N, P, P, P, 64, device="cuda", generator=g)
        out = torch.empty_like(x)
        flop = 2.0 * 27 * 64 * 64 * N * P ** 3
        for name, algo in (("2-D F(4,3)xF(4,3)", ops.ALGO_AUTO), ("2-D F(4,3)^2 bf16x3", ops.ALGO_WINO_BF16X3), ("2-D F(2,3)xF(4,3)", ops.ALGO_WINO_H2), ("1-D F(4,3)", ops.ALGO_WINO_W), ("direct", ops.ALGO_DIRECT)):
            t = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wp, out=out, algo=algo))
            t2 = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_LEAKY, residual=res, wpack=wp, out=out, algo=algo))
            print("fwd %-20s N=%d P=%d : %.3f ms (%.1f TF algorithmic)   +res+leaky %.3f ms" % (name, N, P, t, flop / t / 1e9, t2), flush=True)
        pad = torch.empty(N, P + 2, P + 2, P + 2, 64, device="cuda")
        dxo = torch.empty_like(x)
        for name, algo in (("F(4,3)^2 inner + 1-D shell", ops.ALGO_AUTO), ("F(4,3)^2 bf16x3 inner + 1-D shell", ops.ALGO_WINO_BF16X3), ("F(2,3)xF(4,3) inner + 1-D shell", ops.ALGO_WINO_H2), ("1-D F(4,3), one launch", ops.ALGO_WINO_W)):
            t = timeit(lambda: (ops.conv3d_dgrad_fused(x, wd, pad, dxo, skip=res, y_prev=out, act=ops.ACT_LEAKY, algo=algo),
                                ops.fold_halo_border([pad], dxo, res, out, ops.ACT_LEAKY)))
            ti = timeit(lambda: ops.conv3d_dgrad_fused(x, wd, pad, dxo, skip=res, y_prev=out, act=ops.ACT_LEAKY, parts=1, algo=algo))
            ts = timeit(lambda: ops.conv3d_dgrad_fused(x, wd, pad, dxo, skip=res, y_prev=out, act=ops.ACT_LEAKY, parts=2, algo=algo))
            t2l = None
            if algo == ops.ALGO_AUTO and "--split" in sys.argv:     # the same with inner box and shell as two launches (until round 4's last change)
                with fdn._lib.test_build() as tlib:
                    tlib.fdn_debug_set_conv64_split_dgrad(1)
                    t2l = timeit(lambda: (ops.conv3d_dgrad_fused(x, wd, pad, dxo, skip=res, y_prev=out, act=ops.ACT_LEAKY, algo=algo),
                                          ops.fold_halo_border([pad], dxo, res, out, ops.ACT_LEAKY)))
                    tlib.fdn_debug_set_conv64_split_dgrad(0)
            print("dgrad fused+border %-32s N=%d P=%d : %.3f ms   (inner box alone %.3f, shell alone %.3f%s)"
                  % (name, N, P, t, ti, ts, "" if t2l is None else "; as two launches %.3f" % t2l), flush=True)
        for name, algo in (("F(3,2)_D x F(3,4)_W", ops.ALGO_AUTO), ("F(3,4)_W", ops.ALGO_WINO_W), ("direct", ops.ALGO_DIRECT)):
            ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1, device="cuda")
            dw = torch.empty(3, 3, 3, 64, 64, device="cuda")
            t = timeit(lambda: ops.conv3d_wgrad(x, res, 3, 64, 64, dw=dw, workspace=ws, algo=algo))
            print("wgrad %-22s N=%d P=%d : %.3f ms (%.1f TF algorithmic)" % (name, N, P, t, flop / t / 1e9), flush=True)
        if "--ablate" in sys.argv:
            with fdn._lib.test_build() as lib:
                for bits, what in ((0, "full"), (4, "no staging"), (8, "no epilogue"), (1, "weights from one unit"), (13, "K loop only"), (128, "no XCD remap"), (16, "stages 2, 4 from own rows")):
                    lib.fdn_debug_set_conv64_wino2d_dbg(bits)
                    t = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wp, out=out))
                    ts = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wp, out=out, algo=ops.ALGO_WINO_BF16X3))
                    print("   2-D ablation %-22s: %.3f ms   bf16x3: %.3f ms" % (what, t, ts), flush=True)
                lib.fdn_debug_set_conv64_wino2d_dbg(0)
                for td, ch, cw in ((8, 2, 2), (8, 1, 4), (8, 4, 1), (16, 2, 1), (16, 1, 2), (12, 2, 1), (6, 2, 2), (4, 2, 2)):
                    lib.fdn_debug_set_conv64_wino2d_tile(td | ch << 8 | cw << 16)
                    t = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wp, out=out))
                    print("   2-D tile %2dx%dx%d : %.3f ms" % (td, ch, cw, t), flush=True)
                lib.fdn_debug_set_conv64_wino2d_tile(0)


if __name__ == "__main__":
    main()
