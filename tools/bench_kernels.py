"""Micro-benchmark of the MFMA kernels at the cfg2 shapes (run on the GPU box):
    python tools/bench_kernels.py [--n 8] [--iters 20] [--ablate]
Prints one line per kernel/shape/variant: ms, TFLOP/s (algorithmic FLOPs), fraction of the 157.3 TF fp32-MFMA peak."""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
ops = fdn.ops
PEAK = 157.3
VARIANTS = {0: "auto (2-D winograd)", 7: "winograd F(4,3)", 1: "<2,1,cs2>", 2: "<2,1,cs4>", 3: "<1,1,cs2>", 4: "<1,1,cs4>", 5: "<1,2,cs2>", 6: "<1,2,cs4>"}


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sizes", type=int, nargs="*", default=[24, 48])
    ap.add_argument("--ablate", action="store_true")
    ap.add_argument("--variants", type=int, nargs="*", default=[1, 2, 3, 4, 5, 6, 7, 0])
    args = ap.parse_args()
    torch.manual_seed(0)
    _tb = fdn._lib.test_build()                     # test build: the fdn_debug_* hooks are not in the product library
    lib = _tb.__enter__()                          # (keep _tb alive: closing it restores the product library)
    for P in args.sizes:
        N = args.n
        x = torch.randn((N, P, P, P, 64), device="cuda")
        dz = torch.randn((N, P, P, P, 64), device="cuda")
        w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.02
        wf, wd = ops.pack_conv64_weights(w)
        y = torch.empty_like(x)
        pad = torch.empty((N, P + 2, P + 2, P + 2, 64), device="cuda")
        dxo = torch.empty_like(x)
        flop = 2.0 * 27 * 64 * 64 * N * P ** 3
        rep = lambda name, ms: print("%-34s N=%d P=%d : %8.3f ms  %7.2f TF  %5.1f %% of peak" % (name, N, P, ms, flop / ms * 1e-9, flop / ms * 1e-9 / PEAK * 100))
        for v in args.variants:
            lib.fdn_debug_set_conv64_mt(v)
            rep("conv64 fwd %s" % VARIANTS[v], timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wf, out=y), args.iters))
            rep("conv64 fwd+res+leaky %s" % VARIANTS[v], timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_LEAKY, 0.2, dz, wpack=wf, out=y), args.iters))
            rep("conv64 dgrad(pad) %s" % VARIANTS[v], timeit(lambda: ops.conv3d_dgrad(dz, w, wd, out=pad), args.iters))
            rep("conv64 dgrad fused+border %s" % VARIANTS[v], timeit(lambda: (ops.conv3d_dgrad_fused(dz, wd, pad, dxo, skip=x, y_prev=y, act=ops.ACT_LEAKY), ops.fold_halo_border([pad], dxo, x, y, ops.ACT_LEAKY)), args.iters))
            if v == 0:
                lib.fdn_debug_set_conv64_shell_slabs(0)
                rep("conv64 dgrad fused, one padded launch", timeit(lambda: (ops.conv3d_dgrad_fused(dz, wd, pad, dxo, skip=x, y_prev=y, act=ops.ACT_LEAKY), ops.fold_halo_border([pad], dxo, x, y, ops.ACT_LEAKY)), args.iters))
                lib.fdn_debug_set_conv64_shell_slabs(1)
            if args.ablate and v in (1, 2, 5):
                for dbg, name in ((1, "B stride 0"), (4, "no staging loads"), (8, "no epilogue"), (13, "all three")):
                    lib.fdn_debug_set_conv64_dbg(dbg)
                    rep("   ablation %s" % name, timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wf, out=y), args.iters))
                lib.fdn_debug_set_conv64_dbg(0)
        lib.fdn_debug_set_conv64_mt(0)
        ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1, device="cuda")
        dw = torch.empty_like(w)
        rep("conv64 wgrad", timeit(lambda: ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws), args.iters))
        ms = timeit(lambda: ops.fold_halo([pad], x, y, ops.ACT_LEAKY, 0.2, out=dxo), args.iters)
        print("fold_halo (full)  N=%d P=%d : %8.3f ms  %7.1f GB/s" % (N, P, ms, (pad.numel() + 3 * x.numel()) * 4 / 1e9 / ms * 1e3))
        ms = timeit(lambda: ops.fold_halo_border([pad], dxo, x, y, ops.ACT_LEAKY), args.iters)
        print("fold_halo_border  N=%d P=%d : %8.3f ms" % (N, P, ms))


if __name__ == "__main__":
    main()
