"""Micro-benchmark of the MFMA kernels at the cfg2 shapes (run on the GPU box):
    python tools/bench_kernels.py [--n 8] [--iters 20]
Prints one line per kernel/shape: ms, TFLOP/s (algorithmic FLOPs), fraction of the 157.3 TF fp32-MFMA peak."""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
ops = fdn.ops
PEAK = 157.3


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
    args = ap.parse_args()
    torch.manual_seed(0)
    lib = fdn._lib.load()
    for P in args.sizes:
        N = args.n
        x = torch.randn((N, P, P, P, 64), device="cuda")
        dz = torch.randn((N, P, P, P, 64), device="cuda")
        w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.02
        wf, wd = ops.pack_conv64_weights(w)
        y = torch.empty_like(x)
        pad = torch.empty((N, P + 2, P + 2, P + 2, 64), device="cuda")
        flop = 2.0 * 27 * 64 * 64 * N * P ** 3
        for mt in (1, 2, 0):
            lib.fdn_debug_set_conv64_mt(mt)
            ms = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wf, out=y), args.iters)
            print("conv64 fwd   N=%d P=%d mt=%d : %8.3f ms  %7.2f TF  %5.1f %% of peak" % (N, P, mt, ms, flop / ms * 1e-9, flop / ms * 1e-9 / PEAK * 100))
            ms = timeit(lambda: ops.conv3d_dgrad(dz, w, wd, out=pad), args.iters)
            print("conv64 dgrad N=%d P=%d mt=%d : %8.3f ms  %7.2f TF  %5.1f %% of peak (algorithmic flops)" % (N, P, mt, ms, flop / ms * 1e-9, flop / ms * 1e-9 / PEAK * 100))
        lib.fdn_debug_set_conv64_mt(0)
        ws = torch.empty(ops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1, device="cuda")
        dw = torch.empty_like(w)
        ms = timeit(lambda: ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws), args.iters)
        print("conv64 wgrad N=%d P=%d      : %8.3f ms  %7.2f TF  %5.1f %% of peak" % (N, P, ms, flop / ms * 1e-9, flop / ms * 1e-9 / PEAK * 100))
        dxo = torch.empty_like(x)
        ms = timeit(lambda: ops.fold_halo([pad], x, y, ops.ACT_LEAKY, 0.2, out=dxo), args.iters)
        gb = (pad.numel() + 3 * x.numel()) * 4 / 1e9
        print("fold_halo    N=%d P=%d      : %8.3f ms  %7.1f GB/s" % (N, P, ms, gb / ms * 1e3))


if __name__ == "__main__":
    main()
