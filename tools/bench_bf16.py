"""Microbenchmarks of the bf16 64->64 conv kernels (run on the GPU box):  python tools/bench_bf16.py [--ablate]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
bops = importlib.import_module("4dflownet_amd.ops_bf16")
_tb = fdn._lib.test_build()                     # test build: the fdn_debug_* hooks are not in the product library
lib = _tb.__enter__()                          # (keep _tb alive: closing it restores the product library)
PEAK = 2516.6   # TFLOP/s dense bf16: 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


def main():
    ablate = "--ablate" in sys.argv
    torch.manual_seed(0)
    w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
    wf, wd = bops.pack_conv64_weights(w)
    for N, P in [(8, 24), (8, 48), (4, 32), (4, 128)]:
        x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
        res = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
        out = torch.empty_like(x)
        pad = torch.empty(N, P + 2, P + 2, P + 2, 64, device="cuda")
        gflop = 2 * 27 * 64 * 64 * N * P ** 3 / 1e9
        for mt in (8, 4, 0, 32):
            lib.fdn_debug_set_conv64_bf16_mt(mt & 31)
            lib.fdn_debug_set_conv64_bf16_mode2(0 if mt & 32 else 1)
            name = {8: "<MT8>", 4: "<MT4>", 0: "auto", 32: "auto-4slice"}[mt]
            for label, fn in [
                ("fwd", lambda: bops.conv64_fwd(x, wf, None, 1, 0.2, None, out)),
                ("fwd+res+leaky", lambda: bops.conv64_fwd(x, wf, None, 2, 0.2, res, out)),
                ("dgrad fused+border", lambda: (bops.conv64_dgrad_fused(x, wd, pad, out, skip=res, y_prev=res, act=2),
                                                bops.fold_halo_border([pad], out, res, res, 2))),
            ]:
                ms = timeit(fn)
                print("conv64 bf16 %-20s %-6s N=%d P=%-3d: %8.3f ms %9.1f TF  %5.1f %% of bf16 peak" %
                      (label, name, N, P, ms, gflop / ms, 100 * gflop / ms / PEAK), flush=True)
            if ablate and mt == 8:
                for bits, what in [(1, "W stride 0"), (4, "no staging loads"), (8, "no epilogue"), (12, "K loop only"),
                                   (13, "K loop only, W stride 0")]:
                    lib.fdn_debug_set_conv64_bf16_dbg(bits)
                    ms = timeit(lambda: bops.conv64_fwd(x, wf, None, 1, 0.2, None, out))
                    print("    ablation %-26s: %8.3f ms %9.1f TF" % (what, ms, gflop / ms), flush=True)
                lib.fdn_debug_set_conv64_bf16_dbg(0)
        lib.fdn_debug_set_conv64_bf16_mt(0); lib.fdn_debug_set_conv64_bf16_mode2(1)


if __name__ == "__main__":
    main()
