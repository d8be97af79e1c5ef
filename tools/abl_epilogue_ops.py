"""What the epilogue's operand reads (residual; skip + y of the fused dgrad) cost the 2-D Winograd kernel, and how much of it is the latency of
rows that come from HBM: test-build bit 32 redirects those reads to the first 64 KB of their tensors (cache hits; timing only).
python tools/abl_epilogue_ops.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
from bench_wino2d import timeit
g = torch.Generator(device="cuda").manual_seed(0)
w = torch.randn(3, 3, 3, 64, 64, device="cuda", generator=g) * 0.05
with fdn._lib.test_build() as lib:
    wp, wd = ops.pack_conv64_weights(w)
    for N, P in ((8, 48), (8, 24)):
        x = torch.randn(N, P, P, P, 64, device="cuda", generator=g); res = torch.randn_like(x); out = torch.empty_like(x)
        pad = torch.empty(N, P + 2, P + 2, P + 2, 64, device="cuda"); dxo = torch.empty_like(x)
        for bits in (0, 32, 0, 32):
            lib.fdn_debug_set_conv64_wino2d_dbg(bits)
            t0 = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wp, out=out))
            t1 = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_LEAKY, residual=res, wpack=wp, out=out))
            t2 = timeit(lambda: ops.conv3d_dgrad_fused(x, wd, pad, dxo, skip=None, y_prev=out, act=ops.ACT_LEAKY))
            t3 = timeit(lambda: ops.conv3d_dgrad_fused(x, wd, pad, dxo, skip=res, y_prev=out, act=ops.ACT_LEAKY))
            t4 = timeit(lambda: ops.conv3d_dgrad_fused(x, wd, pad, dxo, skip=None, y_prev=None, act=ops.ACT_NONE))
            print("(%d,%d^3) bits %2d: fwd %.3f  fwd+res %.3f | fused dgrad: no operand %.3f  y %.3f  skip+y %.3f ms" % (N, P, bits, t0, t1, t4, t2, t3), flush=True)
        lib.fdn_debug_set_conv64_wino2d_dbg(0)
