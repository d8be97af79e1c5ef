"""K-loop ablations of the fp32 conv64 forward: python tools/abl_conv_f32.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
ops = fdn.ops
_tb = fdn._lib.test_build()                     # test build: the fdn_debug_* hooks are not in the product library
lib = _tb.__enter__()                          # (keep _tb alive: closing it restores the product library)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for N, P in ((8, 48), (32, 48), (8, 24)):
    x = torch.randn((N, P, P, P, 64), device="cuda")
    w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.02
    wf, wd = ops.pack_conv64_weights(w)
    y = torch.empty_like(x)
    flop = 2.0 * 27 * 64 * 64 * N * P ** 3
    for var, dbg in ((0, 0), (0, 1), (0, 4), (0, 8), (0, 13), (1, 0), (1, 13), (5, 0), (5, 13)):
        lib.fdn_debug_set_conv64_mt(var)
        lib.fdn_debug_set_conv64_dbg(dbg)
        ms = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wf, out=y))
        print("N=%d P=%d var=%d dbg=%3d: %.3f ms %.1f TF" % (N, P, var, dbg, ms, flop / ms * 1e-9))
    lib.fdn_debug_set_conv64_dbg(0)
