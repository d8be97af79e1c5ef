"""Microbenchmark of the 64->1 head kernels (fwd / dgrad-folded / wgrad), MFMA formulation vs the VALU kernels."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
_tb = fdn._lib.test_build()                     # test build: the fdn_debug_* hooks are not in the product library
lib = _tb.__enter__()                          # (keep _tb alive: closing it restores the product library)
lib_dbg = lib


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


for dtype, N, P in (("f32", 8, 48), ("bf16", 8, 48), ("bf16", 4, 128)):
    ops = fdn.ops if dtype == "f32" else importlib.import_module("4dflownet_amd.ops_bf16")
    adt = torch.float32 if dtype == "f32" else torch.bfloat16
    x = torch.randn(N, P, P, P, 64, device="cuda").to(adt)
    w = torch.randn(3, 3, 3, 64, 1, device="cuda") * 0.1
    b = torch.randn(1, device="cuda")
    pred = torch.zeros(N, P, P, P, 3, device="cuda")
    dpred = torch.randn(N, P, P, P, 3, device="cuda")
    out = torch.empty_like(x)
    dbp = torch.empty(64, device="cuda")
    ws = torch.empty(2048 * 64, device="cuda")
    gbytes = x.numel() * x.element_size() / 1e9
    for impl in (1, 0):
        lib_dbg.fdn_debug_set_heads_mfma(impl)
        t_f = timeit(lambda: ops.conv3d_fwd(x, w, b, 0, out=pred, ldy=3, y_coff=1))
        t_d = timeit(lambda: ops.conv_cout1_dgrad_folded(dpred, w, (N, P, P, P), x, 1, lddz=3, dz_coff=1, out=out, dbias_prev=dbp, workspace=ws))
        t_w = timeit(lambda: ops.conv3d_wgrad(x, dpred, 3, 64, 1, want_bias=True, lddz=3, dz_coff=1))
        print("%s N=%d P=%d impl=%s: fwd %.3f ms (%.0f GB/s)  dgrad-folded %.3f ms (%.0f GB/s)  wgrad %.3f ms (%.0f GB/s)" %
              (dtype, N, P, "mfma" if impl else "valu", t_f, gbytes / t_f * 1e3, t_d, 2 * gbytes / t_d * 1e3, t_w, gbytes / t_w * 1e3), flush=True)
    lib_dbg.fdn_debug_set_heads_mfma(1)
