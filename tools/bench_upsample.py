import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters
for dtype, N, P, R in (("f32", 8, 24, 2), ("bf16", 8, 24, 2), ("bf16", 4, 32, 4)):
    ops = fdn.ops if dtype == "f32" else importlib.import_module("4dflownet_amd.ops_bf16")
    adt = torch.float32 if dtype == "f32" else torch.bfloat16
    x = torch.randn(N, P, P, P, 64, device="cuda").to(adt)
    y = torch.empty(N, P * R, P * R, P * R, 64, device="cuda", dtype=adt)
    dy = torch.randn(N, P * R, P * R, P * R, 64, device="cuda").to(adt)
    dx = torch.empty_like(x)
    tf = timeit(lambda: ops.upsample_trilinear_fwd(x, R, y))
    tb = timeit(lambda: ops.upsample_trilinear_bwd(dy, R, x, 2, 0.2, dx))
    gb = y.numel() * y.element_size() / 1e9
    if "--hb" in sys.argv:
        with fdn._lib.test_build() as lib:
            for hb in (1, 2):
                lib.fdn_debug_set_upsample_bwd_hb(hb)
                try:
                    print("   bwd with %d low-res rows per block: %.3f ms" % (hb, timeit(lambda: ops.upsample_trilinear_bwd(dy, R, x, 2, 0.2, dx))), flush=True)
                except Exception as e:
                    print("   bwd with %d rows per block: %s" % (hb, str(e)[:80]))
            lib.fdn_debug_set_upsample_bwd_hb(0)
    print("%s N=%d %d^3 x%d: fwd %.3f ms (%.0f GB/s written)  bwd %.3f ms (%.0f GB/s read)" % (dtype, N, P, R, tf, gb / tf * 1e3, tb, gb / tb * 1e3), flush=True)
