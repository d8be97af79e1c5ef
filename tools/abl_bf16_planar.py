"""Timing prototype for the planar-halves bf16 activation layout [2][N][D][H][W][32]: the MODE 2 forward at (4,128^3)
with its input rows read as planar halves (debug bit 64: same bytes interpreted in the other layout -- timing and traffic only, the
results are those of a permuted input).  python tools/abl_bf16_planar.py [reps]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
bops = importlib.import_module("4dflownet_amd.ops_bf16")
_tb = fdn._lib.test_build()
lib = _tb.__enter__()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


torch.manual_seed(0)
w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
wf, wd = bops.pack_conv64_weights(w)
if len(sys.argv) > 1 and sys.argv[1] in ("rows", "planar"):         # one variant only, a few launches: for a rocprofv3 --pmc pass
    x = torch.randn(4, 128, 128, 128, 64, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(x)
    lib.fdn_debug_set_conv64_bf16_dbg(64 if sys.argv[1] == "planar" else 0)
    for _ in range(5): bops.conv64_fwd(x, wf, None, 1, 0.2, None, out)
    torch.cuda.synchronize()
    lib.fdn_debug_set_conv64_bf16_dbg(0)
    sys.exit(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for N, P in [(4, 128), (4, 32)]:
    x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
    res = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(x)
    pad = torch.empty(N, P + 2, P + 2, P + 2, 64, device="cuda")
    for rep in range(reps):
        for bits in ((0, 64) if rep % 2 == 0 else (64, 0)):
            lib.fdn_debug_set_conv64_bf16_dbg(bits)
            f = timeit(lambda: bops.conv64_fwd(x, wf, None, 1, 0.2, None, out))
            fr = timeit(lambda: bops.conv64_fwd(x, wf, None, 2, 0.2, res, out))
            dg = timeit(lambda: bops.conv64_dgrad_fused(x, wd, pad, out, skip=res, y_prev=res, act=2))
            print("N=%d P=%-3d input %-8s: fwd %.3f ms  fwd+res+leaky %.3f ms  fused dgrad %.3f ms" % (N, P, "planar" if bits else "rows", f, fr, dg), flush=True)
lib.fdn_debug_set_conv64_bf16_dbg(0)
