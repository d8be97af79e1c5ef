"""Forward / fused-dgrad time of the F(4,3) x F(4,3) kernel for forced tile shapes (td depth planes x ch x cw cells; test build), alternating
over several repetitions: python tools/sweep_wino2d_tile.py [reps]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, P = 8, 48
x = torch.randn((N, P, P, P, 64), device="cuda"); w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.02
res = torch.randn_like(x); wf, wd = ops.pack_conv64_weights(w); y = torch.empty_like(x)
pad = torch.empty((N, P + 2, P + 2, P + 2, 64), device="cuda"); out = torch.empty_like(x)
m = ops.new_sign_mask(y); ops.conv3d_fwd(x, w, None, ops.ACT_LEAKY, 0.2, res, wpack=wf, mask=m)
tiles = ((0, 0, 0), (8, 2, 2), (16, 1, 2), (16, 2, 1), (8, 1, 4), (8, 4, 1))
acc = {t: [[], []] for t in tiles}
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 30
with fdn._lib.test_build() as lib:
    try:
        for rep in range(reps):
            for t in (tiles if rep % 2 == 0 else tiles[::-1]):
                lib.fdn_debug_set_conv64_wino2d_tile(t[0] | t[1] << 8 | t[2] << 16)
                acc[t][0].append(timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_LEAKY, 0.2, res, wpack=wf, out=y, mask=m)))
                acc[t][1].append(timeit(lambda: ops.conv3d_dgrad_fused(x, wd, pad, out, skip=res, y_prev=None, act=ops.ACT_LEAKY, mask=m)))
    finally:
        lib.fdn_debug_set_conv64_wino2d_tile(0)
for t in tiles:
    print("tile %2dx%dx%d (0 = planner): forward + res + mask %.4f ms (min %.4f)   fused dgrad %.4f ms (min %.4f)" % (
        t + (np.mean(acc[t][0]), min(acc[t][0]), np.mean(acc[t][1]), min(acc[t][1]))))
