import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
import ctypes
_tb = fdn._lib.test_build()
dbg = _tb.__enter__()
torch.manual_seed(0)
for dtype in ("f32", "bf16"):
    ops = fdn.ops if dtype == "f32" else importlib.import_module("4dflownet_amd.ops_bf16")
    adt = torch.float32 if dtype == "f32" else torch.bfloat16
    for shape in ((2, 6, 6, 6), (1, 8, 8, 8), (1, 4, 8, 8), (1, 12, 16, 16)):
        N, D, H, W = shape
        x = torch.randn(N, D, H, W, 64, device="cuda").to(adt)
        w = torch.randn(3, 3, 3, 64, 1, device="cuda") * 0.1
        b = torch.randn(1, device="cuda")
        outs = []
        for impl in (1, 0):
            dbg.fdn_debug_set_heads_mfma(impl)
            pred = torch.zeros(N, D, H, W, 3, device="cuda")
            ops.conv3d_fwd(x, w, b, 0, out=pred, ldy=3, y_coff=1)
            outs.append(pred[..., 1].clone())
        dbg.fdn_debug_set_heads_mfma(1)
        err = (outs[0] - outs[1]).abs()
        bad = err > 1e-3
        print(dtype, shape, "max err %.3e nbad %d" % (err.max().item(), bad.sum().item()), "bad by d:", [int(bad[0, d].sum().item()) for d in range(D)])
