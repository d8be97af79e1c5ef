"""Ablations of conv64_wino_kernel (test build, runtime bits that leave the instruction stream intact except for the part
switched off): python tools/abl_conv_wino.py [N] [P]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for P in ([int(sys.argv[2])] if len(sys.argv) > 2 else [48, 24]):
    x = torch.randn((N, P, P, P, 64), device="cuda"); w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.02
    wf, _ = ops.pack_conv64_weights(w); y = torch.empty_like(x)
    flop = 2.0 * 27 * 64 * 64 * N * P ** 3
    for bits, name in ((0, "full kernel"), (1, "weight stream stride 0 (L1-resident)"), (4, "no staging (no loads, no transform, no LDS writes)"),
                       (16, "staging without transform arithmetic"), (8, "no epilogue"), (32, "epilogue arithmetic without the stores"), (13, "stride 0 + no staging + no epilogue"), (64, "ONE workgroup per CU (LDS padded)"), (77, "one workgroup per CU, stride 0 + no staging + no epilogue"), (0, "full kernel again")):
        lib.fdn_debug_set_conv64_wino_dbg(bits)
        for _ in range(3): ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wf, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wf, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("P=%d %-52s %7.3f ms  %6.1f TF algorithmic, matrix pipe %.3f" % (P, name, ms, flop / ms * 1e-9, flop / 2 / ms * 1e-9 / 157.3))
    lib.fdn_debug_set_conv64_wino_dbg(0)
