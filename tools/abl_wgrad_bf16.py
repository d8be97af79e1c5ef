"""wgrad64_bf16: LDS-DMA kernel vs the register-staged kernel, with ablations (test build): python tools/abl_wgrad_bf16.py [N] [P]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); bops = importlib.import_module("4dflownet_amd.ops_bf16")
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for P in ([int(sys.argv[2])] if len(sys.argv) > 2 else [128, 32]):
    x = torch.randn((N, P, P, P, 64), device="cuda").to(torch.bfloat16); dz = torch.randn((N, P, P, P, 64), device="cuda").to(torch.bfloat16)
    dw = torch.empty((3, 3, 3, 64, 64), device="cuda")
    ws = torch.empty(bops.wgrad_workspace_bytes(N, P, P, P, 64, 64, 3) // 4 + 1024, device="cuda")
    flop = 2.0 * 27 * 64 * 64 * N * P ** 3
    ref = None
    for variant, bits, name in ((0, 0, "dma (warm-up)"), (0, 0, "dma"), (1, 0, "register-staged"), (0, 1, "dma, no staging loads"),
                                (0, 2, "dma, no LDS operand reads"), (0, 3, "dma, MFMAs + barrier only"), (0, 7, "dma, MFMAs only"), (0, 8, "dma, loads from cache-resident rows"), (0, 0, "dma"),
                                (1, 0, "register-staged")):
        lib.fdn_debug_set_wgrad64_bf16_variant(variant); lib.fdn_debug_set_wgrad64_bf16_dbg(bits)
        for _ in range(3): bops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws)
        torch.cuda.synchronize()
        if bits == 0 and P < 100:
            if ref is None: ref = dw.clone()
            else: assert torch.equal(ref, dw) or (ref - dw).abs().max() <= 1e-5 * ref.abs().max(), (name, (ref - dw).abs().max().item())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): bops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("P=%d %-30s %7.3f ms (incl. reduce)  %.3f of the nominal bf16 peak" % (P, name, ms, flop / ms * 1e-9 / 2516.6))
    lib.fdn_debug_set_wgrad64_bf16_variant(0); lib.fdn_debug_set_wgrad64_bf16_dbg(0)
