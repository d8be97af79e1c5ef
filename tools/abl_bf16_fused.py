"""bf16 fused dgrad: inner box + shell slabs as ONE launch (conv64_bf16_fused_kernel, round 5) beside the two launches of round 4 (test
build, dbg bit 32), same box, bit-equality of what they write.   python tools/abl_bf16_fused.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
bops = importlib.import_module("4dflownet_amd.ops_bf16")
from bench_bf16_dgrad import t  # noqa: E402  (prints its own table first)

w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
wf, wd = bops.pack_conv64_weights(w)
with fdn._lib.test_build() as lib:
    for N, P in ((4, 128), (4, 32), (2, 24)):
        x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16); res = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
        outs = {}
        for bits, name in ((0, "one launch"), (32, "two launches"), (0, "one launch")):
            lib.fdn_debug_set_conv64_bf16_dbg(bits)
            out = torch.zeros_like(x); pad = torch.zeros(N, P + 2, P + 2, P + 2, 64, device="cuda")
            d = t(lambda: (bops.conv64_dgrad_fused(x, wd, pad, out, skip=res, y_prev=res, act=2), bops.fold_halo_border([pad], out, res, res, 2)))
            outs.setdefault(name, (out.clone(), pad.clone()))
            print("bf16 (%d,%d^3) fused dgrad + border, %-12s: %.3f ms" % (N, P, name, d), flush=True)
        lib.fdn_debug_set_conv64_bf16_dbg(0)
        a, b = outs["one launch"], outs["two launches"]
        print("   bit-identical: dz_prev %s, padded scratch %s" % (bool(torch.equal(a[0], b[0])), bool(torch.equal(a[1], b[1]))))
