"""what does ONE workgroup per CU cost the 2-D Winograd forward?  The product kernel (two workgroups per CU,
256 registers per wave) beside the same body at __launch_bounds__(256, 1) with > 80 KB of LDS requested (one workgroup per CU, 512
registers per wave) and product / deeper operand rings / more staging items in flight.   python tools/abl_occ1.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
ops = fdn.ops
from bench_wino2d import timeit  # noqa: E402

VARIANTS = ((0, "product: 2 WG/CU, rings 6/3, 1 item"), (1, "1 WG/CU, rings 6/3, 1 item"), (2, "1 WG/CU, rings 12/6, 1 item"),
            (4, "1 WG/CU, rings 8/4, 2 items"), (3, "1 WG/CU, rings 12/6, 3 items"), (5, "1 WG/CU, rings 24/6, 3 items"))


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    w = torch.randn(3, 3, 3, 64, 64, device="cuda", generator=g) * 0.05
    wp, _ = ops.pack_conv64_weights(w)
    with fdn._lib.test_build() as lib:
        for N, P in ((8, 48), (8, 24)):
            x = torch.randn(N, P, P, P, 64, device="cuda", generator=g)
            ref = None
            for v, what in VARIANTS:
                lib.fdn_debug_set_conv64_wino2d_variant(v)
                out = torch.empty_like(x)
                t = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wp, out=out))
                if ref is None:
                    ref = out
                print("fwd N=%d P=%d  %-40s: %.3f ms   bit-equal to product: %s" % (N, P, what, t, bool(torch.equal(out, ref))), flush=True)
            for bits, name in ((4, "no staging"), (8, "no epilogue"), (12, "K loop only")):
                for v in (0, 2, 3):
                    lib.fdn_debug_set_conv64_wino2d_variant(v)
                    lib.fdn_debug_set_conv64_wino2d_dbg(bits)
                    t = timeit(lambda: ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wp, out=out))
                    print("   ablation %-12s variant %d: %.3f ms" % (name, v, t), flush=True)
                lib.fdn_debug_set_conv64_wino2d_dbg(0)
            lib.fdn_debug_set_conv64_wino2d_variant(0)


if __name__ == "__main__":
    main()
