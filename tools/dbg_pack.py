import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
rng = np.random.default_rng(17)
nl, sz = 3, 27*64*64
gaps = [5, 64, 0]; offs, pos = [], 7
for gp in gaps:
    offs.append(pos); pos += sz + gp
flat = torch.tensor(rng.normal(size=pos).astype(np.float32), device="cuda")
packs = torch.full((nl, 2, ops.CONV64_PACK_FLOATS), float("nan"), device="cuda")
ops.pack_conv64_weights_batch(flat, torch.tensor(offs, device="cuda", dtype=torch.int64), packs)
for li, o in enumerate(offs):
    wf, wd = ops.pack_conv64_weights(flat[o:o+sz].view(3,3,3,64,64))
    for name, a, b in (("fwd", packs[li,0], wf), ("dgrad", packs[li,1], wd)):
        for s, lo, hi in (("direct",0,27),("1d",27,81),("h2",81,153),("h4",153,261),("h4s",261,423)):
            x = a[lo*4096:hi*4096].view(torch.int32); y = b[lo*4096:hi*4096].view(torch.int32)
            ne = (x != y)
            fa, fb = a[lo*4096:hi*4096], b[lo*4096:hi*4096]
            print(li, name, s, "int mismatch", int(ne.sum()), "nan batch/per-layer", int(torch.isnan(fa).sum()), int(torch.isnan(fb).sum()), "float-equal", bool(torch.equal(fa, fb)))
            if ne.any():
                i = int(torch.nonzero(ne)[0]); print("   first at", i, hex(int(x[i]) & 0xffffffff), hex(int(y[i]) & 0xffffffff))
