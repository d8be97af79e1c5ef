"""Layer-by-layer batch independence of the bf16 network forward (debug aid)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import flownet_oracle as O
T = importlib.import_module("test_gpu_bf16_train")
P, R, LB, HB = [int(a) for a in sys.argv[1:5]]
tc, _ = T.make(P, R, LB, HB, seed=5, dtype=sys.argv[5] if len(sys.argv) > 5 else "bfloat16", wscale=1.0)
batch = O.synthetic_batch(2, P, R, seed=51)
def fwd(sl):
    inputs, hires, venc, mask = tc._unpack(tuple(a[sl] for a in batch))
    pred = tc.model.forward(inputs, training=True)
    c = tc.model._cache
    return pred, c
p2, c2 = fwd(slice(0, 2))
p1, c1 = fwd(slice(0, 1))
for k in ("phase", "pc", "a0", "a1", "p0", "p1", "c0", "c1"):
    print(k, torch.equal(c2[k][0], c1[k][0]), (c2[k][0].float() - c1[k][0].float()).abs().max().item())
for i in range(len(c2["blocks"])):
    for j, nm in ((1, "h"), (2, "out")):
        a, b = c2["blocks"][i][j], c1["blocks"][i][j]
        print("block", i, nm, tuple(a.shape), torch.equal(a[0], b[0]), (a[0].float() - b[0].float()).abs().max().item())
if c2["up"] is not None:
    print("up", torch.equal(c2["up"][1].t[0], c1["up"][1].t[0]))
print("pred", torch.equal(p2[0], p1[0]), (p2[0] - p1[0]).abs().max().item())
