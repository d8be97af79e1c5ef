"""Per-layer gradient error of the bf16 path vs the bf16-emulating oracle (debug aid)."""
import importlib, os, sys
import numpy as np, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/tools/ -> repo root
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
from oracle import flownet_oracle as O
fdn = importlib.import_module("4dflownet_amd")
T = importlib.import_module("test_gpu_bf16_train")
P, R, LB, HB, B = [int(a) for a in sys.argv[1:6]]
tc, params = T.make(P, R, LB, HB, seed=3, dtype="bfloat16")
batch = O.synthetic_batch(B, P, R, seed=31)
b64 = tuple(a.astype(np.float64) for a in batch)
ref = O.loss_and_grads(params, b64, R, LB, HB, f32_coeffs=True, bf16=True)
pred_ref, rc = O.network_forward(params, b64[:6], R, LB, HB, f32_coeffs=True, bf16=True)
inputs, hires, venc, mask = tc._unpack(batch)
pred = tc.model.forward(inputs, training=True)
c = tc.model._cache
for k in ("phase", "pc", "a0", "a1", "p0", "p1", "c0", "c1"):
    print("fwd %-6s %.2e" % (k, T.l2_rel(c[k].float().cpu().numpy(), rc[k])))
for i, (x, h, out) in enumerate(c["blocks"]):
    print("fwd block %d h %.2e out %.2e" % (i, T.l2_rel(h.float().cpu().numpy(), rc["blocks"][i][1]), T.l2_rel(out.float().cpu().numpy(), rc["blocks"][i][2])))
print("pred %.2e" % T.l2_rel(pred.cpu().numpy(), ref["pred"]))
out, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask)
g = tc.model.backward(dpred).cpu().numpy().astype(np.float64)
isk = tc.model.is_kernel.cpu().numpy().astype(np.float64)
g_total = g + B * 2 * O.L2_LAMBDA * tc.model.flat_w.cpu().numpy().astype(np.float64) * isk
gref = O.flatten(ref["grads"])
for L in tc.model.layers:
    sl = slice(L.w_off, L.w_off + L.w.numel())
    s = "%-10s (%d,%d,%d) w %.2e" % (L.name, L.k, L.cin, L.cout, T.l2_rel(g_total[sl], gref[sl]))
    if L.b is not None:
        sb = slice(L.b_off, L.b_off + L.cout)
        s += "  b %.2e" % T.l2_rel(g_total[sb], gref[sb])
    print(s)
