import importlib, sys, os
import numpy as np, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/tools/ -> repo root
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
from oracle import flownet_oracle as O
fdn = importlib.import_module("4dflownet_amd")
import test_gpu_train_step as T
for seed in range(4):
  for wscale in (1.0, 3.0):
    P,R,LB,HB,B = 6,2,1,1,2
    tc, params = T.make(fdn, P,R,LB,HB, seed=seed, wscale=wscale)
    batch = O.synthetic_batch(B,P,R,seed=21+seed)
    b64 = tuple(a.astype(np.float64) for a in batch)
    inputs, hires, venc, mask = tc._unpack(batch)
    pred = tc.model.forward(inputs, training=True)
    cache = tc.model._cache
    ref_pred, rc = O.network_forward(params, b64[:6], R, LB, HB, f32_coeffs=True)
    flips = {}
    for k in ("a0","a1","p0","p1","c0","c1"):
        flips[k] = int(((cache[k].cpu().numpy()>0) != (rc[k]>0)).sum())
    for i,(x,h,out) in enumerate(cache["blocks"]):
        flips["h%d"%i] = int(((h.cpu().numpy()>0) != (rc["blocks"][i][1]>0)).sum())
        flips["o%d"%i] = int(((out.cpu().numpy()>0) != (rc["blocks"][i][2]>0)).sum())
    for i,g in enumerate(cache["heads"]):
        flips["g%d"%i] = int(((g.cpu().numpy()>0) != (rc["heads"][i]>0)).sum())
    out, dpred = fdn.ops.loss_metrics(pred, hires[0], hires[1], hires[2], mask)
    g = tc.model.backward(dpred).cpu().numpy().astype(np.float64)
    ref = O.loss_and_grads(params, b64, R, LB, HB, f32_coeffs=True)
    isk = tc.model.is_kernel.cpu().numpy().astype(np.float64)
    gt = g + B*2*O.L2_LAMBDA*tc.model.flat_w.cpu().numpy().astype(np.float64)*isk
    gref = O.flatten(ref["grads"])
    errs = []
    for L in tc.model.layers:
        sl = slice(L.w_off, L.w_off+L.w.numel())
        errs.append("%s:%.1e" % (L.name.replace("conv3d","c"), T.rel_err(gt[sl], gref[sl])))
    print("seed",seed,"wscale",wscale,"pred err %.1e"%T.rel_err(pred.cpu().numpy(), ref_pred), "flips", {k:v for k,v in flips.items() if v}, " ".join(errs))
