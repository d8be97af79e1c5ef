"""cfg2 (or, with a second argument "cfg4", the bf16 cfg4) train step with a boolean switch of the network off / on, alternating on one box:
python tools/abl_overlap.py [attr] [cfg4]
(attr: overlap_wgrad -- the weight-gradient launches on a second stream, the default; sign_masks; batch_wgrad; ...)"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
trainer = importlib.import_module("4dflownet_amd.trainer")
cfg4 = len(sys.argv) > 2 and sys.argv[2] == "cfg4"
P, R, B, LB, HB = (32, 4, 4, 8, 4) if cfg4 else (24, 2, 8, 8, 4)
rng = np.random.default_rng(1234)
f = lambda lo, hi, s: rng.uniform(lo, hi, s).astype(np.float32)
batch = tuple([f(-1, 1, (B, P, P, P, 1)) for _ in range(3)] + [f(0, 0.016, (B, P, P, P, 1)) for _ in range(3)] +
              [f(-0.45, 0.45, (B, P * R, P * R, P * R, 1)) for _ in range(3)] + [np.full((B,), 1.5, np.float32), (rng.random((B, P * R, P * R, P * R)) < 0.12).astype(np.float32)])
tc = trainer.TrainerController(P, R, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, **({"dtype": "bfloat16"} if cfg4 else {}))
nstep = 12 if cfg4 else 30
dev = tuple(tc.model._to_dev(a) for a in batch)
attr = sys.argv[1] if len(sys.argv) > 1 else "overlap_wgrad"
assert isinstance(getattr(tc.model, attr), bool), attr
res = {False: [], True: []}
for rep in range(6):
    for ov in ((False, True) if rep % 2 == 0 else (True, False)):       # alternate the order: clock / thermal drift must not pick the winner
        setattr(tc.model, attr, ov)
        for _ in range(5): tc.train_step(dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(nstep): tc.train_step(dev)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / nstep
        res[ov].append(dt * 1e3)
        print("%s=%s: %.3f ms/step" % (attr, ov, dt * 1e3), flush=True)
for ov in (False, True):
    print("%s=%s: mean %.3f ms/step, min %.3f, max %.3f over %d runs of %d steps" % (attr, ov, np.mean(res[ov]), min(res[ov]), max(res[ov]), len(res[ov]), nstep))
