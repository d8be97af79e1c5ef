"""Headline benchmark: 3-D patches/sec of the full 4DFlowNet train step (forward + loss + backward +
gradient all-reduce + Adam) at the paper-default configuration (BASELINE.json configs[1]):
patch 24, res x2, per-GPU batch 8, 8 low-res + 4 hi-res ResBlocks, fp32, synthetic inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
(N > 1: launched by torch.distributed.run, one rank per GPU, RCCL sum-all-reduce of the flat gradient.)

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the 64->64 3x3x3 MFMA conv, shared by
forward and dgrad), with its launch durations measured live by HIP events on the launch stream inside the
timed region; `cpu_baseline` times the CPU oracle (numpy restatement, kind "port") on a bounded sample."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6       # dense bf16: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (tools/mfma_peak_bf16.hip sustains 1895)
FLOP_PER_VOXEL_CONV64 = 2.0 * 27 * 64 * 64   # 3.0576 GFLOP per 24^3 patch = 221 184 FLOP/voxel


def synthetic_batch(B, P, R, seed, device):
    """SURVEY.md section 8(d) synthetic inputs, generated with numpy default_rng(1234+rank)."""
    rng = np.random.default_rng(seed)
    lr = lambda lo, hi: torch.from_numpy(rng.uniform(lo, hi, size=(B, P, P, P, 1)).astype(np.float32)).to(device)
    u, v, w = lr(-1, 1), lr(-1, 1), lr(-1, 1)
    mu, mv, mw = lr(0, 0.016), lr(0, 0.016), lr(0, 0.016)
    H = P * R
    hr = lambda: torch.from_numpy(rng.uniform(-0.45, 0.45, size=(B, H, H, H, 1)).astype(np.float32)).to(device)
    uh, vh, wh = hr(), hr(), hr()
    mask = torch.from_numpy((rng.uniform(size=(B, H, H, H)) < 0.12).astype(np.float32)).to(device)
    venc = torch.full((B,), 1.5, device=device)
    return (u, v, w, mu, mv, mw, uh, vh, wh, venc, mask)


class ConvTimer:
    """Brackets every 64->64 MFMA conv launch (forward and dgrad) with HIP events on the launch stream."""

    def __init__(self, ops):
        self.ops = ops
        self.records = []
        self.enabled = False
        self._fwd, self._dgrad, self._dgrad_fused = ops.conv3d_fwd, getattr(ops, "conv3d_dgrad", None), ops.conv3d_dgrad_fused

    def install(self):
        ops, rec = self.ops, self.records

        def fwd(x, w, *a, **k):
            if not self.enabled or tuple(w.shape) != (3, 3, 3, 64, 64):
                return self._fwd(x, w, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = self._fwd(x, w, *a, **k); e1.record()
            rec.append(("fwd", x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3], e0, e1))
            return out

        def dgrad(dz, w, *a, **k):
            if not self.enabled or tuple(w.shape) != (3, 3, 3, 64, 64):
                return self._dgrad(dz, w, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = self._dgrad(dz, w, *a, **k); e1.record()
            rec.append(("dgrad", dz.shape[0] * dz.shape[1] * dz.shape[2] * dz.shape[3], e0, e1))
            return out

        def dgrad_fused(dz, *a, **k):
            if not self.enabled:
                return self._dgrad_fused(dz, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = self._dgrad_fused(dz, *a, **k); e1.record()
            rec.append(("dgrad", dz.shape[0] * dz.shape[1] * dz.shape[2] * dz.shape[3], e0, e1))
            return out

        ops.conv3d_fwd, ops.conv3d_dgrad_fused = fwd, dgrad_fused
        if self._dgrad is not None:
            ops.conv3d_dgrad = dgrad

    def summary(self):
        n = len(self.records)
        if n == 0:
            return None
        total_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in self.records)
        total_flop = sum(vox * FLOP_PER_VOXEL_CONV64 for _, vox, _, _ in self.records)
        return n, total_ms / n, total_flop / n


def pmc_traffic_bytes(fname="r1_pmc_traffic.json", kernel="conv64_mfma_kernel"):
    """HBM bytes per conv64 launch from the committed rocprofv3 PMC passes (profiles/r1_pmc_traffic.json for the fp32
    workload, r1_cfg4_pmc_traffic.json for cfg4; produced by tools/pmc_traffic.py from separate --pmc FETCH_SIZE / --pmc
    WRITE_SIZE runs of this same command, read side doubled as MI355X_MICROARCH.md prescribes for gfx950).
    Launch-weighted over the conv64 variants; None if the file is absent."""
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    rows = [(v["launches"], v["hbm_bytes_per_launch"]) for k, v in d.items() if kernel in k]
    n = sum(r[0] for r in rows)
    return sum(r[0] * r[1] for r in rows) / n if n else None


def cpu_baseline(P, R, LB, HB):
    """Time the CPU oracle (numpy float32 restatement of the same train step) on ONE patch: forward + loss +
    backward + Adam at the benchmark's network configuration.  Returns the cpu_baseline object."""
    from oracle import flownet_oracle as O          # checker / baseline only, never on the product path
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    params = O.init_params(0, LB, HB, np.float32)
    batch = O.synthetic_batch(1, P, R, seed=1234, dtype=np.float32)
    state = {}
    t0 = time.time()
    O.train_step(params, state, batch, 1e-4, R, LB, HB)
    dt = time.time() - t0
    return {"value": 1.0 / dt, "unit": "patches/s", "cores": int(cores), "kind": "port",
            "sample": "1 train step (fwd+loss+bwd+Adam) of 1 patch, P%d/R%d/LB%d/HB%d fp32, numpy oracle, %.1f s" % (P, R, LB, HB, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patch", type=int, default=24)
    ap.add_argument("--res", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--low", type=int, default=8)
    ap.add_argument("--hi", type=int, default=4)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="storage type of activations / activation gradients (bf16 = BASELINE.json configs[3] arithmetic)")
    ap.add_argument("--config", choices=["cfg2", "cfg4"], default="cfg2",
                    help="cfg2 = the headline workload (defaults above); cfg4 = patch 32, res x4, batch 4, bf16 (secondary metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.config == "cfg4":
        args.patch, args.res, args.batch, args.dtype = 32, 4, 4, "bf16"

    parallel = importlib.import_module("4dflownet_amd.parallel")
    rank, world, local_rank = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    fdn = importlib.import_module("4dflownet_amd")
    trainer = importlib.import_module("4dflownet_amd.trainer")
    fdn._lib.load()                                   # fail loudly if the HIP library is missing
    P, R, B, LB, HB = args.patch, args.res, args.batch, args.low, args.hi
    bf16 = args.dtype == "bf16"
    tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB,
                                   hi_resblock=HB, device=device, seed=0, dtype="bfloat16" if bf16 else "float32")
    batch = synthetic_batch(B, P, R, 1234 + rank, device)
    timer = ConvTimer(tc.model.ops)
    timer.install()

    for _ in range(args.warmup):
        tc.train_step(batch)
    parallel.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tc.train_step(batch)
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False

    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    if rank != 0:
        return
    n_launch, avg_ms, avg_flop = timer.summary()
    achieved = avg_flop / (avg_ms * 1e-3) / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    # forward FLOPs per patch from the layer list (328.83 GFLOP at cfg2), train step = 3x
    fwd_flop = 0.0
    hr_from = 6 + 2 * LB
    for i, (_, k, ci, co, _) in enumerate(tc.model.specs):
        vox = P ** 3 * (R ** 3 if i >= hr_from else 1)
        fwd_flop += 2.0 * k ** 3 * ci * co * vox
    line = {
        "metric": "3D patches/sec (train step, patch=%d, res×%d)" % (P, R),
        "value": args.steps * B * world / dt,
        "unit": "patches/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16" if bf16 else "f32",
        "data": "synthetic (SURVEY 8d: default_rng(1234+rank) inputs, Glorot-uniform default_rng(0) weights)",
        "config": {"workload": "%s train_step: patch_size=%d res_increase=%d batch=%d/GPU low_resblock=%d hi_resblock=%d %s"
                               % (args.config, P, R, B, LB, HB, "bf16 activations, fp32 accumulation/parameters" if bf16 else "fp32"),
                   "global_batch": B * world, "parallelism": "dp%d" % world},
        "roofline": {"bound": "mfma", "kernel": "%s (3x3x3 64->64 fwd + dgrad launches)" % ("conv64_bf16_kernel" if bf16 else "conv64_mfma_kernel"),
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak,
                     "traffic": (pmc_traffic_bytes("r1_cfg4_pmc_traffic.json", "conv64_bf16_kernel") if args.config == "cfg4" else None)
                                if bf16 else pmc_traffic_bytes(),
                     "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r1%s_pmc_traffic.json)" % ("_cfg4" if bf16 else ""),
                     "algorithmic_bytes_per_launch": avg_flop / FLOP_PER_VOXEL_CONV64 * (256.0 if bf16 else 512.0) + (221184.0 if bf16 else 442368.0),
                     "launches_timed": n_launch, "avg_launch_ms": avg_ms, "avg_launch_gflop": avg_flop / 1e9},
        "train_step_tflops": args.steps * B * world / dt * 3.0 * fwd_flop / 1e12,
    }
    if world == 1 and not args.no_cpu_baseline and not bf16:
        line["cpu_baseline"] = cpu_baseline(P, R, LB, HB)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
