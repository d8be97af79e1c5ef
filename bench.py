"""Headline benchmark: 3-D patches/sec of the full 4DFlowNet train step (forward + loss + backward +
gradient all-reduce + Adam) at the paper-default configuration (BASELINE.json configs[1]):
patch 24, res x2, per-GPU batch 8, 8 low-res + 4 hi-res ResBlocks, fp32, synthetic inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU, RCCL sum-all-reduce of the flat gradient.  Either launched by torch.distributed.run (RANK /
WORLD_SIZE in the environment) or -- when those are absent -- bench.py re-executes itself under torch.distributed.run
with N ranks.  Asking for more ranks than GPUs is a hard error (unless --oversubscribe, a launcher self-test that puts
several ranks on one device over gloo and marks the line as such).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the 64->64 3x3x3 MFMA conv, shared by forward and
dgrad), `roofline_wgrad` for the second one (the 64->64 weight-gradient kernel); both from launch durations measured
live by HIP events on the launch stream inside the timed region.  `achieved`/`frac` count the FLOPs the kernel EXECUTES on the
matrix pipe (frac <= 1 = pipe utilisation); the fp32 kernels are Winograd kernels that execute half of the direct algorithm's
multiplies, so the ALGORITHMIC rate (221 184 FLOP per voxel) is reported beside it as `algorithmic_achieved` /
`algorithmic_frac` (> 1) / `algorithmic_speedup`.  At N=1, after the timed region: `cpu_baseline` (the same
train step on the host cores: torch-CPU/oneDNN, plus the numpy oracle as a second figure) and `secondary` (a sustained run of
>= 300 steps with per-step min/median/max, a loader-fed cfg2 run with the on-device input pipeline inside the timed loop, and a
short cfg4 bf16 run).  At N>1 the line carries `per_rank_ms_per_step` and `allreduce_exposed_ms` (HIP events around the point
where the compute stream waits for the gradient all-reduce), so a scaling shortfall is attributable at first sight."""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6       # dense bf16: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz
FLOP_PER_VOXEL_CONV64 = 2.0 * 27 * 64 * 64   # 3.0576 GFLOP per 24^3 patch = 221 184 FLOP/voxel
# committed PMC traffic summaries (tools/pmc_traffic.py), newest round first
CFG2_TRAFFIC = ["r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"]
CFG4_TRAFFIC = ["r5_cfg4_pmc_traffic.json", "r4_cfg4_pmc_traffic.json", "r3_cfg4_pmc_traffic.json", "r2_cfg4_pmc_traffic.json", "r1_cfg4_pmc_traffic.json"]
PMC_LIVE = None          # --pmc: {kernel name: {...hbm_bytes_per_launch}} measured by two rocprofv3 passes of this very run


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


class LaunchTimer:
    """Brackets every 64->64 3x3x3 launch (forward, dgrad, wgrad) with HIP events on the launch stream (torch's current
    stream is the stream handed through the C-ABI)."""

    def __init__(self, ops):
        self.ops = ops
        self.records = {"conv": [], "wgrad": []}
        self.enabled = False
        self._orig = {}

    def install(self):
        ops = self.ops
        shp = lambda t: tuple(t.shape[:4])

        def bracket(kind, vox, exec_flop, fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = fn(); e1.record()
            self.records[kind].append((vox, e0, e1, exec_flop))
            return out

        fwd, dgf, wg = ops.conv3d_fwd, ops.conv3d_dgrad_fused, ops.conv3d_wgrad
        self._orig = {"conv3d_fwd": fwd, "conv3d_dgrad_fused": dgf, "conv3d_wgrad": wg}

        def conv3d_fwd(x, w, *a, **k):
            if not self.enabled or tuple(w.shape) != (3, 3, 3, 64, 64):
                return fwd(x, w, *a, **k)
            N, D, H, W = shp(x)
            return bracket("conv", N * D * H * W, executed_conv64_flop(N, D, H, W, x.dtype, k.get("algo", 0)), lambda: fwd(x, w, *a, **k))

        def conv3d_dgrad_fused(dz, *a, **k):
            if not self.enabled:
                return dgf(dz, *a, **k)
            N, D, H, W = shp(dz)
            if dz.dtype != torch.float32:
                # bf16 mode: ONE launch covers the inner box and the shell (conv64_bf16.hip); priced as algorithmic work
                return bracket("conv", N * D * H * W, N * D * H * W * FLOP_PER_VOXEL_CONV64, lambda: dgf(dz, *a, **k))
            # fp32: ONE launch (conv64_wino2d_shell_kernel): the inner box on the 2-D Winograd body, the shell faces behind it on the 1-D
            # body -- executed FLOPs = both; a caller-issued part is priced as that part alone
            parts = k.get("parts", 3)
            ex = (executed_conv64_flop(N, D, H, W, dz.dtype, k.get("algo", 0)) if parts & 1 else 0.0) + (executed_shell_flop(N, D, H, W) if parts & 2 else 0.0)
            return bracket("conv", N * D * H * W if parts & 1 else 0, ex, lambda: dgf(dz, *a, **k))

        def conv3d_wgrad(x, dz, K, Cin, Cout, *a, **k):
            if not self.enabled or (K, Cin, Cout) != (3, 64, 64):
                return wg(x, dz, K, Cin, Cout, *a, **k)
            N, D, H, W = shp(x)
            # Winograd F(3,4) along W: 13.5 of 27 tap-equivalents; with F(3,2) along D on top (D even): 9
            f = 1.0 if x.dtype != torch.float32 or W % 4 or k.get("algo", 0) == 1 else (1.0 / 3 if D % 2 == 0 and k.get("algo", 0) == 0 else 0.5)
            return bracket("wgrad", N * D * H * W, f * N * D * H * W * FLOP_PER_VOXEL_CONV64, lambda: wg(x, dz, K, Cin, Cout, *a, **k))

        dgm = getattr(ops, "conv3d_dgrad_fused_multi", None)
        self._orig["conv3d_dgrad_fused_multi"] = dgm

        def conv3d_dgrad_fused_multi(dzs, *a, **k):
            # ONE launch over len(dzs) sources (the three heads' 64->64 convs share their input): priced as that many layers' work
            if not self.enabled:
                return dgm(dzs, *a, **k)
            N, D, H, W = shp(dzs[0])
            n = len(dzs)
            if dzs[0].dtype != torch.float32:
                return bracket("conv", n * N * D * H * W, n * N * D * H * W * FLOP_PER_VOXEL_CONV64, lambda: dgm(dzs, *a, **k))
            ex = n * (executed_conv64_flop(N, D, H, W, dzs[0].dtype, k.get("algo", 0)) + executed_shell_flop(N, D, H, W))
            return bracket("conv", n * N * D * H * W, ex, lambda: dgm(dzs, *a, **k))

        if dgm is not None:
            ops.conv3d_dgrad_fused_multi = conv3d_dgrad_fused_multi

        wgb = getattr(ops, "conv3d_wgrad_batch", None)
        self._orig["conv3d_wgrad_batch"] = wgb

        def conv3d_wgrad_batch(xs, dzs, *a, **k):
            # ONE launch over len(xs) layers of one grid (the low-res layers of a gradient bucket): priced as that many layers' work
            if not self.enabled:
                return wgb(xs, dzs, *a, **k)
            N, D, H, W = shp(xs[0])
            f = 1.0 if xs[0].dtype != torch.float32 or W % 4 or k.get("algo", 0) == 1 else (1.0 / 3 if D % 2 == 0 else 0.5)     # (bf16: direct)
            vox = len(xs) * N * D * H * W
            return bracket("wgrad", vox, f * vox * FLOP_PER_VOXEL_CONV64, lambda: wgb(xs, dzs, *a, **k))

        ops.conv3d_fwd, ops.conv3d_dgrad_fused, ops.conv3d_wgrad = conv3d_fwd, conv3d_dgrad_fused, conv3d_wgrad
        if wgb is not None:
            ops.conv3d_wgrad_batch = conv3d_wgrad_batch

    def uninstall(self):
        for name, orig in self._orig.items():
            if orig is not None:
                setattr(self.ops, name, orig)

    def summary(self, kind):
        """(launches, mean ms, mean ALGORITHMIC FLOP per launch = 221 184 x voxels, mean EXECUTED FLOP per launch)"""
        recs = self.records[kind]
        n = len(recs)
        if n == 0:
            return None
        total_ms = sum(e0.elapsed_time(e1) for _, e0, e1, _ in recs)
        total_flop = sum(vox * FLOP_PER_VOXEL_CONV64 for vox, _, _, _ in recs)
        return n, total_ms / n, total_flop / n, sum(ex for _, _, _, ex in recs) / n


def executed_conv64_flop(N, D, H, W, dtype=None, algo=0):
    """FLOPs the matrix pipe executes for one 64->64 3x3x3 forward (or fused-dgrad inner box) over N x D x H x W voxels, by the kernel
    FDN_ALGO_AUTO picks (conv64_mfma.hip: fdn_conv64_launch_ex): 2-D Winograd F(4,3)xF(4,3) = 3 x 36 / 16 = 6.75 of the 27
    tap-equivalents per voxel when H and W are multiples of 4, F(2,3)xF(4,3) = 9 when H is only even (or FDN_ALGO_WINO_H2), 1-D
    Winograd F(4,3) along W = 13.5 when only W qualifies (or FDN_ALGO_WINO_W), else all 27."""
    taps = 27.0
    if (dtype is None or dtype == torch.float32) and algo != 1:
        if W % 4 == 0:
            taps = 13.5
            if H % 2 == 0 and algo in (0, 3, 4):
                taps = 6.75 if H % 4 == 0 and algo in (0, 4) else 9.0          # (algo 4 = FDN_ALGO_WINO_BF16X3: the same products, as bf16 x 3)
    return N * D * H * W * taps * 2.0 * 64 * 64


def executed_shell_flop(N, D, H, W):
    """The shell launch of a fused dgrad (padded grid (D+2)(H+2)(W+2) minus the inner box), conv64_wino_kernel regions: the two d
    faces and the two h faces over the inner W range have ONE depth resp. height tap x 3 x 3 others, transformed along W
    (9 taps x 6/4 / ... = 4.5 tap-equivalents per position); the two w faces run 9 (kd,kh) taps x one Winograd coordinate each
    (9 tap-equivalents per position)."""
    dh_faces = 2 * (H + 2) * W + 2 * D * W
    w_faces = 2 * (D + 2) * (H + 2)
    return N * (dh_faces * 4.5 + w_faces * 9.0) * 2.0 * 64 * 64


def _weighted(d, kernel):
    rows = [(v["launches"], v["hbm_bytes_per_launch"]) for k, v in d.items() if isinstance(v, dict) and "launches" in v and kernel in k]
    n = sum(r[0] for r in rows)
    return sum(r[0] * r[1] for r in rows) / n if n else None


def pmc_traffic_bytes(fname, kernel):
    """HBM bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command,
    summarised by tools/pmc_traffic.py, read side doubled as MI355X_MICROARCH.md prescribes for gfx950): measured by THIS run when
    it was started with --pmc (PMC_LIVE), else quoted from the newest committed profiles/*_pmc_traffic.json.  Launch-weighted over
    the kernel's variants.  Returns (bytes or None, source description or None)."""
    if PMC_LIVE is not None:
        v = _weighted(PMC_LIVE, kernel)
        if v is not None:
            return v, "two rocprofv3 --pmc passes of this run (bench.py --pmc)"
    for f in ([fname] if isinstance(fname, str) else fname):
        path = os.path.join(ROOT, "profiles", f)
        if os.path.exists(path):
            d = json.load(open(path))
            v = _weighted(d, kernel)
            if v is not None:
                stamp = (d.get("_meta") or {}).get("lib_source_stamp", "")
                try:
                    cur = importlib.import_module("4dflownet_amd.build").source_stamp()
                except Exception:
                    cur = None
                state = "unstamped" if not stamp else ("measured on this library" if cur and cur.startswith(stamp[:16]) else
                                                        "STALE: measured on library %s, this one is %s" % (stamp[:16], (cur or "?")[:16]))
                return v, "profiles/%s (%s)" % (f, state)
    return None, None


def pmc_live_passes(argv_tail):
    """bench.py --pmc: re-run this command (headline only, 2 steps) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
    passes, kernel-trace only -- the combination the GPU pool allows) and summarise per kernel.  None when rocprofv3 is missing or a
    pass fails."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    tmp = tempfile.mkdtemp(prefix="fdn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", FDN_OVERLAP_WGRAD="0")      # one stream: a counter belongs to the one kernel that runs
    try:
        dirs = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            dirs[ctr] = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", dirs[ctr], "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--no-pmc", "--event-steps", "0", "--steps", "2", "--warmup", "1"] + argv_tail
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                print("bench.py --pmc: the %s pass failed: %s" % (ctr, r.stderr[-300:]), file=sys.stderr)
                return None
        return pmc_traffic.summarise(dirs["FETCH_SIZE"], dirs["WRITE_SIZE"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def host_cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def one_socket_cores():
    """One logical CPU per physical core of socket 0 (from /sys topology), or None when the topology cannot be read."""
    try:
        seen = {}
        for cpu in sorted(os.sched_getaffinity(0)):
            base = "/sys/devices/system/cpu/cpu%d/topology/" % cpu
            pkg = int(open(base + "physical_package_id").read())
            core = int(open(base + "core_id").read())
            if pkg == min([pkg] + [k[0] for k in seen]):
                seen.setdefault((pkg, core), cpu)
        pkg0 = min(k[0] for k in seen)
        cpus = sorted(v for k, v in seen.items() if k[0] == pkg0)
        return cpus or None
    except Exception:
        return None


def cpu_baseline(P, R, LB, HB):
    """The same train step (forward + loss + backward + Adam) on the GPU box's host cores, on a bounded sample.
    Primary figure: torch-CPU (oneDNN conv3d + autograd, float32) -- the closest available stand-in for the reference's TensorFlow
    CPU path, which cannot run here (TensorFlow absent) -- in a child process pinned to the physical cores of ONE socket
    (sched_setaffinity + OMP_PROC_BIND=close, OMP_PLACES=cores): median of 5 steps after 2 warm-up steps, spread reported (round 4's
    unpinned best-of-2 on all logical CPUs moved by 37 % between runs).  Second figure: the numpy oracle."""
    import subprocess
    from oracle import flownet_oracle as O          # checker / baseline only, never on the product path
    out = {"unit": "patches/s", "kind": "port", "cpu_model": host_cpu_model(), "logical_cpus": os.cpu_count()}
    cpus = one_socket_cores()
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
    if cpus:
        env["OMP_NUM_THREADS"] = str(len(cpus))
    cmd = [sys.executable, "-m", "oracle.torch_cpu", str(P), str(R), str(LB), str(HB), "2", "5"] + ([str(len(cpus))] if cpus else [])
    def pin():                                      # in the child, before exec: a refused affinity call must not cost the run
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, preexec_fn=pin if cpus else None, timeout=1800)
    if r.returncode != 0:
        raise RuntimeError("cpu_baseline child failed: " + r.stderr[-500:])
    res = json.loads(r.stdout.strip().splitlines()[-1])
    ts = sorted(res["times"])
    med = ts[len(ts) // 2]
    threads = int(res["threads"])
    out.update({"value": 1.0 / med, "cores": threads, "spread": {"min_s": ts[0], "median_s": med, "max_s": ts[-1], "steps": len(ts)},
                "sample": "train steps (fwd+loss+bwd+Adam) of 1 patch, P%d/R%d/LB%d/HB%d fp32, torch-CPU %s (oneDNN conv3d + autograd), %d threads "
                          "pinned to %s: median of %d after 2 warm-up steps %.2f s (min %.2f, max %.2f)"
                          % (P, R, LB, HB, torch.__version__, threads, ("the %d physical cores of socket 0" % len(cpus)) if cpus else "all CPUs (topology unreadable)",
                             len(ts), med, ts[0], ts[-1])})
    try:
        from threadpoolctl import threadpool_info
        blas = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas = os.cpu_count() or 1
    params = O.init_params(0, LB, HB, np.float32)
    batch = O.synthetic_batch(1, P, R, seed=1234, dtype=np.float32)
    t0 = time.time()
    O.train_step(params, {}, batch, 1e-4, R, LB, HB)
    dn = time.time() - t0
    out["numpy_oracle"] = {"value": 1.0 / dn, "unit": "patches/s", "cores": int(blas),
                           "sample": "same step, numpy float32 restatement (oracle/flownet_oracle.py), %d BLAS threads: %.1f s" % (blas, dn)}
    return out


def fwd_flop_per_patch(specs, P, R, LB):
    """Forward FLOPs per patch from the layer list (328.83 GFLOP at cfg2); a train step is 3x."""
    f = 0.0
    hr_from = 6 + 2 * LB
    for i, (_, k, ci, co, _) in enumerate(specs):
        vox = P ** 3 * (R ** 3 if i >= hr_from else 1)
        f += 2.0 * k ** 3 * ci * co * vox
    return f


def roofline_obj(timer, kind, bf16, kernel, traffic_files):
    s = timer.summary(kind)
    if s is None:
        return None
    n_launch, avg_ms, avg_flop, avg_exec = s
    if bf16:
        avg_exec = avg_flop                                   # the bf16 kernels are direct convolutions
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    achieved = avg_exec / (avg_ms * 1e-3) / 1e12              # FLOPs issued on the matrix pipe per second
    algorithmic = avg_flop / (avg_ms * 1e-3) / 1e12           # direct-convolution FLOPs per second
    traffic, tfile = pmc_traffic_bytes(traffic_files, kernel.split(" ")[0])
    vox = avg_flop / FLOP_PER_VOXEL_CONV64
    esz = 2.0 if bf16 else 4.0
    # conv: in + out rows + the weight stream; wgrad: x + dz rows + the dW it writes (fp32)
    alg = vox * 64 * esz * 2 + 27 * 64 * 64 * (4.0 if kind == "wgrad" else esz)
    return {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)" if tfile else None, "traffic_source": tfile,
            "algorithmic_bytes_per_launch": alg, "launches_timed": n_launch, "avg_launch_ms": avg_ms,
            "timing": "HIP events on the launch stream around every launch of --event-steps steps run right behind the timed region with every launch "
                      "on ONE stream (FDN_OVERLAP_WGRAD=0 semantics: the timed steps overlap the weight gradients with the dgrad chain on a second stream)",
            "executed_gflop_per_launch": avg_exec / 1e9,
            # the direct 3x3x3 algorithm's FLOPs (221 184 per voxel) over the same launch time: what the contract calls
            # ALGORITHMIC work.  > peak when the kernel executes fewer multiplies than the direct algorithm (fp32: Winograd along W)
            "algorithmic_gflop_per_launch": avg_flop / 1e9, "algorithmic_achieved": algorithmic,
            "algorithmic_frac": algorithmic / peak, "algorithmic_speedup": avg_flop / avg_exec,
            "note": None if bf16 else "achieved/frac = FLOPs the Winograd kernel EXECUTES on the fp32 MFMA pipe (forward / dgrad inner box: "
                                      "2-D F(4,3)xF(4,3), a quarter of the direct algorithm's multiplies; wgrad: F(3,2) along D x F(3,4) along W, a third) = matrix-pipe "
                                      "utilisation; PMC SQ_VALU_MFMA_BUSY_CYCLES agrees (profiles/README.md).  algorithmic_* prices the same "
                                      "launches with the direct 3x3x3 FLOP count of SURVEY 8d and therefore exceeds the peak"}


def products_note(model, bf16):
    """The arithmetic of the 64->64 contractions, in words (no reader should have to guess whether precision was traded)."""
    if bf16:
        return "bf16 activations x bf16 weights on v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 parameters (BASELINE configs[3])"
    algos = set(model.conv_algo[L.name] for L in model.layers if L.wp_f is not None)
    if algos == {4}:
        return ("FDN_ALGO_WINO_BF16X3: fp32 operands, transformed in fp32, each split EXACTLY into 3 bf16 pieces; 6 of the 9 cross terms on "
                "v_mfma_f32_16x16x32_bf16 with fp32 accumulation (dropped terms <= 2^-25 |u||v|: below an fp32 multiply's rounding); weight gradients on fp32 MFMA")
    return "exact fp32 products: v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate)" + ("" if algos == {0} else "; conv_algo %s" % sorted(algos))


def event_pass(tc, batch, timer, steps):
    """`steps` train steps with the weight gradients on the main stream and the LaunchTimer on; returns their mean wall time (ms)."""
    if steps <= 0:
        return None
    ov = tc.model.overlap_wgrad
    tc.model.overlap_wgrad = False
    try:
        tc.train_step(batch)
        torch.cuda.synchronize()
        timer.enabled = True
        t0 = time.perf_counter()
        for _ in range(steps):
            tc.train_step(batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / max(steps, 1) * 1e3
    finally:
        timer.enabled = False
        tc.model.overlap_wgrad = ov


def timed_steps(step_fn, steps, warmup, parallel):
    for _ in range(warmup):
        step_fn()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    parallel.barrier()
    return time.perf_counter() - t0


def secondary_runs(trainer, parallel, device, P, R, B, LB, HB, sustained_steps=300):
    """Short extra measurements at N=1, outside the headline timed region."""
    sec = {}
    # (0) sustained: the headline workload for >= 300 steps / >= 10 s, one HIP event per step boundary (no host sync inside):
    # the spread of the per-step times shows what the 20-step headline window can hide (clock / thermal state)
    try:
        tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB,
                                       device=device, seed=0)
        batch = synthetic_batch(B, P, R, 1234, device)
        for _ in range(5):
            tc.train_step(batch)
        torch.cuda.synchronize()
        n_sus = sustained_steps
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_sus + 1)]
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(n_sus):
            tc.train_step(batch)
            evs[i + 1].record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n_sus)])
        sec["sustained"] = {"value": n_sus * B / wall, "unit": "patches/s", "steps": n_sus, "seconds": wall,
                            "ms_per_step": {"mean": wall / n_sus * 1e3, "min": float(per.min()), "median": float(np.median(per)),
                                            "p95": float(np.percentile(per, 95)), "max": float(per.max())},
                            "ms_per_step_by_quarter": [float(q.mean()) for q in np.array_split(per, 4)],
                            "workload": "the headline cfg2 train_step, same synthetic batch, %d consecutive steps; per-step times from "
                                        "HIP events recorded at the step boundaries on the launch stream" % n_sus}
        del tc, batch, evs
    except Exception as e:
        sec["sustained"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # (1) cfg2 with the on-device input pipeline inside the timed loop: CSV rows -> fdn_gather_patches -> train_step
    try:
        data_device = importlib.import_module("4dflownet_amd.data_device")
        patch_index = importlib.import_module("4dflownet_amd.patch_index")
        data = importlib.import_module("4dflownet_amd.data")
        ddir = os.path.join(ROOT, "tests", "golden", "data")
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            csv_path = os.path.join(td, "bench%d.csv" % P)
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                patch_index.generate_patch_index(ddir, "example_data.h5", "example_data_HR.h5", csv_path, patch_size=P, n_patch=8 * B,
                                                 minimum_coverage=0.05, seed=0)
                rows = data.load_indexes(csv_path)
                ph = data_device.DevicePatchHandler3D(ddir, P, R, B, 0.6, device=device)
                ds = ph.initialize_dataset(rows, shuffle=True, shard=(0, 1))
        tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB,
                                       device=device, seed=0)
        n_full = [0]

        def epoch():
            for batch in ds:
                if batch[0].shape[0] == B:
                    tc.train_step(batch)
                    n_full[0] += 1
        epoch()                                            # warm-up epoch (uploads the volumes once)
        torch.cuda.synchronize()
        n_full[0] = 0
        t0 = time.perf_counter()
        epoch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sec["cfg2_loader_fed"] = {"value": n_full[0] * B / dt, "unit": "patches/s", "ms_per_step": dt / n_full[0] * 1e3, "steps": n_full[0],
                                  "workload": "cfg2 train_step fed by DevicePatchHandler3D (example_data*.h5 resident in HBM, %d rows with "
                                              "random rotations, shuffle, fdn_gather_patches inside the timed loop)" % len(rows)}
        del tc, ds, ph
    except Exception as e:                                 # secondary figures must never take the headline line down
        sec["cfg2_loader_fed"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # (1b) the same with the HOST loader (scripts/trainer.py under FDN_HOST_LOADER=1): PatchHandler3D, producer thread + worker pool
    # + pinned staging ring, TrainerController.device_batches (non-blocking H2D one batch ahead); and the loader alone (what one rank's host side can supply)
    try:
        data = importlib.import_module("4dflownet_amd.data")
        patch_index = importlib.import_module("4dflownet_amd.patch_index")
        ddir = os.path.join(ROOT, "tests", "golden", "data")
        import contextlib, io, tempfile
        with tempfile.TemporaryDirectory() as td:
            csv_path = os.path.join(td, "benchh%d.csv" % P)
            with contextlib.redirect_stdout(io.StringIO()):
                patch_index.generate_patch_index(ddir, "example_data.h5", "example_data_HR.h5", csv_path, patch_size=P, n_patch=8 * B,
                                                 minimum_coverage=0.05, seed=0)
                rows = data.load_indexes(csv_path)
                ph = data.PatchHandler3D(ddir, P, R, B, 0.6)
                ds = ph.initialize_dataset(rows, shuffle=True, shard=(0, 1), pinned=True)
        for _ in ds:                                       # warm the volume cache / allocate the pinned ring
            pass
        t0 = time.perf_counter()
        n_rows = sum(b[0].shape[0] for b in ds)
        dt_alone = time.perf_counter() - t0
        tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB,
                                       device=device, seed=0)
        n_full = [0]

        def epoch_h():
            for batch in tc.device_batches(ds):            # (what train_network does: batch k + 1 copied on a copy stream while step k runs)
                if batch[0].shape[0] == B:
                    tc.train_step(batch)
                    n_full[0] += 1
        epoch_h()
        torch.cuda.synchronize()
        n_full[0] = 0
        t0 = time.perf_counter()
        epoch_h()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sec["cfg2_host_loader_fed"] = {"value": n_full[0] * B / dt, "unit": "patches/s", "ms_per_step": dt / n_full[0] * 1e3, "steps": n_full[0],
                                       "loader_alone_patches_per_s": n_rows / dt_alone, "loader_threads": ds.n_parallel, "prefetch": ds.prefetch,
                                       "workload": "cfg2 train_step fed by the host loader data.PatchHandler3D (volumes cached in host memory, "
                                                   "producer thread + %d worker threads, pinned staging ring, batch k + 1 copied non-blocking on a copy stream while step k runs); "
                                                   "loader_alone = the same epoch without the train step" % ds.n_parallel}
        del tc, ds, ph
    except Exception as e:
        sec["cfg2_host_loader_fed"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # (2) cfg4: patch 32, res x4, batch 4, bf16 activations (BASELINE.json configs[3])
    try:
        P4, R4, B4 = 32, 4, 4
        tc = trainer.TrainerController(P4, R4, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB,
                                       device=device, seed=0, dtype="bfloat16")
        batch = synthetic_batch(B4, P4, R4, 1234, device)
        timer = LaunchTimer(tc.model.ops)
        timer.install()
        for _ in range(2):
            tc.train_step(batch)
        torch.cuda.synchronize()
        steps = 5
        t0 = time.perf_counter()
        for _ in range(steps):
            tc.train_step(batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        event_pass(tc, batch, timer, 3)
        timer.uninstall()
        f4 = fwd_flop_per_patch(tc.model.specs, P4, R4, LB)
        sec["cfg4_bf16"] = {"value": steps * B4 / dt, "unit": "patches/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
                            "workload": "cfg4 train_step: patch_size=32 res_increase=4 batch=4 bf16 activations, fp32 accumulation/parameters",
                            "train_step_tflops": steps * B4 / dt * 3.0 * f4 / 1e12,
                            "roofline": roofline_obj(timer, "conv", True, "conv64_bf16_kernel (3x3x3 64->64 fwd + dgrad launches)",
                                                     CFG4_TRAFFIC),
                            "roofline_wgrad": roofline_obj(timer, "wgrad", True, "wgrad64_bf16_dma_kernel (3x3x3 64->64 weight gradient)",
                                                           CFG4_TRAFFIC)}
        del tc, batch
    except Exception as e:
        sec["cfg4_bf16"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # (3) cfg5: sliding-window inference (predictor.py:67-115) -- tiler -> batched forward -> gather, host<->device copies included
    try:
        predictor = importlib.import_module("4dflownet_amd.predictor")
        tiler = importlib.import_module("4dflownet_amd.tiler")

        class _Vol:
            pass
        rng = np.random.default_rng(0)
        vol = _Vol()
        for n_ in ("u", "v", "w"):
            setattr(vol, n_, rng.uniform(-1, 1, (100, 100, 100)).astype(np.float32))
        for n_ in ("mag_u", "mag_v", "mag_w"):
            setattr(vol, n_, rng.uniform(0, 0.016, (100, 100, 100)).astype(np.float32))
        net = predictor.prepare_network(P, R, LB, HB, device=device)
        pg = tiler.PatchGenerator(P, R)
        vel, mag = pg.patchify(vol)
        predictor.predict_patches(net, vel, mag, B)                     # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = predictor.predict_patches(net, vel, mag, B)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sec["cfg5_predictor"] = {"value": len(res) / dt, "unit": "patches/s", "patches": int(len(res)),
                                 "workload": "predictor.predict_patches on a synthetic 100^3 volume (patch_size=%d res_increase=%d batch=%d, fp32): "
                                             "forward + gather, host<->device copies included; stitching excluded" % (P, R, B)}
        del net, res
    except Exception as e:
        sec["cfg5_predictor"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    return sec


def guarded(fn, line, rank, seconds):
    """Runs fn() under a watchdog.  If it has not returned after `seconds`, rank 0 prints the headline `line` with the failure
    recorded in `secondary` and every rank leaves the process (os._exit: a rank stuck inside a collective cannot unwind)."""
    import threading
    done = threading.Event()

    def fire():
        if done.is_set():
            return
        if rank == 0 and line is not None:
            line["secondary"] = {"error": "the N > 1 secondary legs did not finish within %.0f s; headline unaffected (measured before)" % seconds}
            print(json.dumps(line), flush=True)
        os._exit(0)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    try:
        return fn()
    finally:
        done.set()
        t.cancel()


def secondary_runs_dp(trainer, parallel, device, P, R, B, LB, HB, rank, world):
    """N > 1: the two secondary legs that have a data-parallel form, run by ALL ranks (collectives inside), barrier + synchronize on
    both sides, max over ranks -- same timing rule as the headline.
      cfg5_predictor  : predictor.predict_patches on the synthetic 100^3 volume, the patch list sharded over the ranks, rows gathered
                        to rank 0 (gather included, stitching excluded); value = patches of the whole job per second
      cfg2_loader_fed : the cfg2 train step fed by the on-device loader, every rank drawing its shard of each global batch."""
    sec = {}
    try:
        predictor = importlib.import_module("4dflownet_amd.predictor")
        tiler = importlib.import_module("4dflownet_amd.tiler")

        class _Vol:
            pass
        rng = np.random.default_rng(0)
        vol = _Vol()
        for n_ in ("u", "v", "w"):
            setattr(vol, n_, rng.uniform(-1, 1, (100, 100, 100)).astype(np.float32))
        for n_ in ("mag_u", "mag_v", "mag_w"):
            setattr(vol, n_, rng.uniform(0, 0.016, (100, 100, 100)).astype(np.float32))
        net = predictor.prepare_network(P, R, LB, HB, device=device)
        vel, mag = tiler.PatchGenerator(P, R).patchify(vol)
        n_patch = len(vel[0])
        predictor.predict_patches(net, vel, mag, B)                     # warm-up (allocations, staging buffers, RCCL channels)
        parallel.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        predictor.predict_patches(net, vel, mag, B)
        torch.cuda.synchronize(); parallel.barrier()
        dt = parallel.allreduce_sum_host([time.perf_counter() - t0], op="max")[0]
        sec["cfg5_predictor"] = {"value": n_patch / dt, "unit": "patches/s", "patches": int(n_patch), "ranks": world,
                                 "patches_per_rank": [b - a for a, b in zip(predictor.shard_bounds(n_patch, world)[:-1],
                                                                             predictor.shard_bounds(n_patch, world)[1:])],
                                 "workload": "predictor.predict_patches on a synthetic 100^3 volume (patch_size=%d res_increase=%d batch=%d, "
                                             "fp32): contiguous shards of the patch list, pipelined forward per rank, rows gathered to rank 0; "
                                             "host<->device copies and the gather included, stitching excluded" % (P, R, B)}
        del net
    except Exception as e:
        sec["cfg5_predictor"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    try:
        data_device = importlib.import_module("4dflownet_amd.data_device")
        patch_index = importlib.import_module("4dflownet_amd.patch_index")
        data = importlib.import_module("4dflownet_amd.data")
        ddir = os.path.join(ROOT, "tests", "golden", "data")
        import contextlib, io, tempfile
        with tempfile.TemporaryDirectory() as td:
            csv_path = os.path.join(td, "bench%d_%d.csv" % (P, rank))
            with contextlib.redirect_stdout(io.StringIO()):
                patch_index.generate_patch_index(ddir, "example_data.h5", "example_data_HR.h5", csv_path, patch_size=P, n_patch=8 * B * world,
                                                 minimum_coverage=0.05, seed=0)          # same seed on every rank: the same rows
                rows = data.load_indexes(csv_path)
                ph = data_device.DevicePatchHandler3D(ddir, P, R, B, 0.6, device=device)
                ds = ph.initialize_dataset(rows, shuffle=True, shard=(rank, world))
        tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB, hi_resblock=HB,
                                       device=device, seed=0)
        n_steps = [0]

        def epoch():
            for batch in ds:
                tc.train_step(batch)                               # ragged tails included: an empty shard still joins the all-reduce
                n_steps[0] += 1
        epoch()
        parallel.barrier(); torch.cuda.synchronize()
        n_steps[0] = 0
        t0 = time.perf_counter()
        epoch()
        torch.cuda.synchronize()
        mine = [0.0] * world
        mine[rank] = (time.perf_counter() - t0) / max(n_steps[0], 1) * 1e3
        parallel.barrier()
        dt = parallel.allreduce_sum_host([time.perf_counter() - t0], op="max")[0]
        sec["cfg2_loader_fed"] = {"value": len(rows) / dt, "unit": "patches/s", "ms_per_step": dt / n_steps[0] * 1e3, "steps": n_steps[0],
                                  "per_rank_ms_per_step": parallel.allreduce_sum_host(mine), "ranks": world,
                                  "workload": "cfg2 train_step fed by DevicePatchHandler3D on every rank (example_data*.h5 resident in HBM, %d rows "
                                              "with random rotations, shuffle, each global batch of %d split over the ranks, gradient all-reduce)"
                                              % (len(rows), B * world)}
        del tc, ds, ph
    except Exception as e:
        sec["cfg2_loader_fed"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    return sec


def self_spawn(args):
    """--gpus N > 1 without a torch.distributed.run environment: launch N ranks of this script."""
    ngpu = torch.cuda.device_count()
    if ngpu < args.gpus and not args.oversubscribe:
        raise SystemExit("bench.py: --gpus %d requested but only %d GPU(s) are visible; refusing to run fewer ranks silently "
                         "(use --oversubscribe only to self-test the launcher)" % (args.gpus, ngpu))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    # This pool's host driver only supports dmabuf IPC (image documentation: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's buffer
    # registration across processes fails with `hipIpcGetMemHandle: invalid argument`).  The image exports it already; setdefault
    # keeps a caller's explicit choice and only covers an environment that was built without it.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
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
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--event-steps", type=int, default=5,
                    help="steps of the per-kernel pass behind the timed region: every 64->64 launch bracketed by HIP events, all launches on ONE stream")
    ap.add_argument("--sustained-steps", type=int, default=300, help="length of the secondary `sustained` run (N=1)")
    ap.add_argument("--single-allreduce", action="store_true",
                    help="N>1: ONE all-reduce of the whole gradient buffer after backward instead of the three buckets started inside it")
    ap.add_argument("--pmc", dest="pmc", action="store_true", default=None,
                    help="N=1: measure roofline.traffic in this run (two extra rocprofv3 --pmc passes of 2 steps) instead of quoting profiles/ "
                         "(default: on for the full default run when rocprofv3 is on the box, off with --no-secondary)")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="launcher self-test: allow more ranks than GPUs (ranks share devices, gloo with host staging); not a scaling number")
    args = ap.parse_args()
    if args.config == "cfg4":
        args.patch, args.res, args.batch, args.dtype = 32, 4, 4, "bf16"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)

    parallel = importlib.import_module("4dflownet_amd.parallel")
    ngpu = torch.cuda.device_count()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, env_world))
    oversub = env_world > ngpu
    if oversub and not args.oversubscribe:
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible" % (env_world, ngpu))
    if oversub:
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(ngpu, 1))
    rank, world, local_rank = parallel.init_from_env(backend="gloo" if oversub else None)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # the world size the collective actually sees: all-reduce a one per rank
    collective_ranks = int(round(parallel.allreduce_sum_(torch.ones(1, device=device)).item()))
    backend = "none" if world == 1 else ("gloo" if oversub else "rccl")

    # never benchmark a binary older than its sources: (re)build under the build lock (one rank compiles, the others wait), then
    # load -- load() itself re-checks the stamp and fails loudly if the HIP library is missing or stale
    build = importlib.import_module("4dflownet_amd.build")
    build.build_library()
    parallel.barrier()
    fdn = importlib.import_module("4dflownet_amd")
    trainer = importlib.import_module("4dflownet_amd.trainer")
    fdn._lib.load()
    P, R, B, LB, HB = args.patch, args.res, args.batch, args.low, args.hi
    bf16 = args.dtype == "bf16"
    tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=LB,
                                   hi_resblock=HB, device=device, seed=0, dtype="bfloat16" if bf16 else "float32",
                                   bucketed_allreduce=not args.single_allreduce)
    tc.profile_allreduce = world > 1
    batch = synthetic_batch(B, P, R, 1234 + rank, device)
    timer = LaunchTimer(tc.model.ops)
    timer.install()

    for _ in range(args.warmup):
        tc.train_step(batch)
    tc.allreduce_wait_events.clear()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tc.train_step(batch)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0               # this rank's own work is done (before the closing barrier)
    parallel.barrier()
    dt = time.perf_counter() - t0
    # The per-kernel pass, in steps of its own BEHIND the timed region: the product step runs the weight gradients on a second
    # stream beside the dgrad chain (network.overlap_wgrad), and an event pair around a launch that shares the chip with another
    # stream's kernel times neither.  Here everything is issued on one stream and every 64->64 launch is bracketed by HIP events
    # on it; same kernels, same operands, same order within each stream as the timed steps.
    per_kernel_ms = event_pass(tc, batch, timer, args.event_steps)
    timer.uninstall()

    dt = parallel.allreduce_sum_host([dt], op="max")[0]
    # N>1 diagnostics: every rank's own time to finish its K steps, and the EXPOSED gradient all-reduce time per step (HIP events
    # around the stream wait in front of Adam; the rest of the collective ran under backward)
    mine = [0.0] * world
    mine[rank] = dt_local / args.steps * 1e3
    per_rank_ms = parallel.allreduce_sum_host(mine)
    waits = [e0.elapsed_time(e1) for e0, e1 in tc.allreduce_wait_events]
    mine[rank] = float(np.mean(waits)) if waits else 0.0
    per_rank_wait = parallel.allreduce_sum_host(mine)
    # rank 0 assembles the headline object BEFORE the N > 1 secondary legs run: those legs use point-to-point RCCL traffic that no
    # multi-GPU box has exercised yet, and a watchdog prints the headline (with the failure noted) rather than lose it to a hang
    line = None
    if args.pmc is None:                               # the driver-facing default run measures its own traffic figure where the line is made
        args.pmc = not args.no_secondary and world == 1
    if rank == 0:
        fwd_flop = fwd_flop_per_patch(tc.model.specs, P, R, LB)
        tr = CFG4_TRAFFIC if args.config == "cfg4" else ([] if bf16 else CFG2_TRAFFIC)
        line = {
            "metric": "3D patches/sec (train step, patch=%d, res×%d)" % (P, R),
            "value": args.steps * B * world / dt,
            "unit": "patches/s",
            # n_gpus = distinct devices the job ran on; collective_ranks = ranks the gradient all-reduce saw, over `backend`;
            # rccl_ranks counts RCCL peers only: a gloo run (the --oversubscribe launcher self-test) reports 0
            "n_gpus": min(collective_ranks, ngpu) if oversub else collective_ranks,
            "collective_ranks": collective_ranks,
            "backend": backend,
            "rccl_ranks": 0 if backend == "gloo" else collective_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_one_stream": per_kernel_ms,          # the per-kernel pass (event records included): what the overlap buys is the difference
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32",
            "products": products_note(tc.model, bf16),
            "data": "synthetic (SURVEY 8d: default_rng(1234+rank) inputs, Glorot-uniform default_rng(0) weights)",
            "config": {"workload": "%s train_step: patch_size=%d res_increase=%d batch=%d/GPU low_resblock=%d hi_resblock=%d %s"
                                   % (args.config, P, R, B, LB, HB, "bf16 activations, fp32 accumulation/parameters" if bf16 else "fp32"),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "collective": ("none" if world == 1 else
                                      "gloo, host-staged (OVERSUBSCRIBED launcher self-test: %d ranks on %d GPU -- not a scaling number)" % (world, ngpu)
                                      if oversub else "RCCL sum all-reduce of the flat fp32 gradient (%d B) per step, %s"
                                      % (4 * (tc.model.n_params + 1), "3 buckets started inside backward" if tc.bucketed_allreduce
                                         else "one call after backward"))},
            "roofline": roofline_obj(timer, "conv", bf16, "%s (3x3x3 64->64 forward + fused-dgrad%s launches%s)"
                                     % (("conv64_bf16_kernel", "", "") if bf16 else
                                        ("conv64_wino2d", "", ": conv64_wino2d_kernel forward, conv64_wino2d_shell_kernel fused dgrad = inner box + shell faces in one "
                                         "launch; 2-D Winograd F(4,3) along H x F(4,3) along W, the faces F(4,3) along W")), tr),
            "roofline_wgrad": roofline_obj(timer, "wgrad", bf16, "%s (3x3x3 64->64 weight gradient + partial reduction%s)"
                                           % (("wgrad64_bf16_dma_kernel", "") if bf16 else ("wgrad64_wino_kernel", "; Winograd F(3,2) along D x F(3,4) along W")), tr),
            "train_step_tflops": args.steps * B * world / dt * 3.0 * fwd_flop / 1e12,
            "lib_source_stamp": build.source_stamp()[:16],          # sha256 prefix of csrc/ + include/fdn.h + flags the binary was built from
        }
        if world > 1:
            line["per_rank_ms_per_step"] = per_rank_ms
            line["allreduce_exposed_ms"] = {"per_rank_mean": per_rank_wait, "max": max(per_rank_wait),
                                            "how": "HIP events on the compute stream around allreduce_wait (trainer.train_step), mean over the timed steps"}
        if oversub:
            line["oversubscribed"] = True
    sec_dp = None
    if world > 1 and not bf16 and args.config == "cfg2" and not args.no_secondary:
        del tc, batch
        torch.cuda.empty_cache()
        sec_dp = guarded(lambda: secondary_runs_dp(trainer, parallel, device, P, R, B, LB, HB, rank, world), line, rank,
                         float(os.environ.get("FDN_BENCH_SECONDARY_TIMEOUT", "300")))
        tc = batch = None
    if rank != 0:
        if parallel.is_dist():
            parallel.barrier()
        return
    if sec_dp is not None:
        line["secondary"] = sec_dp
    del tc, batch
    torch.cuda.empty_cache()
    if world == 1 and not bf16 and args.config == "cfg2":
        if not args.no_secondary:
            try:                                    # secondary legs run after the headline was measured: a failing leg is recorded, not fatal
                line["secondary"] = secondary_runs(trainer, parallel, device, P, R, B, LB, HB, args.sustained_steps)
            except Exception as e:
                line["secondary"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
        if not args.no_cpu_baseline:
            try:                                    # a reported baseline, measured AFTER the headline: it can fail, the line still goes out
                line["cpu_baseline"] = cpu_baseline(P, R, LB, HB)
            except Exception as e:
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:]), "kind": "port", "value": None, "unit": "patches/s"}
    if args.pmc and world == 1:
        # roofline.traffic measured by THIS run: two rocprofv3 --pmc passes of the same command (2 steps each), behind every timed leg (a
        # profiler child that has just left the device cost the next leg one 2.5-s step when the passes ran in front of them)
        global PMC_LIVE
        tail = ["--config", args.config] if args.config != "cfg2" else []
        try:
            PMC_LIVE = pmc_live_passes(tail)
        except Exception as e:                          # (a box without working counters must not cost the line)
            print("bench.py --pmc: %s: %s" % (type(e).__name__, str(e)[-200:]), file=sys.stderr)
            PMC_LIVE = None
        if PMC_LIVE is not None:
            tr = CFG4_TRAFFIC if args.config == "cfg4" else ([] if bf16 else CFG2_TRAFFIC)
            for key in ("roofline", "roofline_wgrad"):
                obj = line.get(key)
                if obj:
                    traffic, tfile = pmc_traffic_bytes(tr, obj["kernel"].split(" ")[0])
                    if tfile:
                        obj["traffic"], obj["traffic_source"] = traffic, tfile
                        obj["traffic_unit"] = "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)"
    print(json.dumps(line), flush=True)
    if parallel.is_dist():
        parallel.barrier()


if __name__ == "__main__":
    main()
