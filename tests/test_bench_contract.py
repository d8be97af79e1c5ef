"""The JSON contract of bench.py's roofline objects, checked on the CPU with a stand-in for the HIP-event launch timer
(`frac` = FLOPs executed on the matrix pipe / peak <= 1, the algorithmic ratio under its own keys)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _FakeTimer:
    """summary(kind) -> (launches, mean ms, mean algorithmic FLOP per launch, mean executed FLOP per launch)"""

    def __init__(self, rows):
        self.rows = rows

    def summary(self, kind):
        return self.rows.get(kind)


def test_roofline_object_keys_and_semantics():
    b = _bench()
    vox = 8 * 48 ** 3
    alg = vox * b.FLOP_PER_VOXEL_CONV64
    t = _FakeTimer({"conv": (300, 0.75, alg, 0.5 * alg), "wgrad": (150, 0.80, alg, 0.5 * alg)})
    r = b.roofline_obj(t, "conv", False, "conv64_wino_kernel (test)", b.CFG2_TRAFFIC)
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "launches_timed",
              "avg_launch_ms", "executed_gflop_per_launch", "algorithmic_gflop_per_launch", "algorithmic_achieved", "algorithmic_frac",
              "algorithmic_speedup", "note"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == b.PEAK_FP32_MFMA_TFLOPS
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] <= 1.0          # executed work can never beat the pipe
    assert abs(r["achieved"] - 0.5 * alg / 0.75e-3 / 1e12) < 1e-6
    assert abs(r["algorithmic_frac"] - 2 * r["frac"]) < 1e-9 and abs(r["algorithmic_speedup"] - 2.0) < 1e-12
    assert r["algorithmic_bytes_per_launch"] == vox * 64 * 4 * 2 + 27 * 64 * 64 * 4
    # traffic comes from the newest committed PMC summary that has the kernel (or from this run's own passes with --pmc); the source and
    # whether it was measured on the library being benchmarked are named in traffic_source
    assert r["traffic"] is None or ("profiles/" in r["traffic_source"] and r["traffic"] > 0 and
                                    ("measured on this library" in r["traffic_source"] or "STALE" in r["traffic_source"] or "unstamped" in r["traffic_source"]))
    # bf16 kernels are direct convolutions: executed == algorithmic, priced against the bf16 peak
    rb = b.roofline_obj(_FakeTimer({"conv": (10, 1.6, 4 * 128 ** 3 * b.FLOP_PER_VOXEL_CONV64, 0.0)}), "conv", True, "conv64_bf16_kernel (test)", b.CFG4_TRAFFIC)
    assert rb["peak"] == b.PEAK_BF16_MFMA_TFLOPS and abs(rb["algorithmic_speedup"] - 1.0) < 1e-12 and rb["frac"] == rb["algorithmic_frac"] < 1
    assert b.roofline_obj(_FakeTimer({}), "conv", False, "none", []) is None
    json.dumps(r), json.dumps(rb)                           # both serialise


def test_pmc_traffic_summaries_are_readable_and_skip_metadata(tmp_path):
    b = _bench()
    v, f = b.pmc_traffic_bytes(["does_not_exist.json"], "conv64_wino_kernel")
    assert v is None and f is None
    # every committed summary parses; "_meta" (commit / source stamp) is not mistaken for a kernel row
    for name in b.CFG2_TRAFFIC + b.CFG4_TRAFFIC:
        if os.path.exists(os.path.join(ROOT, "profiles", name)):
            kern = "conv64_bf16_kernel" if "cfg4" in name else ("conv64_mfma_kernel" if name.startswith("r1_") else
                                                                    ("conv64_wino_kernel" if name[:2] in ("r2", "r3") else "conv64_wino2d"))
            v, f = b.pmc_traffic_bytes([name], kern)
            assert f.startswith("profiles/" + name + " (") and v > 1e6, (name, v, f)


def test_bench_frac_agrees_with_the_pmc_busy_counter():
    """roofline.frac of the committed bench line (executed FLOPs from the launch shapes / HIP-event time / peak) against the hardware's
    own count, SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), from the PMC pass of the same round
    (profiles/r*_pmc_sq.txt, tools/profile_round.sh): within 0.02 for the conv kernel.  The wgrad bracket of bench.py also contains the
    12-us partial-sum reduction (no MFMA), so its frac sits below the kernel's own busy ratio: within 0.04."""
    import glob
    import re
    import pytest
    rounds = sorted(int(m.group(1)) for m in (re.fullmatch(r"r(\d+)_pmc_sq\.txt", os.path.basename(f))
                                              for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.txt"))) if m)
    rounds = [r for r in rounds if r >= 4 and os.path.exists(os.path.join(ROOT, "profiles", "r%d_bench_line_nosecondary.json" % r))]
    if not rounds:
        pytest.skip("no round >= 4 profile collected yet")
    r = rounds[-1]
    line = json.load(open(os.path.join(ROOT, "profiles", "r%d_bench_line_nosecondary.json" % r)))
    conv_k = line["roofline"]["kernel"].split(" ")[0]
    wg_k = line["roofline_wgrad"]["kernel"].split(" ")[0]
    acc = {}
    for l in open(os.path.join(ROOT, "profiles", "r%d_pmc_sq.txt" % r)):
        m = re.search(r"::(\w+).*GRBM_GUI_ACTIVE=([0-9.e+]+).*SQ_VALU_MFMA_BUSY_CYCLES=([0-9.e+]+)", l)
        if m:
            for key in (conv_k, wg_k):                       # pooled over the kernels the label names (forward + fused-dgrad launch)
                if m.group(1).startswith(key):
                    a = acc.setdefault(key, [0.0, 0.0])
                    a[0] += float(m.group(3)); a[1] += float(m.group(2))
    busy = {k: v[0] / (1024.0 * v[1] / 8.0) for k, v in acc.items()}
    assert conv_k in busy and wg_k in busy, (conv_k, wg_k, sorted(busy))
    assert abs(line["roofline"]["frac"] - busy[conv_k]) <= 0.02, (line["roofline"]["frac"], busy[conv_k])
    assert -0.005 <= busy[wg_k] - line["roofline_wgrad"]["frac"] <= 0.04, (line["roofline_wgrad"]["frac"], busy[wg_k])


def test_executed_flop_pricing_follows_the_kernel_the_planner_picks():
    """bench.py prices `achieved` / `frac` with the multiplies the selected kernel EXECUTES (include/fdn.h, FDN_ALGO_*): 6.75 of 27
    tap-equivalents per voxel for the 2-D Winograd conv kernel with F(4,3) along H (H % 4 == 0, W % 4 == 0), 9 with F(2,3) along H
    (H even, or FDN_ALGO_WINO_H2), 13.5 for W-only Winograd, 27 direct; the shell launch of a fused dgrad: 4.5 per d/h-face position,
    9 per w-face position."""
    b = _bench()
    per_tap = 2.0 * 64 * 64
    vox = 8 * 48 ** 3
    assert b.executed_conv64_flop(8, 48, 48, 48) == vox * 6.75 * per_tap
    assert b.executed_conv64_flop(8, 48, 48, 48, algo=3) == vox * 9 * per_tap             # FDN_ALGO_WINO_H2
    assert b.executed_conv64_flop(1, 5, 10, 12) == 5 * 10 * 12 * 9 * per_tap                # H even, not a multiple of 4
    assert b.executed_conv64_flop(8, 48, 48, 48, algo=2) == vox * 13.5 * per_tap          # FDN_ALGO_WINO_W
    assert b.executed_conv64_flop(8, 48, 48, 48, algo=1) == vox * 27 * per_tap            # FDN_ALGO_DIRECT
    assert b.executed_conv64_flop(1, 5, 7, 12) == 5 * 7 * 12 * 13.5 * per_tap               # odd H: 1-D kernel
    assert b.executed_conv64_flop(1, 5, 7, 9) == 5 * 7 * 9 * 27 * per_tap                   # W % 4 != 0: direct
    assert abs(b.executed_conv64_flop(8, 48, 48, 48) / (vox * b.FLOP_PER_VOXEL_CONV64) - 0.25) < 1e-12
    D = H = W = 24
    shell = b.executed_shell_flop(8, D, H, W)
    assert shell == 8 * ((2 * (H + 2) * W + 2 * D * W) * 4.5 + 2 * (D + 2) * (H + 2) * 9.0) * per_tap
    # the shell against the inner box's executed work: 24 % at 24^3, 12 % at 48^3
    assert 0.22 < shell / b.executed_conv64_flop(8, D, H, W) < 0.26
    assert 0.11 < b.executed_shell_flop(8, 48, 48, 48) / b.executed_conv64_flop(8, 48, 48, 48) < 0.13


def test_secondary_watchdog_prints_the_headline_and_leaves():
    """bench.guarded(): the N > 1 secondary legs run after the headline has been measured; if they hang (point-to-point RCCL traffic
    no multi-GPU box has exercised), rank 0 still prints the ONE JSON line -- with the failure recorded -- and exits 0."""
    import subprocess
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "bench.guarded(lambda: time.sleep(30), {'metric': 'm', 'value': 1.0}, 0, 0.5)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.0 and "did not finish" in line["secondary"]["error"]
    # and a leg that returns in time is passed through untouched
    code = ("import sys; sys.path.insert(0, %r); import bench; print(bench.guarded(lambda: {'ok': 1}, {}, 0, 30))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("{'ok': 1}")
