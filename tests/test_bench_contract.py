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
    # traffic comes from the newest committed PMC summary that has the kernel; the file is named in traffic_unit
    assert r["traffic"] is None or ("profiles/" in r["traffic_unit"] and r["traffic"] > 0)
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
            kern = "conv64_bf16_kernel" if "cfg4" in name else ("conv64_wino_kernel" if not name.startswith("r1_") else "conv64_mfma_kernel")
            v, f = b.pmc_traffic_bytes([name], kern)
            assert f == name and v > 1e6, (name, v)
