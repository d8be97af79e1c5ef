"""fft_downsampling (offline data synthesis) against vectors produced by the reference's own module
(tests/golden/make_golden_fft.py).  Same numpy -> results agree to rounding; the seeded random stream is consumed identically."""
import importlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "fft_golden.json")))
fftd = importlib.import_module("4dflownet_amd.fft_downsampling")


def volume(seed, shape):
    rng = np.random.default_rng(seed)
    vel = rng.uniform(-1.2, 1.2, shape)
    mask = (rng.uniform(size=shape) < 0.4).astype(np.float64)
    return vel, mask * 120.0


def check(a, d, rtol=1e-9):
    a = np.asarray(a)
    assert list(a.shape) == d["shape"] and str(a.dtype) == d["dtype"]
    flat = a.reshape(-1)
    idx = np.linspace(0, flat.size - 1, num=min(8, flat.size)).astype(np.int64)
    if np.iscomplexobj(a):
        np.testing.assert_allclose(flat.real.sum(), d["sum_re"], rtol=rtol, atol=1e-6)
        np.testing.assert_allclose((np.abs(flat) ** 2).sum(), d["sumsq"], rtol=rtol)
        np.testing.assert_allclose([[flat[i].real, flat[i].imag] for i in idx], d["samples"], rtol=rtol, atol=1e-9)
    else:
        np.testing.assert_allclose(flat.sum(), d["sum"], rtol=rtol, atol=1e-9)
        np.testing.assert_allclose((flat ** 2).sum(), d["sumsq"], rtol=rtol)
        np.testing.assert_allclose([flat[i] for i in idx], d["samples"], rtol=rtol, atol=1e-12)


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: "seed%d" % c["seed"])
def test_matches_reference_vectors(case):
    vel, mag = volume(case["seed"], tuple(case["shape"]))
    crop = fftd.rectangular_crop3d(np.fft.fftn(mag * np.exp(1j * vel)), case["crop_ratio"])
    check(crop, case["crop"])
    np.random.seed(case["np_seed"])
    lr_v, lr_m = fftd.downsample_phase_img(vel, mag, case["venc"], case["crop_ratio"], case["snr_db"])
    check(lr_v, case["lr_velocity"])
    check(lr_m, case["lr_mag"])
    assert np.random.normal() == case["next_random"]          # the random stream was consumed exactly like the reference


@pytest.mark.gpu
@pytest.mark.parametrize("case", G["cases"][:2], ids=lambda c: "seed%d" % c["seed"])
def test_gpu_fft_path_matches(case):
    vel, mag = volume(case["seed"], tuple(case["shape"]))
    np.random.seed(case["np_seed"])
    lr_v, lr_m = fftd.downsample_phase_img_gpu(vel, mag, case["venc"], case["crop_ratio"], case["snr_db"])
    check(lr_v, case["lr_velocity"], rtol=1e-7)
    check(lr_m, case["lr_mag"], rtol=1e-7)


def test_prepare_lowres_dataset_writes_reference_layout(tmp_path):
    """End to end on the shipped high-res example: dataset names / shapes / dtypes of prepare_lowres_dataset.py, values
    usable by the loader (venc above the row's velocity range, magnitudes scaled by the crop)."""
    import random
    h5io = importlib.import_module("4dflownet_amd.h5io")
    prep = importlib.import_module("4dflownet_amd.prepare_lowres")
    src = os.path.join(HERE, "golden", "data", "example_data_HR.h5")
    with h5io.open_read(src) as hf:
        names = set(hf.keys()) if hasattr(hf, "keys") else set()
    if not {"u_max", "v_max", "w_max"} <= names:
        pytest.skip("example_data_HR.h5 has no u_max/v_max/w_max rows")
    random.seed(3); np.random.seed(3)
    out = prep.prepare_lowres_dataset(src, str(tmp_path / "lr.h5"), downsample=2)
    d = h5io.read_all(out)
    hr = h5io.read_all(src)
    n = hr["u"].shape[0]
    for k in ("u", "v", "w", "mag_u", "mag_v", "mag_w"):
        assert d[k].shape == (n,) + tuple(s // 2 for s in hr["u"].shape[1:]) and d[k].dtype == np.float32
    for k in ("venc_u", "venc_v", "venc_w", "SNRdb"):
        assert d[k].shape == (n,)
    assert d["mask"].shape[0] == 1
    assert np.all(np.abs(d["u"]) <= d["venc_u"][:, None, None, None] + 1e-6)
