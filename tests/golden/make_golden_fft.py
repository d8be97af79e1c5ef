"""Generates tests/golden/fft_golden.json by IMPORTING the reference's prepare_data/fft_downsampling.py (numpy only) and
running it on seeded synthetic volumes.  Run in the build container only:  python tests/golden/make_golden_fft.py
Only inputs (seeds, shapes) and outputs (shape, sums, sampled voxels) are stored -- no reference source."""
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src/prepare_data")
import fft_downsampling as ref          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def digest(a):
    a = np.asarray(a)
    flat = a.reshape(-1)
    idx = np.linspace(0, flat.size - 1, num=min(8, flat.size)).astype(np.int64)
    d = {"shape": list(a.shape), "dtype": str(a.dtype)}
    if np.iscomplexobj(a):
        d.update(sum_re=float(flat.real.sum()), sum_im=float(flat.imag.sum()), sumsq=float((np.abs(flat) ** 2).sum()),
                 samples=[[float(flat[i].real), float(flat[i].imag)] for i in idx])
    else:
        d.update(sum=float(flat.sum()), sumsq=float((flat.astype(np.float64) ** 2).sum()), samples=[float(flat[i]) for i in idx])
    return d


def volume(seed, shape):
    rng = np.random.default_rng(seed)
    vel = rng.uniform(-1.2, 1.2, shape)
    mask = (rng.uniform(size=shape) < 0.4).astype(np.float64)
    return vel, mask * 120.0


def main():
    out = {"generated_by": "tests/golden/make_golden_fft.py importing /root/reference/src/prepare_data/fft_downsampling.py",
           "cases": []}
    for seed, shape, ratio, venc, snr in ((1, (16, 12, 20), 0.5, 1.5, 15.3), (2, (24, 24, 24), 0.5, 2.0, 16.9),
                                          (3, (20, 16, 12), 0.25, 0.6, 14.0)):
        vel, mag = volume(seed, shape)
        f = np.fft.fftn(mag * np.exp(1j * vel))
        crop = ref.rectangular_crop3d(f, ratio)
        np.random.seed(100 + seed)
        lr_v, lr_m = ref.downsample_phase_img(vel, mag, venc, ratio, snr)
        out["cases"].append({"seed": seed, "shape": list(shape), "crop_ratio": ratio, "venc": venc, "snr_db": snr,
                             "np_seed": 100 + seed, "crop": digest(crop), "lr_velocity": digest(lr_v), "lr_mag": digest(lr_m),
                             "next_random": float(np.random.normal())})
    json.dump(out, open(os.path.join(HERE, "fft_golden.json"), "w"), indent=1)
    print("wrote fft_golden.json with", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
