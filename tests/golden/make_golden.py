"""Generates tests/golden/reference_golden.json by IMPORTING the reference's own TF-free modules
(/root/reference/src: Network.PatchHandler3D, Network.PatchGenerator, utils.ImageDataset) and running them on
the reference's shipped data files.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden.py

(h5py lives in that interpreter; TensorFlow is absent everywhere, so a stub module exposing only
`tensorflow.newaxis = None` -- the single attribute PatchHandler3D.load_patches_from_index_file touches,
PatchHandler3D.py:78-81 -- is placed in sys.modules.  No reference source is copied: only inputs (the data
files, copied verbatim to tests/golden/data/) and outputs (shapes, float64 sums, sha1 of the float32 bytes,
sampled voxels) are stored.)"""
import hashlib
import json
import os
import shutil
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "src"))
tf_stub = types.ModuleType("tensorflow")
tf_stub.newaxis = None
sys.modules["tensorflow"] = tf_stub

from Network.PatchHandler3D import PatchHandler3D          # noqa: E402
from Network.PatchGenerator import PatchGenerator          # noqa: E402
from utils.ImageDataset import ImageDataset                # noqa: E402
import h5py                                                # noqa: E402


def digest(a):
    a = np.ascontiguousarray(np.asarray(a))
    flat = a.reshape(-1)
    idx = np.linspace(0, flat.size - 1, num=min(8, flat.size)).astype(np.int64)
    return {"shape": list(a.shape), "dtype": str(a.dtype), "sum": float(flat.astype(np.float64).sum()),
            "sumsq": float((flat.astype(np.float64) ** 2).sum()), "sha1": hashlib.sha1(a.tobytes()).hexdigest(),
            "samples": [float(flat[i]) for i in idx]}


class Cell:
    """Stands in for one element of the tf string tensor row (PatchHandler3D.py:52-59)."""

    def __init__(self, s):
        self.s = s

    def numpy(self):
        return self.s.encode()

    def __int__(self):
        return int(self.s)


def main():
    data_dir = os.path.join(REF, "data")
    out = {"generated_by": "tests/golden/make_golden.py importing /root/reference/src modules"}
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    for f in ("example_data.h5", "example_data_HR.h5", "train.csv", "validate.csv", "benchmark.csv"):
        shutil.copyfile(os.path.join(data_dir, f), os.path.join(HERE, "data", f))

    # ---- raw HDF5 content (pins the built-in HDF5 reader) ----
    raw = {}
    for f in ("example_data.h5", "example_data_HR.h5"):
        with h5py.File(os.path.join(data_dir, f), "r") as hl:
            raw[f] = dict((k, digest(hl[k][...])) for k in hl.keys())
    out["hdf5"] = raw

    # ---- CSV parse (trainer.py:5-10) ----
    idx = np.genfromtxt(os.path.join(data_dir, "train.csv"), delimiter=",", skip_header=True, dtype="unicode")
    out["csv"] = {"shape": list(idx.shape), "dtype": str(idx.dtype), "row0": list(idx[0]), "row49": list(idx[49])}

    # ---- loader (PatchHandler3D.load_patches_from_index_file) ----
    names = ["u", "v", "w", "u_mag", "v_mag", "w_mag", "u_hr", "v_hr", "w_hr", "venc", "mask"]
    loader = []
    seen = set()
    rows = []
    for r in range(idx.shape[0]):                 # un-rotated + every (plane, k) combination present
        key = (idx[r][6], idx[r][7], idx[r][8])
        if key not in seen:
            seen.add(key)
            rows.append(r)
    rows += [49]
    for (P, R, thr) in ((16, 2, 0.6), (16, 1, 0.6), (12, 2, 0.3)):
        ph = PatchHandler3D(data_dir, P, R, 4, thr)
        for r in rows:
            res = ph.load_patches_from_index_file([Cell(c) for c in idx[r]])
            loader.append({"patch_size": P, "res_increase": R, "mask_threshold": thr, "row": int(r),
                           "csv_row": list(idx[r]), "outputs": dict((n, digest(a)) for n, a in zip(names, res))})
    out["loader"] = loader

    # ---- inference volume reader + tiler (ImageDataset, PatchGenerator) ----
    ds = ImageDataset()
    fpath = os.path.join(data_dir, "example_data.h5")
    out["image_dataset"] = {"len": int(ds.get_dataset_len(fpath))}
    ds.load_vectorfield(fpath, 0)
    out["image_dataset"].update({k: digest(getattr(ds, k)) for k in ("u", "v", "w", "mag_u", "mag_v", "mag_w")})
    out["image_dataset"]["venc"] = float(ds.venc)
    out["image_dataset"]["velocity_per_px"] = float(ds.velocity_per_px)
    out["image_dataset"]["dx"] = [float(x) for x in ds.dx]
    tiler = []
    for (P, R) in ((24, 2), (16, 1), (16, 2), (32, 4), (12, 3)):
        pg = PatchGenerator(P, R)
        vel, mag = pg.patchify(ds)
        entry = {"patch_size": P, "res_increase": R, "n_patches": int(len(vel[0])), "nr": [pg.nr_x, pg.nr_y, pg.nr_z],
                 "padding": list(pg.padding), "u_stacks": digest(vel[0]), "w_stacks": digest(vel[2]),
                 "wmag_stacks": digest(mag[2])}
        # stitch a deterministic fake network output: nearest-neighbour repeat of each LR patch (x R), 3 components
        res = np.stack([np.repeat(np.repeat(np.repeat(v[..., 0], R, 1), R, 2), R, 3) for v in vel], axis=-1)
        pu, pv, pw = pg.unpatchify(res)
        entry["stitched_u"] = digest(pu)
        entry["stitched_w"] = digest(pw)
        tiler.append(entry)
    out["tiler"] = tiler

    # ---- predictor post-processing (predictor.py:103-107, ImageDataset.py:31) on the stitched (24,2) volume ----
    pg = PatchGenerator(24, 2)
    vel, mag = pg.patchify(ds)
    res = np.stack([np.repeat(np.repeat(np.repeat(v[..., 0], 2, 1), 2, 2), 2, 3) for v in vel], axis=-1).astype(np.float64)
    v = pg._patchup_with_overlap(res[:, :, :, :, 0], pg.nr_x, pg.nr_y, pg.nr_z)
    v = v * ds.venc
    v[np.abs(v) < ds.velocity_per_px] = 0
    out["postprocess_u"] = digest(np.array(np.expand_dims(v, 0), dtype="float32"))

    with open(os.path.join(HERE, "reference_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "reference_golden.json"), "rows:", rows)


if __name__ == "__main__":
    main()
