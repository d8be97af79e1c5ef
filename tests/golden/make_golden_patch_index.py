"""Generates tests/golden/patch_index_golden.json by IMPORTING the reference's own patch-index generator
(/root/reference/src/prepare_data/PatchData.py -- numpy + `random` only) and running it, seeded, on the reference's
example mask exactly as prepare_patches.py:14-47 drives it.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden_patch_index.py

Only inputs (parameters, seed) and outputs (the CSV text the reference wrote) are stored -- no reference source."""
import contextlib
import io
import json
import os
import random
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "src", "prepare_data"))
import PatchData as pd            # noqa: E402
import h5py                       # noqa: E402

CASES = [
    # the script's defaults (prepare_patches.py:15-20), then the knobs one at a time
    dict(patch_size=16, n_patch=10, n_empty_patch_allowed=0, all_rotation=False, mask_threshold=0.4, minimum_coverage=0.2, seed=0),
    dict(patch_size=16, n_patch=3, n_empty_patch_allowed=0, all_rotation=True, mask_threshold=0.4, minimum_coverage=0.2, seed=1),
    dict(patch_size=24, n_patch=6, n_empty_patch_allowed=0, all_rotation=False, mask_threshold=0.4, minimum_coverage=0.2, seed=5),
    dict(patch_size=24, n_patch=4, n_empty_patch_allowed=2, all_rotation=False, mask_threshold=0.6, minimum_coverage=0.3, seed=7),
    dict(patch_size=32, n_patch=3, n_empty_patch_allowed=0, all_rotation=False, mask_threshold=0.4, minimum_coverage=0.1, seed=2),
    # the not_found > 100 give-up branch (PatchData.py:19-21) with and without the empty-patch budget
    dict(patch_size=16, n_patch=2, n_empty_patch_allowed=0, all_rotation=False, mask_threshold=0.4, minimum_coverage=1.1, seed=3),
    dict(patch_size=16, n_patch=2, n_empty_patch_allowed=1, all_rotation=False, mask_threshold=0.4, minimum_coverage=1.1, seed=3),
    dict(patch_size=12, n_patch=5, n_empty_patch_allowed=1, all_rotation=True, mask_threshold=0.5, minimum_coverage=0.45, seed=11),
]


def main():
    data_dir = os.path.join(REF, "data")
    lr_file, hr_file = "example_data.h5", "example_data_HR.h5"
    with h5py.File(os.path.join(data_dir, lr_file), "r") as f:
        n_frames = len(f["u"])
        mask = np.asarray(f["mask"][0])
    out = {"generated_by": "tests/golden/make_golden_patch_index.py importing /root/reference/src/prepare_data/PatchData.py",
           "lr_file": lr_file, "hr_file": hr_file, "n_frames": int(n_frames), "cases": []}
    for case in CASES:
        binary_mask = (mask >= case["mask_threshold"]) * 1
        random.seed(case["seed"])
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "index.csv")
            pd.write_header(path)
            with contextlib.redirect_stdout(io.StringIO()):
                for index in range(n_frames):
                    pd.generate_random_patches(lr_file, hr_file, path, index, case["n_patch"], binary_mask, case["patch_size"],
                                               case["minimum_coverage"], case["n_empty_patch_allowed"], case["all_rotation"])
            text = open(path, newline="").read()
        # direct coverage vectors (PatchData.calculate_patch_coverage, :97-102) at fixed starts
        cov = []
        for start in ((0, 0, 0), (5, 7, 3), (13, 11, 9), (18, 14, 12)):
            if all(s + case["patch_size"] <= n for s, n in zip(start, binary_mask.shape)):
                p = pd.PatchData(lr_file, hr_file, case["patch_size"])
                p.set_patch(0, *start)
                p.calculate_patch_coverage(binary_mask, case["minimum_coverage"])
                cov.append({"start": list(start), "coverage": float(p.coverage)})
        out["cases"].append({"params": case, "csv": text, "coverage": cov})
    with open(os.path.join(HERE, "patch_index_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote patch_index_golden.json:", [c["csv"].count("\n") - 1 for c in out["cases"]], "rows")


if __name__ == "__main__":
    main()
