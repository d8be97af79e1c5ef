"""Patch-index generator (counterpart of prepare_patches.py / PatchData.py): schema, bounds, coverage rule, determinism,
and that the loader consumes its rows at patch sizes the shipped CSVs cannot serve (P=24)."""
import importlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")
pi = importlib.import_module("4dflownet_amd.patch_index")
data = importlib.import_module("4dflownet_amd.data")
h5io = importlib.import_module("4dflownet_amd.h5io")


def test_generate_p24_index_and_load_rows(tmp_path):
    out = str(tmp_path / "train24.csv")
    rows = pi.generate_patch_index(DATA, "example_data.h5", "example_data_HR.h5", out, patch_size=24, n_patch=6,
                                   all_rotation=False, mask_threshold=0.4, minimum_coverage=0.2, seed=5)
    assert len(rows) == 12 and [r["rotate"] for r in rows] == [0, 1] * 6
    header = open(out).readline().strip()
    assert header == ",".join(pi.FIELDNAMES) == open(os.path.join(DATA, "train.csv")).readline().strip()
    with h5io.H5File(os.path.join(DATA, "example_data.h5")) as f:
        mask = f["mask"].read()[0]
    bm = (mask >= 0.4) * 1
    for r in rows:
        assert 0 <= r["start_x"] <= 42 - 24 and 0 <= r["start_y"] <= 38 - 24 and 0 <= r["start_z"] <= 36 - 24
        cov = np.count_nonzero(bm[r["start_x"]:r["start_x"] + 24, r["start_y"]:r["start_y"] + 24, r["start_z"]:r["start_z"] + 24]) / 24 ** 3
        assert r["coverage"] == round(cov * 1000) / 1000 and r["coverage"] >= 0.2
        if r["rotate"]:
            assert r["rotation_plane"] in (1, 2, 3) and r["rotation_degree_idx"] in (1, 2, 3)
    again = pi.generate_patch_index(DATA, "example_data.h5", "example_data_HR.h5", str(tmp_path / "b.csv"), patch_size=24,
                                    n_patch=6, seed=5)
    assert again == rows
    idx = data.load_indexes(out)
    assert idx.shape == (12, 10)
    ph = data.PatchHandler3D(DATA, 24, 2, 4, 0.6)
    for b in ph.initialize_dataset(idx, shuffle=False, shard=(0, 1)):
        assert b[0].shape[1:] == (24, 24, 24, 1) and b[6].shape[1:] == (48, 48, 48, 1) and b[10].shape[1:] == (48, 48, 48)


def test_all_rotations_and_coverage_budget(tmp_path):
    rows = pi.generate_patch_index(DATA, "example_data.h5", "example_data_HR.h5", str(tmp_path / "a.csv"), patch_size=16,
                                   n_patch=2, all_rotation=True, seed=1)
    assert len(rows) == 20
    assert sorted((r["rotation_plane"], r["rotation_degree_idx"]) for r in rows[1:10]) == [(p, d) for p in (1, 2, 3) for d in (1, 2, 3)]
    # impossible coverage: gives up after 100 misses without looping forever
    none = pi.generate_patch_index(DATA, "example_data.h5", "example_data_HR.h5", str(tmp_path / "n.csv"), patch_size=16,
                                   n_patch=2, minimum_coverage=1.1, seed=1)
    assert none == []
    one = pi.generate_patch_index(DATA, "example_data.h5", "example_data_HR.h5", str(tmp_path / "o.csv"), patch_size=16,
                                  n_patch=2, minimum_coverage=1.1, n_empty_patch_allowed=1, seed=1)
    assert len(one) == 2


def test_pinned_to_reference_generator(tmp_path):
    """byte-for-byte the CSV the reference's PatchData.generate_random_patches writes after random.seed(s)
    (tests/golden/patch_index_golden.json, produced by tests/golden/make_golden_patch_index.py importing
    /root/reference/src/prepare_data/PatchData.py): acceptance / skip / not_found > 100 state machine, empty-patch budget,
    all-rotation expansion, the single random rotation, and the 3-decimal coverage rounding (PatchData.py:12-68,97-102)."""
    import json
    gold = json.load(open(os.path.join(HERE, "golden", "patch_index_golden.json")))
    assert len(gold["cases"]) >= 8
    with h5io.H5File(os.path.join(DATA, gold["lr_file"])) as f:
        mask = f["mask"].read()[0]
    for i, case in enumerate(gold["cases"]):
        p = case["params"]
        out = str(tmp_path / ("g%d.csv" % i))
        rows = pi.generate_patch_index(DATA, gold["lr_file"], gold["hr_file"], out, patch_size=p["patch_size"], n_patch=p["n_patch"],
                                       n_empty_patch_allowed=p["n_empty_patch_allowed"], all_rotation=p["all_rotation"],
                                       mask_threshold=p["mask_threshold"], minimum_coverage=p["minimum_coverage"], seed=p["seed"])
        assert open(out, newline="").read() == case["csv"], p
        assert len(rows) == case["csv"].count("\n") - 1
        bm = (mask >= p["mask_threshold"]) * 1
        for c in case["coverage"]:
            assert pi.patch_coverage(bm, tuple(c["start"]), p["patch_size"]) == c["coverage"]
