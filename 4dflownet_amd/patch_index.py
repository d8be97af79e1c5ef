"""Patch-index CSV generator: the counterpart of the reference's src/prepare_data/prepare_patches.py + PatchData.py.

Same on-disk contract (header `source,target,index,start_x,start_y,start_z,rotate,rotation_plane,rotation_degree_idx,
coverage`; one un-rotated row per accepted patch followed by either all 9 (plane, degree) rotations or one random one),
same acceptance rule (fluid coverage of the LR mask inside the patch, rounded to 3 decimals, must reach
`minimum_coverage` unless the per-frame budget of empty patches allows it; give up after 100 misses) -- but seedable,
so the P=24 / P=32 index files the shipped CSVs lack (they were generated for patch_size 16) can be
produced reproducibly from data/example_data*.h5."""
import csv
import os

import numpy as np

from . import h5io

FIELDNAMES = ['source', 'target', 'index', 'start_x', 'start_y', 'start_z', 'rotate', 'rotation_plane',
              'rotation_degree_idx', 'coverage']


def patch_coverage(binary_mask, start, patch_size):
    """PatchData.calculate_patch_coverage (PatchData.py:97-102): non-zero fraction, rounded to 3 decimals."""
    x, y, z = start
    patch = binary_mask[x:x + patch_size, y:y + patch_size, z:z + patch_size]
    return float(np.round(np.count_nonzero(patch) / patch_size ** 3 * 1000) / 1000)


def generate_rows(lr_file, hr_file, index, n_patch, binary_mask, patch_size, minimum_coverage, empty_patch_allowed,
                  apply_all_rotation, rng):
    """Rows for one time frame (generate_random_patches, PatchData.py:12-68)."""
    if any(s < patch_size for s in binary_mask.shape):
        raise ValueError("patch_size %d does not fit the volume %s" % (patch_size, binary_mask.shape))
    rows, empties, misses = [], 0, 0
    while len([r for r in rows if r['rotate'] == 0]) < n_patch:
        if misses > 100:
            print("Cannot find enough patches above %s coverage, please lower the minimum_coverage" % minimum_coverage)
            break
        start = tuple(int(rng.integers(0, s - patch_size + 1)) for s in binary_mask.shape)
        cov = patch_coverage(binary_mask, start, patch_size)
        if cov < minimum_coverage:
            if empties < empty_patch_allowed:
                empties += 1
            else:
                misses += 1
                continue
        base = {'source': lr_file, 'target': hr_file, 'index': index, 'start_x': start[0], 'start_y': start[1],
                'start_z': start[2], 'rotate': 0, 'rotation_plane': 0, 'rotation_degree_idx': 0, 'coverage': cov}
        rows.append(base)
        if apply_all_rotation:
            for plane in (1, 2, 3):
                for deg in (1, 2, 3):
                    rows.append(dict(base, rotate=1, rotation_plane=plane, rotation_degree_idx=deg))
        else:
            rows.append(dict(base, rotate=1, rotation_plane=int(rng.integers(1, 4)), rotation_degree_idx=int(rng.integers(1, 4))))
    return rows


def generate_patch_index(base_path, lr_file, hr_file, output_filename, patch_size=16, n_patch=10, n_empty_patch_allowed=0,
                         all_rotation=False, mask_threshold=0.4, minimum_coverage=0.2, seed=0):
    """prepare_patches.py:14-47 as a function.  Returns the list of row dicts written."""
    with h5io.open_read(os.path.join(base_path, lr_file)) as f:
        n_frames = f['u'].shape[0]
        mask = np.asarray(f['mask'][...] if hasattr(f['mask'], 'id') else f['mask'].read())[0]   # one mask per file (:36-37)
    binary_mask = (mask >= mask_threshold) * 1
    rng = np.random.default_rng(seed)
    rows = []
    for index in range(n_frames):
        rows += generate_rows(lr_file, hr_file, index, n_patch, binary_mask, patch_size, minimum_coverage,
                              n_empty_patch_allowed, all_rotation, rng)
    with open(output_filename, mode='w', newline='') as csv_file:
        writer = csv.DictWriter(csv_file, fieldnames=FIELDNAMES)
        writer.writeheader()
        for r in rows:
            writer.writerow(r)
    return rows
