"""Low-res dataset synthesis from a high-res CFD volume file (prepare_data/prepare_lowres_dataset.py): per row choose a
magnitude level and VENCs, k-space down-sample u, v, w (fft_downsampling) and append the results to an HDF5 file with the
reference's dataset names.  The reference is a script with hard-coded paths; this is the same procedure as a function,
drawing from the same generators in the same order (random.choice for the venc mode, np.random for SNR / venc picks / noise)
so a seeded run is reproducible."""
import random

import numpy as np

from . import fft_downsampling as fft
from . import h5io
from .h5io import append_dataset

MAG_VALUES = np.asarray([60, 80, 120, 180, 240])                      # px values [0-4095]   (:33)
VENC_VALUES = np.asarray([0.3, 0.6, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5])    # m/s                  (:34)


def choose_venc():
    """68 % 'same' venc on all three components (prepare_lowres_dataset.py:9-14)."""
    return random.choice(['same'] * 68 + ['diff'] * 32)


def pick_vencs(max_u, max_v, max_w, venc_choice):
    """prepare_lowres_dataset.py:66-107."""
    all_max = np.array([max_u, max_v, max_w])
    if venc_choice == 'same':
        max_vel = np.max(all_max)
        if max_vel < 1.5:
            return 1.5, 1.5, 1.5
        venc = VENC_VALUES[np.where(VENC_VALUES > max_vel)][np.random.randint(2)]
        return venc, venc, venc
    vencs = [VENC_VALUES[np.where(VENC_VALUES > m)][np.random.randint(2)] for m in (max_u, max_v, max_w)]
    main_vel = int(np.argmax(all_max))
    if vencs[main_vel] < 1.5:
        vencs[main_vel] = 1.5
    return tuple(vencs)


def save_row(output_filepath, col_name, dataset):
    """prepare_data/h5functions.save_to_h5: one row appended along a new leading axis, float64 stored as float32."""
    append_dataset(output_filepath, col_name, np.expand_dims(np.asarray(dataset), axis=0))


def zoom_mask(mask, factor):
    """scipy.ndimage.zoom(mask, factor, order=1) (:128)."""
    import scipy.ndimage as ndimage
    return ndimage.zoom(mask, factor, order=1)


def prepare_lowres_dataset(input_filepath, output_filename, downsample=2, base_venc_multiplier=1.1, use_gpu=False):
    crop_ratio = 1 / downsample
    with h5io.open_read(input_filepath) as hf:
        rd = lambda n: np.asarray(hf[n][...] if hasattr(hf[n], "id") else hf[n].read())
        mask = rd('mask')[0]
        U, V, W = rd('u'), rd('v'), rd('w')
        umax, vmax, wmax = rd('u_max'), rd('v_max'), rd('w_max')
    down = fft.downsample_phase_img_gpu if use_gpu else fft.downsample_phase_img
    for idx in range(len(U)):
        targetSNRdb = np.random.randint(140, 170) / 10
        mag_image = mask * MAG_VALUES[idx % len(MAG_VALUES)]
        venc_u, venc_v, venc_w = pick_vencs(umax[idx] * base_venc_multiplier, vmax[idx] * base_venc_multiplier,
                                            wmax[idx] * base_venc_multiplier, choose_venc())
        lr_u, mag_u = down(U[idx], mag_image, venc_u, crop_ratio, targetSNRdb)
        lr_v, mag_v = down(V[idx], mag_image, venc_v, crop_ratio, targetSNRdb)
        lr_w, mag_w = down(W[idx], mag_image, venc_w, crop_ratio, targetSNRdb)
        for name, val in (("u", lr_u), ("v", lr_v), ("w", lr_w), ("mag_u", mag_u), ("mag_v", mag_v), ("mag_w", mag_w),
                          ("venc_u", venc_u), ("venc_v", venc_v), ("venc_w", venc_w), ("SNRdb", targetSNRdb)):
            save_row(output_filename, name, val)
        if idx == 0:
            save_row(output_filename, "mask", zoom_mask(mask, crop_ratio))
    return output_filename
