"""Sliding-window tiler / stitcher, API-compatible with src/Network/PatchGenerator.py.

LR volume -> zero-pad 2 voxels per side (+ far-side pad so the stride P-4 tiles it exactly) -> overlapping P^3
patches at stride P-4 -> (network) -> crop 2*R HR voxels per side of every patch -> stitch -> crop the far-side pad.
Same outputs as the reference (pinned by tests/golden/reference_golden.json); the implementation is vectorised
(strided window view for patchify, one reshape/transpose for the stitch) instead of Python loops."""
import numpy as np


class PatchGenerator:
    def __init__(self, patch_size, res_increase):
        self.patch_size = patch_size
        self.effective_patch_size = patch_size - 4      # 2 voxels stripped per side on LR (PatchGenerator.py:8)
        self.res_increase = res_increase
        self.padding = (0, 0, 0)
        self.nr_x = self.nr_y = self.nr_z = 0

    # ---- padding rule of _pad_to_patch_size_with_overlap (PatchGenerator.py:53-86) ----
    def _far_pad(self, padded_len):
        side_pad = (self.patch_size - self.effective_patch_size) // 2
        res = padded_len % self.effective_patch_size
        if res > 2 * side_pad:
            return self.patch_size - res
        return 2 * side_pad - res

    def _pad_to_patch_size_with_overlap(self, img):
        side_pad = (self.patch_size - self.effective_patch_size) // 2
        img = np.pad(img, ((side_pad, side_pad),) * 3, 'constant')
        pads = tuple(self._far_pad(n) for n in img.shape)
        img = np.pad(img, tuple((0, p) for p in pads), 'constant')
        self.padding = tuple(p * self.res_increase for p in pads)       # HR voxels to crop after stitching
        return img

    def _generate_overlapping_patches(self, img):
        P, E = self.patch_size, self.effective_patch_size
        img = self._pad_to_patch_size_with_overlap(img)
        all_pads = P - E
        nr = tuple((n - all_pads) // E for n in img.shape)
        win = np.lib.stride_tricks.sliding_window_view(img, (P, P, P))[::E, ::E, ::E]
        win = win[:nr[0], :nr[1], :nr[2]]
        stack = np.ascontiguousarray(win).reshape((-1, P, P, P))
        return stack, nr[0], nr[1], nr[2]

    def patchify(self, dataset):
        """-> ((u,v,w stacks), (mag stacks)), each (n_patches,P,P,P,1).  PatchGenerator.py:13-40."""
        stacks = []
        for a in (dataset.u, dataset.v, dataset.w, dataset.mag_u, dataset.mag_v, dataset.mag_w):
            s, i, j, k = self._generate_overlapping_patches(a)
            stacks.append(np.expand_dims(s, -1))
        self.nr_x, self.nr_y, self.nr_z = i, j, k
        return tuple(stacks[:3]), tuple(stacks[3:])

    def unpatchify(self, results):
        return tuple(self._patchup_with_overlap(results[:, :, :, :, c], self.nr_x, self.nr_y, self.nr_z) for c in range(3))

    def _patchup_with_overlap(self, patches, x, y, z):
        """patches (n,S,S,S) in (i,j,k) order with k fastest -> stitched volume.  PatchGenerator.py:116-154."""
        side_pad_hr = ((self.patch_size - self.effective_patch_size) // 2) * self.res_increase
        S = patches.shape[1]
        core = patches[:, side_pad_hr:S - side_pad_hr, side_pad_hr:S - side_pad_hr, side_pad_hr:S - side_pad_hr]
        c = core.shape[1]
        nx = len(patches) // (y * z)
        vol = core[:nx * y * z].reshape(nx, y, z, c, c, c).transpose(0, 3, 1, 4, 2, 5).reshape(nx * c, y * c, z * c)
        px, py, pz = self.padding
        if px > 0:
            vol = vol[:-px]
        if py > 0:
            vol = vol[:, :-py]
        if pz > 0:
            vol = vol[:, :, :-pz]
        return vol
