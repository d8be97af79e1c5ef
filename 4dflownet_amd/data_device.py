"""On-device input pipeline: PatchHandler3D whose volumes live in HBM.

Every HDF5 volume is decoded once and uploaded once; a batch is then produced by ten launches of
fdn_gather_patches (slice + np.rot90 + component swap/sign + /venc, /4095 or mask threshold), driven by a
descriptor table built from the CSV rows.  Outputs are device tensors, bit-identical to the host loader
(data.PatchHandler3D), which in turn is bit-identical to the reference's loader."""
import numpy as np
import torch

from . import _lib, parallel
from ._lib import check
from .data import _ROT, PatchHandler3D

DESC_DTYPE = np.dtype([('src', '<u8'), ('X', '<i4'), ('Y', '<i4'), ('Z', '<i4'), ('t', '<i4'), ('x0', '<i4'), ('y0', '<i4'),
                       ('z0', '<i4'), ('plane', '<i4'), ('k', '<i4'), ('mode', '<i4'), ('sign', '<f4'), ('div', '<f4')])
assert DESC_DTYPE.itemsize == 56


class _DeviceBatches:
    def __init__(self, handler, indexes, shuffle, seed, shard):
        self.h = handler
        self.indexes = np.atleast_2d(indexes)
        self.sampler = parallel.ShardedIndexSampler(len(self.indexes), handler.batch_size, shuffle, seed,
                                                    rank_=shard[0], world=shard[1])

    def __len__(self):
        return len(self.sampler)

    def __iter__(self):
        for rows in self.sampler:
            yield self.h.load_batch_device([self.indexes[r] for r in rows])


class DevicePatchHandler3D(PatchHandler3D):
    def __init__(self, data_dir, patch_size, res_increase, batch_size, mask_threshold=0.6, device=None):
        super().__init__(data_dir, patch_size, res_increase, batch_size, mask_threshold)
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self._dev = {}

    def _dvol(self, path, name):
        key = (path, name)
        if key not in self._dev:
            a = self._cache.get(path, name)
            if a is None:
                raise KeyError("%s has no dataset %r" % (path, name))
            self._dev[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        return self._dev[key]

    def initialize_dataset(self, indexes, shuffle, n_parallel=None, seed=0, shard=None, prefetch=None, pinned=None):
        print("Total dataset:", len(np.atleast_2d(indexes)), 'shuffle', shuffle)
        if shard is None:
            shard = (parallel.rank(), parallel.world_size())
        return _DeviceBatches(self, indexes, shuffle, seed, shard)

    def load_batch_device(self, rows):
        """rows: CSV rows -> the 11-tuple of PatchHandler3D batches as device tensors."""
        P, R = self.patch_size, self.res_increase
        H = P * R
        B = len(rows)
        z = lambda *s: torch.zeros(s, device=self.device)
        if B == 0:
            return tuple([z(0, P, P, P, 1)] * 6 + [z(0, H, H, H, 1)] * 3 + [z(0), z(0, H, H, H)])
        descs = [np.zeros(B, DESC_DTYPE) for _ in range(10)]        # lr u,v,w | mag u,v,w | hr u,v,w | mask
        vencs = np.zeros(B, np.float32)
        for b, row in enumerate(rows):
            row = [self._cell(c) for c in row]
            lr_path = '{}/{}'.format(self.data_directory, row[0])
            hr_path = '{}/{}'.format(self.data_directory, row[1])
            idx = int(row[2])
            x0, y0, z0 = int(row[3]), int(row[4]), int(row[5])
            is_rotate, plane, k = int(row[6]), int(row[7]), int(row[8])
            venc = np.max([self._cache.get(lr_path, n)[idx] for n in self.venc_colnames])
            vencs[b] = venc
            rot = _ROT.get((plane, k)) if is_rotate > 0 else None
            perm, sign = rot if rot else ((0, 1, 2), (1, 1, 1))
            pl, kk = (plane, k) if rot else (0, 0)

            def fill(d, vol, t, start, mode, sgn, div, pl_, kk_):
                S = P if d is descs[0] or d is descs[1] or d is descs[2] or d is descs[3] or d is descs[4] or d is descs[5] else H
                if t >= vol.shape[0] or any(s0 < 0 or s0 + S > n for s0, n in zip(start, vol.shape[1:])):
                    raise ValueError("patch %s + %d exceeds volume %s (row %s): the host loader would return a ragged patch"
                                     % (start, S, tuple(vol.shape), row))
                d[b] =(vol.data_ptr(), vol.shape[1], vol.shape[2], vol.shape[3], t, start[0], start[1], start[2], pl_, kk_,
                        mode, sgn, div)
            for c in range(3):
                src = perm[c]
                fill(descs[c], self._dvol(lr_path, self.lr_colnames[src]), idx, (x0, y0, z0), 0, float(sign[c]), float(venc), pl, kk)
                fill(descs[3 + c], self._dvol(lr_path, self.mag_colnames[src]), idx, (x0, y0, z0), 0, 1.0, 4095.0, pl, kk)
                fill(descs[6 + c], self._dvol(hr_path, self.hr_colnames[src]), idx, (x0 * R, y0 * R, z0 * R), 0, float(sign[c]),
                     float(venc), pl, kk)
            # the mask is rotated for any rotation_plane in 1..3 when rotate > 0 (rotate_object), one mask per file (row 0)
            mpl, mk = (plane, k) if (is_rotate > 0 and plane in (1, 2, 3)) else (0, 0)
            fill(descs[9], self._dvol(hr_path, self.mask_colname), 0, (x0 * R, y0 * R, z0 * R), 1, 1.0, float(self.mask_threshold),
                 mpl, mk % 4)
        table = torch.from_numpy(np.concatenate(descs).view(np.uint8)).to(self.device)      # one small H2D copy per batch
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        outs = []
        for i in range(10):
            S = P if i < 6 else H
            o = torch.empty((B, S, S, S, 1) if i < 9 else (B, S, S, S), device=self.device, dtype=torch.float32)
            check(lib.fdn_gather_patches(table.data_ptr() + i * B * DESC_DTYPE.itemsize, o.data_ptr(), B, S, stream),
                  "fdn_gather_patches")
            outs.append(o)
        self._keep = table                                   # keep the table alive until the next batch is built
        venc_t = torch.from_numpy(vencs).to(self.device)
        return tuple(outs[:9]) + (venc_t, outs[9])
