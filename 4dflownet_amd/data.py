"""Data side of the path, API-compatible with the reference:

  PatchHandler3D  (src/Network/PatchHandler3D.py)  CSV rows + HDF5 volumes -> batches of 11-tuples
  ImageDataset    (src/utils/ImageDataset.py)      one inference volume, normalised
  load_indexes    (src/trainer.py:5-10)            patch-index CSV -> (N,10) unicode array

Unlike the reference (two h5py.File opens + gzip chunk decodes PER SAMPLE, PatchHandler3D.py:122,133), every
volume is decoded once and cached; a sample is then pure slicing.  Outputs are bit-identical to the reference's
loader (pinned by tests/golden/reference_golden.json)."""
import os

import numpy as np

from . import h5io, parallel


def load_indexes(index_file):
    """np.genfromtxt(..., delimiter=',', skip_header=True, dtype='unicode')  (trainer.py:9)."""
    return np.genfromtxt(index_file, delimiter=',', skip_header=True, dtype='unicode')


# (plane, k) -> (source component of each output component, sign of each output component) for PHASE images.
# Magnitude images use the same permutation with all signs +1.  Derived from the behaviour of
# rotate90 / rotate180_3d (PatchHandler3D.py:166-274): e.g. plane 1, 90 deg: v <- w, w <- -v.
_ROT = {
    (1, 1): ((0, 2, 1), (1, 1, -1)), (1, 3): ((0, 2, 1), (1, -1, 1)), (1, 2): ((0, 1, 2), (1, -1, -1)),
    (2, 1): ((2, 1, 0), (-1, 1, 1)), (2, 3): ((2, 1, 0), (1, 1, -1)), (2, 2): ((0, 1, 2), (-1, 1, -1)),
    (3, 1): ((1, 0, 2), (-1, 1, 1)), (3, 3): ((1, 0, 2), (1, -1, 1)), (3, 2): ((0, 1, 2), (-1, -1, 1)),
}
_AXES = {1: (0, 1), 2: (0, 2), 3: (1, 2)}


def rotate_vector_field(comps, rotation_idx, plane_nr, is_phase_image):
    """apply_rotation (PatchHandler3D.py:97-108): swap/sign the components, then np.rot90 each."""
    key = (plane_nr, rotation_idx)
    if key not in _ROT:
        return tuple(comps)
    perm, sign = _ROT[key]
    out = []
    for i in range(3):
        c = comps[perm[i]]
        if is_phase_image and sign[i] < 0:
            c = c * np.float32(-1)
        out.append(np.rot90(c, k=rotation_idx, axes=_AXES[plane_nr]))
    return tuple(out)


def rotate_object(img, rotation_idx, plane_nr):
    """PatchHandler3D.rotate_object (:83-95)."""
    if plane_nr not in _AXES:
        return img
    return np.rot90(img, k=rotation_idx, axes=_AXES[plane_nr])


class _VolumeCache:
    """Decoded HDF5 datasets, keyed by (path, file mtime, file size, dataset name): a file rewritten in the same process
    (prepare_lowres followed by predict) is re-read, never served stale.  Bounded: least-recently-used datasets are dropped
    once the decoded bytes exceed `max_bytes` (FDN_VOLUME_CACHE_BYTES, default 8 GiB -- a clinical 4D file with tens of
    frames is a few GB; the reference holds nothing and re-opens the file for every row, PatchHandler3D.py:133-147)."""

    def __init__(self, max_bytes=None):
        import collections
        import threading
        self._d = collections.OrderedDict()
        self._lock = threading.RLock()             # the loader's worker threads share one cache
        self._bytes = 0
        self.max_bytes = int(os.environ.get("FDN_VOLUME_CACHE_BYTES", 8 << 30)) if max_bytes is None else int(max_bytes)

    @staticmethod
    def _file_id(path):
        st = os.stat(path)
        return (os.path.realpath(path), st.st_mtime_ns, st.st_size)

    def get(self, path, name):
        """The lock covers the dictionary only; the file read runs outside it, so the loader's n_parallel workers decode different
        datasets concurrently (two workers missing on the same key both read it; the second result is dropped)."""
        fid = self._file_id(path)
        key = fid + (name,)
        with self._lock:
            if key in self._d:
                self._d.move_to_end(key)
                return self._d[key]
        with h5io.open_read(path) as f:
            obj = f.get(name)
            arr = None if obj is None else np.asarray(obj[...] if hasattr(obj, "id") else obj.read())
        with self._lock:
            if key in self._d:                      # another worker was faster
                self._d.move_to_end(key)
                return self._d[key]
            for k in [k for k in self._d if k[0] == fid[0] and k[:3] != fid]:      # same file, older contents
                self._bytes -= 0 if self._d[k] is None else self._d[k].nbytes
                del self._d[k]
            self._d[key] = arr
            self._bytes += 0 if arr is None else arr.nbytes
            while self._bytes > self.max_bytes and len(self._d) > 1:
                _, old = self._d.popitem(last=False)
                self._bytes -= 0 if old is None else old.nbytes
        return arr


class _BatchedDataset:
    """What initialize_dataset returns: iterable of batched 11-tuples; reshuffled on every pass
    (ds.shuffle(len).map(load, num_parallel_calls).batch(bs).prefetch(bs), PatchHandler3D.py:25-36).

    Like the reference's tf.data pipeline it runs AHEAD of the consumer: a producer thread assembles up to `prefetch` batches
    (FDN_LOADER_PREFETCH, default 2; 0 = synchronous) while the train step of the previous one runs, loading the samples of a batch
    on `n_parallel` worker threads (the `map` parallelism: None = min(4, cores), as AUTOTUNE would pick; <= 1 = in the producer
    thread).  pinned=True (scripts/trainer.py with FDN_HOST_LOADER=1; FDN_LOADER_PINNED=1) stacks the samples straight into PINNED
    host buffers, so the consumer's host-to-device copy is one DMA per tensor instead of a pageable copy; those buffers form a
    ring of prefetch + 3 batches owned by the iterator: while the consumer works on batch k (and may still hold batch k - 1, e.g.
    for a non-blocking copy), the producer has at most batches k + 1 .. k + prefetch queued and k + prefetch + 1 under assembly --
    prefetch + 3 distinct slots, so a batch stays untouched until the consumer has asked for TWO more.  The default hands out
    fresh arrays the caller may keep, like tf.data does."""

    def __init__(self, handler, indexes, shuffle, seed, shard, n_parallel=None, prefetch=None, pinned=None):
        self.h = handler
        self.indexes = np.atleast_2d(indexes)
        self.sampler = parallel.ShardedIndexSampler(len(self.indexes), handler.batch_size, shuffle, seed,
                                                    rank_=shard[0], world=shard[1])
        self.n_parallel = min(4, os.cpu_count() or 1) if n_parallel is None else int(n_parallel)
        self.prefetch = int(os.environ.get("FDN_LOADER_PREFETCH", "2")) if prefetch is None else int(prefetch)
        self.pinned = (os.environ.get("FDN_LOADER_PINNED", "0") not in ("", "0")) if pinned is None else bool(pinned)

    def __len__(self):
        return len(self.sampler)

    def _nslot(self):
        return max(self.prefetch, 0) + 3

    def _new_ring(self):
        """Ring of pinned batch buffers (plain numpy without a GPU) for ONE iterator: slot -> 11 arrays of the full batch shape.
        (Per iterator: an abandoned iterator's producer may still be assembling when the next epoch's iterator starts.)"""
        P, H, B = self.h.patch_size, self.h.patch_size * self.h.res_increase, self.h.batch_size
        shapes = [(B, P, P, P, 1)] * 6 + [(B, H, H, H, 1)] * 3 + [(B,), (B, H, H, H)]
        pin = False
        try:
            import torch
            pin = torch.cuda.is_available()
        except Exception:
            pass

        def alloc(shape):
            if pin:
                import torch
                return torch.empty(shape, dtype=torch.float32).pin_memory().numpy()
            return np.empty(shape, np.float32)
        return [[alloc(sh) for sh in shapes] for _ in range(self._nslot())]

    def _assemble(self, rows, pool, bufs):
        if len(rows) == 0:
            P, H = self.h.patch_size, self.h.patch_size * self.h.res_increase
            z = lambda *s: np.zeros(s, np.float32)
            return tuple([z(0, P, P, P, 1)] * 6 + [z(0, H, H, H, 1)] * 3 + [z(0), z(0, H, H, H)])
        load = lambda r: self.h.load_patches_from_index_file(self.indexes[r])
        samples = list(pool.map(load, rows)) if pool is not None else [load(r) for r in rows]
        if bufs is None:
            return tuple(np.stack([s_[i] for s_ in samples], axis=0) for i in range(11))
        out = []
        for i in range(11):
            dst = bufs[i][:len(samples)]
            if tuple(dst.shape[1:]) != tuple(np.shape(samples[0][i])):      # a patch cut short by the volume's edge: no fixed buffer
                out.append(np.stack([s_[i] for s_ in samples], axis=0))
                continue
            for b, s_ in enumerate(samples):
                dst[b] = s_[i]
            out.append(dst)
        return tuple(out)

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(self.n_parallel) if self.n_parallel > 1 else None
        nslot = self._nslot()
        ring = self._new_ring() if self.pinned else None
        slot = lambda k: None if ring is None else ring[k % nslot]
        if self.prefetch <= 0:
            try:
                for k, rows in enumerate(self.sampler):
                    yield self._assemble(rows, pool, slot(k))
            finally:
                if pool is not None:
                    pool.shutdown(wait=False)
            return
        import queue
        import threading
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def put(msg):                                       # every put gives up once the consumer is gone: no producer is left blocked
            while not stop.is_set():
                try:
                    q.put(msg, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                for k, rows in enumerate(self.sampler):
                    if stop.is_set() or not put(("batch", self._assemble(rows, pool, slot(k)))):
                        return
                put(("end", None))
            except BaseException as e:                      # surfaces in the consumer, not in a dead thread
                put(("error", e))

        t = threading.Thread(target=produce, name="fdn-loader", daemon=True)
        t.start()
        try:
            while True:
                kind, item = q.get()
                if kind == "end":
                    break
                if kind == "error":
                    raise item
                yield item
        finally:
            stop.set()
            t.join()                                        # the producer leaves within one put timeout (or one batch assembly) ...
            if pool is not None:
                pool.shutdown(wait=False)                   # ... and only then does its worker pool go away


class PatchHandler3D:
    def __init__(self, data_dir, patch_size, res_increase, batch_size, mask_threshold=0.6):
        self.patch_size = patch_size
        self.res_increase = res_increase
        self.batch_size = batch_size
        self.mask_threshold = mask_threshold
        self.data_directory = data_dir
        self.hr_colnames = ['u', 'v', 'w']
        self.lr_colnames = ['u', 'v', 'w']
        self.venc_colnames = ['venc_u', 'venc_v', 'venc_w']
        self.mag_colnames = ['mag_u', 'mag_v', 'mag_w']
        self.mask_colname = 'mask'
        self._cache = _VolumeCache()

    def initialize_dataset(self, indexes, shuffle, n_parallel=None, seed=0, shard=None, prefetch=None, pinned=None):
        """indexes: (N,10) array from load_indexes.  n_parallel = worker threads that load the samples of a batch (the reference's
        `map(..., num_parallel_calls=n_parallel)`; None = auto), prefetch = batches assembled ahead of the consumer (the
        reference's `.prefetch`; None = FDN_LOADER_PREFETCH or 2, 0 = synchronous), pinned = stack into a ring of pinned host
        buffers (see _BatchedDataset).  shard=(rank, world) splits every global batch across ranks; default: the current
        torch.distributed rank/world (single process -> no sharding)."""
        print("Total dataset:", len(np.atleast_2d(indexes)), 'shuffle', shuffle)
        if shard is None:
            shard = (parallel.rank(), parallel.world_size())
        return _BatchedDataset(self, indexes, shuffle, seed, shard, n_parallel, prefetch, pinned)

    def load_data_using_patch_index(self, indexes):
        """The tf.py_function bridge of the reference (PatchHandler3D.py:40-47) is unnecessary here; same result."""
        return self.load_patches_from_index_file(indexes)

    @staticmethod
    def _cell(c):
        if hasattr(c, "numpy"):
            c = c.numpy()
        if isinstance(c, bytes):
            c = c.decode()
        return c

    def load_patches_from_index_file(self, indexes):
        """One CSV row [source,target,index,start_x,start_y,start_z,rotate,rotation_plane,rotation_degree_idx,
        coverage] -> (u,v,w, mag_u,mag_v,mag_w, u_hr,v_hr,w_hr)[...,None], venc (), mask (PR,PR,PR).
        PatchHandler3D.py:49-81."""
        row = [self._cell(c) for c in indexes]
        lr_path = '{}/{}'.format(self.data_directory, row[0])
        hr_path = '{}/{}'.format(self.data_directory, row[1])
        idx = int(row[2])
        x0, y0, z0 = int(row[3]), int(row[4]), int(row[5])
        is_rotate, plane, rot_idx = int(row[6]), int(row[7]), int(row[8])
        P, R = self.patch_size, self.res_increase
        H = P * R
        lr_sl = np.index_exp[idx, x0:x0 + P, y0:y0 + P, z0:z0 + P]
        hr_sl = np.index_exp[idx, x0 * R:x0 * R + H, y0 * R:y0 * R + H, z0 * R:z0 * R + H]
        mask_sl = np.index_exp[0, x0 * R:x0 * R + H, y0 * R:y0 * R + H, z0 * R:z0 * R + H]   # one mask per file (:129)

        vol = self._cache.get
        hires = np.asarray([vol(hr_path, n)[hr_sl] for n in self.hr_colnames])
        mask = (vol(hr_path, self.mask_colname)[mask_sl] >= self.mask_threshold) * 1.
        lowres = np.asarray([vol(lr_path, n)[lr_sl] for n in self.lr_colnames])
        mags = np.asarray([vol(lr_path, n)[lr_sl] for n in self.mag_colnames])
        venc = np.max([vol(lr_path, n)[idx] for n in self.venc_colnames])
        hires = hires / venc                      # :152-154
        lowres = lowres / venc
        mags = mags / 4095.
        lr = tuple(lowres[i].astype('float32') for i in range(3))
        hr = tuple(hires[i].astype('float32') for i in range(3))
        mg = tuple(mags[i].astype('float32') for i in range(3))
        mask = mask.astype('float32')
        if is_rotate > 0:                          # :71-75
            lr = rotate_vector_field(lr, rot_idx, plane, True)
            hr = rotate_vector_field(hr, rot_idx, plane, True)
            mg = rotate_vector_field(mg, rot_idx, plane, False)
            mask = rotate_object(mask, rot_idx, plane)
        ex = lambda a: np.ascontiguousarray(a)[..., None]
        return (ex(lr[0]), ex(lr[1]), ex(lr[2]), ex(mg[0]), ex(mg[1]), ex(mg[2]), ex(hr[0]), ex(hr[1]), ex(hr[2]),
                venc.astype('float32'), np.ascontiguousarray(mask))


class ImageDataset:
    """src/utils/ImageDataset.py: one row of an inference file, velocities / venc, magnitudes / 4095."""

    def __init__(self):
        self.velocity_colnames = ['u', 'v', 'w']
        self.venc_colnames = ['venc_u', 'venc_v', 'venc_w']
        self.mag_colnames = ['mag_u', 'mag_v', 'mag_w']
        self.dx_colname = 'dx'
        self._cache = _VolumeCache()          # every dataset is decoded once per file, not once per row (ImageDataset.py:19-29 re-opens)

    def get_dataset_len(self, filepath):
        with h5io.open_read(filepath) as hl:
            return hl[self.velocity_colnames[0]].shape[0]

    def load_vectorfield(self, filepath, idx):
        rd = lambda n: self._cache.get(filepath, n)
        dxa = rd(self.dx_colname)
        dx = dxa[idx] if dxa is not None else None
        vel = np.asarray([rd(n)[idx] for n in self.velocity_colnames])
        mag = np.asarray([rd(n)[idx] for n in self.mag_colnames])
        venc = np.max([rd(n)[idx] for n in self.venc_colnames])
        vel = vel / venc
        mag = mag / 4095.
        self.u, self.v, self.w = (vel[i].astype('float32') for i in range(3))
        self.mag_u, self.mag_v, self.mag_w = (mag[i].astype('float32') for i in range(3))
        self.venc = venc.astype('float32')
        self.velocity_per_px = self.venc / 2048          # ImageDataset.py:31
        self.dx = dx

    def postprocess_result(self, results, zerofy=True):
        results = results * self.venc
        if zerofy:
            results[np.abs(results) < self.velocity_per_px] = 0
        return results
