"""Minimal HDF5 reader / writer for the files on the 4DFlowNet path, used when h5py is not importable
(the ROCm image's interpreter has no h5py).  Covers exactly the subset those files use:

  reading : superblock v0/v1, object header v1 (+continuations), old-style groups (symbol table: B-tree v1 +
            local heap + SNOD), datasets with contiguous / compact / chunked (B-tree v1) layout, filters
            deflate + shuffle, little-endian IEEE float32/float64, int8..int64/uint8..uint64, fixed strings;
            v1 attributes are skipped.  That is what h5py/libhdf5 writes by default for the reference's data
            files (data/example_data*.h5: chunked, deflate-4) and for Keras `model.save` weight files.
  writing : one flat image per file: superblock v0, old-style groups, every dataset as ONE chunk (chunked
            layout, unlimited max dims like `maxshape=(None,...)` in prediction_utils.py:15-28) with optional
            deflate, v1 attributes with fixed-length string arrays (Keras layer_names / weight_names).
            "Append along axis 0" (h5util.py:5-23) is done by read-modify-rewrite of the whole file.

If h5py is importable it is used instead (same call surface: open_read(path)[name][index])."""
import os
import struct
import zlib

import numpy as np

try:                                   # pragma: no cover - not present in the ROCm image
    import h5py as _h5py
except Exception:                      # noqa: BLE001
    _h5py = None

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


# =====================================================================================================
# reader
# =====================================================================================================
class _Buf:
    def __init__(self, data):
        self.d = data

    def u8(self, o): return self.d[o]
    def u16(self, o): return struct.unpack_from("<H", self.d, o)[0]
    def u32(self, o): return struct.unpack_from("<I", self.d, o)[0]
    def u64(self, o): return struct.unpack_from("<Q", self.d, o)[0]


def _parse_datatype(b, o):
    cv = b.u8(o)
    cls, ver = cv & 0x0F, cv >> 4
    bits0 = b.u8(o + 1)
    size = b.u32(o + 4)
    if cls == 0:       # fixed point
        signed = bool(bits0 & 0x08)
        return np.dtype("<%s%d" % ("i" if signed else "u", size))
    if cls == 1:       # floating point
        if bits0 & 1:
            raise NotImplementedError("big-endian floats")
        return np.dtype("<f%d" % size)
    if cls == 3:       # fixed-length string
        return np.dtype("S%d" % size)
    raise NotImplementedError("HDF5 datatype class %d (version %d)" % (cls, ver))


class H5Dataset:
    def __init__(self, f, name, msgs):
        self._f = f
        self.name = name
        b = f._b
        self.shape = ()
        self.maxshape = None
        self.dtype = None
        self._layout = None
        self._filters = []
        for mtype, o, size in msgs:
            if mtype == 0x0001:
                ver, rank, flags = b.u8(o), b.u8(o + 1), b.u8(o + 2)
                p = o + (8 if ver == 1 else 4)
                self.shape = tuple(b.u64(p + 8 * i) for i in range(rank))
                if flags & 1:
                    p2 = p + 8 * rank
                    self.maxshape = tuple(None if b.u64(p2 + 8 * i) == UNDEF else b.u64(p2 + 8 * i) for i in range(rank))
            elif mtype == 0x0003:
                self.dtype = _parse_datatype(b, o)
            elif mtype == 0x0008:
                ver = b.u8(o)
                if ver != 3:
                    raise NotImplementedError("data layout message version %d" % ver)
                cls = b.u8(o + 1)
                if cls == 0:
                    sz = b.u16(o + 2)
                    self._layout = ("compact", o + 4, sz)
                elif cls == 1:
                    self._layout = ("contiguous", b.u64(o + 2), b.u64(o + 10))
                elif cls == 2:
                    nd = b.u8(o + 2)
                    addr = b.u64(o + 3)
                    dims = tuple(b.u32(o + 11 + 4 * i) for i in range(nd))
                    self._layout = ("chunked", addr, dims)
            elif mtype == 0x000B:
                ver, nf = b.u8(o), b.u8(o + 1)
                p = o + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid = b.u16(p)
                    if ver == 1 or fid >= 256:
                        nlen = b.u16(p + 2); p += 4
                    else:
                        nlen = 0; p += 2
                    flags, ncd = b.u16(p), b.u16(p + 2)
                    p += 4
                    if nlen:
                        p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
                    cd = [b.u32(p + 4 * i) for i in range(ncd)]
                    p += 4 * ncd
                    if ver == 1 and ncd % 2:
                        p += 4
                    self._filters.append((fid, cd))

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    def _defilter(self, raw, mask):
        for i, (fid, cd) in reversed(list(enumerate(self._filters))):
            if mask & (1 << i):
                continue
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cd[0] if cd else self.dtype.itemsize
                a = np.frombuffer(raw, dtype=np.uint8)
                n = a.size // es
                raw = a[:n * es].reshape(es, n).T.tobytes() + a[n * es:].tobytes()
            else:
                raise NotImplementedError("HDF5 filter id %d" % fid)
        return raw

    def _chunks(self, addr, nd):
        b = self._f._b
        if addr == UNDEF:
            return
        if bytes(b.d[addr:addr + 4]) != b"TREE":
            raise ValueError("bad chunk B-tree node")
        level, used = b.u8(addr + 5), b.u16(addr + 6)
        p = addr + 24
        ksz = 8 + 8 * nd
        for i in range(used):
            csize, mask = b.u32(p), b.u32(p + 4)
            offs = tuple(b.u64(p + 8 + 8 * j) for j in range(nd))
            child = b.u64(p + ksz)
            if level == 0:
                yield csize, mask, offs, child
            else:
                for c in self._chunks(child, nd):
                    yield c
            p += ksz + 8

    def read(self):
        b = self._f._b
        n = int(np.prod(self.shape)) if self.shape else 1
        kind = self._layout[0]
        if kind == "compact":
            _, o, sz = self._layout
            return np.frombuffer(bytes(b.d[o:o + sz]), dtype=self.dtype, count=n).reshape(self.shape).copy()
        if kind == "contiguous":
            _, addr, sz = self._layout
            if addr == UNDEF:
                return np.zeros(self.shape, dtype=self.dtype)
            return np.frombuffer(bytes(b.d[addr:addr + n * self.dtype.itemsize]), dtype=self.dtype).reshape(self.shape).copy()
        _, addr, cdims = self._layout
        nd = len(cdims)
        cshape = cdims[:-1]
        out = np.zeros(self.shape, dtype=self.dtype)
        for csize, mask, offs, child in self._chunks(addr, nd):
            raw = self._defilter(bytes(b.d[child:child + csize]), mask)
            chunk = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(cshape))).reshape(cshape)
            sl_out, sl_in = [], []
            for d in range(len(self.shape)):
                s = offs[d]
                e = min(s + cshape[d], self.shape[d])
                sl_out.append(slice(s, e)); sl_in.append(slice(0, e - s))
            out[tuple(sl_out)] = chunk[tuple(sl_in)]
        return out

    def __getitem__(self, idx):
        return self.read()[idx]

    def __array__(self, dtype=None):
        a = self.read()
        return a if dtype is None else a.astype(dtype)


class H5Group:
    def __init__(self, f, name, btree, heap):
        self._f = f
        self.name = name
        self._entries = dict(f._read_symbols(btree, heap))

    def keys(self):
        return list(self._entries.keys())

    def __iter__(self):
        return iter(self.keys())

    def __contains__(self, k):
        return self.get(k) is not None

    def get(self, k, default=None):
        try:
            return self[k]
        except KeyError:
            return default

    def __getitem__(self, k):
        parts = [p for p in k.split("/") if p]
        obj = self
        for p in parts:
            if not isinstance(obj, H5Group) or p not in obj._entries:
                raise KeyError(k)
            obj = obj._f._object(obj._entries[p], (obj.name.rstrip("/") + "/" + p))
        return obj


class H5File(H5Group):
    def __init__(self, path):
        with open(path, "rb") as fh:
            data = fh.read()
        self._b = _Buf(data)
        self.filename = path
        base = data.find(SIG)
        if base != 0:
            raise ValueError("%s: not an HDF5 file (or user block present)" % path)
        ver = self._b.u8(8)
        if ver not in (0, 1):
            raise NotImplementedError("HDF5 superblock version %d" % ver)
        if self._b.u8(13) != 8 or self._b.u8(14) != 8:
            raise NotImplementedError("only 8-byte offsets/lengths")
        p = 24 + (4 if ver == 1 else 0)
        p += 32                                   # base, free-space, eof, driver addresses
        # root symbol table entry
        self._cache = {}
        hdr = self._b.u64(p + 8)
        ctype = self._b.u32(p + 16)
        if ctype == 1:
            btree, heap = self._b.u64(p + 24), self._b.u64(p + 32)
        else:
            btree, heap = self._stab_of(hdr)
        H5Group.__init__(self, self, "/", btree, heap)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    # ---- low-level pieces ----
    def _messages(self, addr):
        b = self._b
        if b.u8(addr) != 1:
            raise NotImplementedError("object header version %d (only v1 files are supported)" % b.u8(addr))
        nmsg = b.u16(addr + 2)
        size = b.u32(addr + 8)
        blocks = [(addr + 16, size)]
        msgs = []
        while blocks and len(msgs) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(msgs) < nmsg:
                mtype, msize = b.u16(p), b.u16(p + 2)
                o = p + 8
                if mtype == 0x0010:
                    blocks.append((b.u64(o), b.u64(o + 8)))
                msgs.append((mtype, o, msize))
                p = o + msize
        return msgs

    def _stab_of(self, hdr):
        for mtype, o, size in self._messages(hdr):
            if mtype == 0x0011:
                return self._b.u64(o), self._b.u64(o + 8)
        raise KeyError("no symbol table message")

    def _read_symbols(self, btree, heap):
        b = self._b
        if bytes(b.d[heap:heap + 4]) != b"HEAP":
            raise ValueError("bad local heap")
        hdata = b.u64(heap + 24)

        def name_at(off):
            s = hdata + off
            e = b.d.index(b"\0", s)
            return bytes(b.d[s:e]).decode()

        def walk(node):
            if bytes(b.d[node:node + 4]) == b"TREE":
                level, used = b.u8(node + 5), b.u16(node + 6)
                p = node + 24 + 8
                for _ in range(used):
                    child = b.u64(p)
                    for x in walk(child):
                        yield x
                    p += 16
            elif bytes(b.d[node:node + 4]) == b"SNOD":
                n = b.u16(node + 6)
                p = node + 8
                for _ in range(n):
                    yield name_at(b.u64(p)), b.u64(p + 8)
                    p += 40
            else:
                raise ValueError("bad group node")
        return list(walk(btree))

    def _object(self, hdr, name):
        if hdr in self._cache:
            return self._cache[hdr]
        msgs = self._messages(hdr)
        types = set(m[0] for m in msgs)
        if 0x0011 in types:
            bt, hp = self._stab_of(hdr)
            obj = H5Group(self, name, bt, hp)
        elif 0x0008 in types:
            obj = H5Dataset(self, name, msgs)
        else:
            raise NotImplementedError("object %s is neither an old-style group nor a dataset" % name)
        self._cache[hdr] = obj
        return obj


def open_read(path):
    """h5py.File(path,'r') if h5py exists, else the built-in reader (same subset of the call surface)."""
    if _h5py is not None:
        return _h5py.File(path, "r")
    return H5File(path)


def read_all(path, with_compression=False):
    """{name: ndarray} for every dataset, groups as nested dicts.  with_compression=True returns
    (ndarray, 'gzip' | None) leaves so a rewrite can keep each dataset's filter."""
    def rec(g):
        out = {}
        for k in g.keys():
            o = g[k]
            if hasattr(o, "keys"):
                out[k] = rec(o)
                continue
            arr = np.asarray(o[...] if _h5py is not None else o.read())
            if with_compression:
                comp = o.compression if _h5py is not None else ("gzip" if any(f[0] == 1 for f in o._filters) else None)
                out[k] = (arr, comp)
            else:
                out[k] = arr
        return out
    with open_read(path) as f:
        return rec(f)


# =====================================================================================================
# writer
# =====================================================================================================
def _pad8(b):
    return b + b"\0" * ((-len(b)) % 8)


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        if dt.itemsize == 4:
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        elif dt.itemsize == 8:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        else:
            raise NotImplementedError(dt)
        # class 1 v1; bits: little-endian, mantissa normalisation = implied msb (2<<4), sign position in byte 1
        return struct.pack("<BBBBI", 0x11, 0x20, dt.itemsize * 8 - 1, 0, dt.itemsize) + props
    if dt.kind in "iu":
        bits0 = 0x08 if dt.kind == "i" else 0
        return struct.pack("<BBBBI", 0x10, bits0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dt.itemsize)     # null-padded ASCII
    raise NotImplementedError("dtype %s" % dt)


def _dataspace_msg(shape, unlimited):
    rank = len(shape)
    flags = 1 if (unlimited and rank) else 0
    m = struct.pack("<BBB5x", 1, rank, flags)
    m += b"".join(struct.pack("<Q", s) for s in shape)
    if flags:
        m += struct.pack("<Q", UNDEF) + b"".join(struct.pack("<Q", s) for s in shape[1:])
    return m


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _attr_msg(name, value):
    value = np.asarray(value)
    nm = name.encode() + b"\0"
    dtm = _dtype_msg(value.dtype)
    dsm = _dataspace_msg(value.shape, False) if value.shape else struct.pack("<BBB5x", 1, 0, 0)
    body = struct.pack("<BxHHH", 1, len(nm), len(dtm), len(dsm)) + _pad8(nm) + _pad8(dtm) + _pad8(dsm) + value.tobytes()
    return _msg(0x000C, body)


class _Writer:
    def __init__(self):
        self.buf = bytearray()

    def tell(self):
        return len(self.buf)

    def alloc(self, data):
        self.buf += b"\0" * ((-len(self.buf)) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def object_header(self, msgs):
        body = b"".join(msgs)
        hdr = struct.pack("<BxHII4x", 1, len(msgs), 1, len(body))
        return self.alloc(hdr + body)

    def dataset(self, arr, compression):
        arr = np.ascontiguousarray(arr)
        if arr.dtype == np.float64 and False:
            arr = arr.astype(np.float32)
        raw = arr.tobytes()
        msgs = [_msg(0x0001, _dataspace_msg(arr.shape, True)), _msg(0x0003, _dtype_msg(arr.dtype), flags=1),
                _msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0))]          # fill value v2: alloc late, never written, undefined
        if arr.ndim == 0 or arr.size == 0:
            addr = self.alloc(raw) if raw else UNDEF
            msgs.append(_msg(0x0008, struct.pack("<BBQQ", 3, 1, addr, len(raw))))
            return self.object_header(msgs)
        filt = compression in ("gzip", "deflate", True)
        data = zlib.compress(raw, 4) if filt else raw
        caddr = self.alloc(data)
        nd = arr.ndim + 1
        key = lambda size, offs: struct.pack("<II", size, 0) + b"".join(struct.pack("<Q", o) for o in offs) + struct.pack("<Q", 0)
        node = b"TREE" + struct.pack("<BBHQQ", 1, 0, 1, UNDEF, UNDEF)
        node += key(len(data), (0,) * arr.ndim) + struct.pack("<Q", caddr) + key(0, tuple(arr.shape))
        # libhdf5 reads whole nodes: pad to the full size for the default indexed-storage K = 32
        full = 24 + (2 * 32 + 1) * (8 + 8 * nd) + 2 * 32 * 8
        node += b"\0" * (full - len(node))
        baddr = self.alloc(node)
        if filt:
            name = _pad8(b"deflate\0")
            msgs.append(_msg(0x000B, struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", 1, len(name), 1, 1) + name +
                             struct.pack("<I", 4) + b"\0" * 4))
        layout = struct.pack("<BBBQ", 3, 2, nd, baddr) + b"".join(struct.pack("<I", s) for s in arr.shape) + \
            struct.pack("<I", arr.dtype.itemsize)
        msgs.append(_msg(0x0008, layout))
        return self.object_header(msgs)

    def group(self, children, attrs=None):
        """children: {name: object header address}.  Returns (header address, btree address, heap address)."""
        names = sorted(children.keys(), key=lambda s: s.encode())
        heap_data = bytearray(b"\0" * 8)
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += _pad8(n.encode() + b"\0")
        free_off = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16)                # free block: next = 1 (none), size 16
        hd_addr = self.alloc(bytes(heap_data))
        heap = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, hd_addr))
        snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
        for n in names:
            snod += struct.pack("<QQII16x", offs[n], children[n], 0, 0)
        snod += b"\0" * (40 * (2 * _Writer.LEAF_K - len(names)))
        saddr = self.alloc(snod)
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF)
        node += struct.pack("<Q", 0)
        if names:
            node += struct.pack("<QQ", saddr, offs[names[-1]])
        node += b"\0" * (16 * (2 * _Writer.INTERNAL_K + 1))
        btree = self.alloc(node)
        msgs = [_msg(0x0011, struct.pack("<QQ", btree, heap))]
        for k, v in (attrs or {}).items():
            msgs.append(_attr_msg(k, v))
        return self.object_header(msgs), btree, heap

    LEAF_K = 128          # up to 256 links per group in one symbol-table node
    INTERNAL_K = 16


def write_file(path, tree, compression=None, attrs=None):
    """tree: {name: ndarray | {nested group}}.  attrs: {group path ('' = root): {attr name: array}}."""
    attrs = attrs or {}
    w = _Writer()
    w.buf += b"\0" * 96                                          # superblock placeholder

    def build(node, prefix):
        children = {}
        for k, v in node.items():
            if isinstance(v, dict):
                children[k] = build(v, prefix + "/" + k)[0]
            elif isinstance(v, tuple):                           # (array, per-dataset compression)
                children[k] = w.dataset(np.asarray(v[0]), v[1])
            else:
                children[k] = w.dataset(np.asarray(v), compression)
        if len(children) > 2 * _Writer.LEAF_K:
            raise NotImplementedError("more than %d links in one group" % (2 * _Writer.LEAF_K))
        return w.group(children, attrs.get(prefix.strip("/"), None))

    root_hdr, btree, heap = build(tree, "")
    eof = len(w.buf) + ((-len(w.buf)) % 8)
    w.buf += b"\0" * (eof - len(w.buf))
    sb = SIG + struct.pack("<BBBxBBBxHHI", 0, 0, 0, 0, 8, 8, _Writer.LEAF_K, _Writer.INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", btree, heap)
    assert len(sb) == 96, len(sb)
    w.buf[0:96] = sb
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(bytes(w.buf))
    os.replace(tmp, path)


def append_dataset(path, col_name, dataset, compression=None):
    """prediction_utils.save_to_h5 / h5util.save_predictions semantics (prediction_utils.py:15-28, h5util.py:5-23):
    float64 is stored as float32; create the dataset (resizable along axis 0) or append along axis 0."""
    append_datasets(path, [(col_name, dataset)], compression=compression)


def append_datasets(path, columns, compression=None):
    """Several (col_name, dataset) appends in ONE pass over the file.  Without h5py an append is a read-modify-rewrite of
    the whole file, so callers that add many columns per row (quicksave: 12, predictor: 4) batch them here instead of
    paying one rewrite per column."""
    cols = []
    for col_name, dataset in columns:
        dataset = np.asarray(dataset)
        if dataset.dtype == np.float64:
            dataset = dataset.astype(np.float32)
        cols.append((col_name, dataset))
    d = os.path.dirname(path)
    if d and not os.path.isdir(d):
        os.makedirs(d)
    if _h5py is not None:                                        # pragma: no cover
        with _h5py.File(path, "a") as hf:
            for col_name, dataset in cols:
                if col_name not in hf:
                    ms = (None,) + tuple(dataset.shape[1:]) if dataset.ndim > 1 else (None,)
                    hf.create_dataset(col_name, data=dataset, maxshape=ms, compression=compression)
                else:
                    hf[col_name].resize(hf[col_name].shape[0] + dataset.shape[0], axis=0)
                    hf[col_name][-dataset.shape[0]:] = dataset
        return
    tree = read_all(path, with_compression=True) if os.path.exists(path) else {}
    for col_name, dataset in cols:
        if col_name in tree:
            old, comp = tree[col_name]
            tree[col_name] = (np.concatenate([old, dataset.astype(old.dtype)], axis=0), comp)
        else:
            tree[col_name] = (dataset, compression)
    write_file(path, tree)


# =====================================================================================================
# Keras weight files  (model.save / load_weights: TrainerController.py:356,394; predictor.py:61)
# =====================================================================================================
def write_keras_weights(path, layers):
    """layers: [(layer_name, kernel, bias_or_None)] in creation order.  Layout written:
    /model_weights/<layer>/<layer>/kernel:0 (+ bias:0) with the layer_names / weight_names / backend /
    keras_version attributes Keras' load_weights_from_hdf5_group reads."""
    mw, attrs = {}, {}
    fixed = lambda strs: np.array([s.encode() for s in strs], dtype="S%d" % max(1, max(len(s) for s in strs)))
    for name, k, b in layers:
        inner = {"kernel:0": np.asarray(k, np.float32)}
        wn = ["%s/kernel:0" % name]
        if b is not None:
            inner["bias:0"] = np.asarray(b, np.float32)
            wn.append("%s/bias:0" % name)
        mw[name] = {name: inner}
        attrs["model_weights/%s" % name] = {"weight_names": fixed(wn)}
    attrs["model_weights"] = {"layer_names": fixed([n for n, _, _ in layers]), "backend": np.array(b"tensorflow"),
                              "keras_version": np.array(b"2.3.0-tf")}
    attrs[""] = {"backend": np.array(b"tensorflow"), "keras_version": np.array(b"2.3.0-tf")}
    write_file(path, {"model_weights": mw}, attrs=attrs)


def read_keras_weights(path):
    """{layer_name: (kernel, bias_or_None)} from a Keras .h5 (full model with /model_weights, or weights-only)."""
    with open_read(path) as f:
        g = f["model_weights"] if "model_weights" in f else f
        out = {}
        for lname in g.keys():
            lg = g[lname]
            if not hasattr(lg, "keys") or lname not in lg.keys():
                continue
            inner = lg[lname]
            k = np.asarray(inner["kernel:0"][...] if _h5py is not None else inner["kernel:0"].read())
            b = None
            if "bias:0" in inner.keys():
                b = np.asarray(inner["bias:0"][...] if _h5py is not None else inner["bias:0"].read())
            out[lname] = (k, b)
    return out
