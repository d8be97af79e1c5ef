"""Loader / tiler / HDF5 reader pinned against the reference itself: tests/golden/reference_golden.json was
produced by importing the reference's PatchHandler3D / PatchGenerator / ImageDataset (tests/golden/make_golden.py).
Bit-exact (sha1 of the float32 bytes)."""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")
G = json.load(open(os.path.join(HERE, "golden", "reference_golden.json")))

data = importlib.import_module("4dflownet_amd.data")
tiler = importlib.import_module("4dflownet_amd.tiler")
h5io = importlib.import_module("4dflownet_amd.h5io")


def check(arr, d, name):
    arr = np.ascontiguousarray(np.asarray(arr))     # (0-d -> (1,), exactly like make_golden.digest)
    assert list(arr.shape) == d["shape"], (name, arr.shape, d["shape"])
    assert str(arr.dtype) == d["dtype"], (name, arr.dtype, d["dtype"])
    assert hashlib.sha1(np.ascontiguousarray(arr).tobytes()).hexdigest() == d["sha1"], \
        "%s: content differs (sum %r vs %r)" % (name, float(arr.astype(np.float64).sum()), d["sum"])


def test_hdf5_reader_reads_reference_files_bit_exact():
    for fname, dsets in G["hdf5"].items():
        with h5io.H5File(os.path.join(DATA, fname)) as f:
            assert sorted(f.keys()) == sorted(dsets.keys())
            for k, d in dsets.items():
                check(f[k].read(), d, fname + ":" + k)
                assert f[k].maxshape[0] is None          # extendible along axis 0 like the reference writes them


def test_csv_parse():
    idx = data.load_indexes(os.path.join(DATA, "train.csv"))
    assert list(idx.shape) == G["csv"]["shape"] and str(idx.dtype) == G["csv"]["dtype"]
    assert list(idx[0]) == G["csv"]["row0"] and list(idx[49]) == G["csv"]["row49"]


@pytest.mark.parametrize("case", G["loader"], ids=lambda c: "P%d_R%d_row%d" % (c["patch_size"], c["res_increase"], c["row"]))
def test_loader_matches_reference(case):
    ph = data.PatchHandler3D(DATA, case["patch_size"], case["res_increase"], 4, case["mask_threshold"])
    out = ph.load_patches_from_index_file(case["csv_row"])
    names = ["u", "v", "w", "u_mag", "v_mag", "w_mag", "u_hr", "v_hr", "w_hr", "venc", "mask"]
    for n, a in zip(names, out):
        check(a, case["outputs"][n], n)


def test_image_dataset_matches_reference():
    ds = data.ImageDataset()
    f = os.path.join(DATA, "example_data.h5")
    g = G["image_dataset"]
    assert ds.get_dataset_len(f) == g["len"]
    ds.load_vectorfield(f, 0)
    for k in ("u", "v", "w", "mag_u", "mag_v", "mag_w"):
        check(getattr(ds, k), g[k], k)
    assert float(ds.venc) == g["venc"] and float(ds.velocity_per_px) == g["velocity_per_px"]
    assert [float(x) for x in ds.dx] == g["dx"]


@pytest.mark.parametrize("case", G["tiler"], ids=lambda c: "P%d_R%d" % (c["patch_size"], c["res_increase"]))
def test_tiler_matches_reference(case):
    ds = data.ImageDataset()
    ds.load_vectorfield(os.path.join(DATA, "example_data.h5"), 0)
    P, R = case["patch_size"], case["res_increase"]
    pg = tiler.PatchGenerator(P, R)
    vel, mag = pg.patchify(ds)
    assert len(vel[0]) == case["n_patches"] and [pg.nr_x, pg.nr_y, pg.nr_z] == case["nr"]
    assert list(pg.padding) == case["padding"]
    check(vel[0], case["u_stacks"], "u_stacks"); check(vel[2], case["w_stacks"], "w_stacks")
    check(mag[2], case["wmag_stacks"], "wmag_stacks")
    res = np.stack([np.repeat(np.repeat(np.repeat(v[..., 0], R, 1), R, 2), R, 3) for v in vel], axis=-1)
    pu, pv, pw = pg.unpatchify(res)
    check(pu, case["stitched_u"], "stitched_u"); check(pw, case["stitched_w"], "stitched_w")
    # stitching the nearest-repeated patches reproduces the repeated volume exactly
    assert np.array_equal(pu, np.repeat(np.repeat(np.repeat(ds.u, R, 0), R, 1), R, 2))


def test_predictor_postprocess_and_output_file(tmp_path):
    """predictor.py:99-115 with the network replaced by nearest-neighbour repetition (host-only plumbing check)."""
    pred = importlib.import_module("4dflownet_amd.predictor")
    ds = data.ImageDataset()
    ds.load_vectorfield(os.path.join(DATA, "example_data.h5"), 0)
    pg = tiler.PatchGenerator(24, 2)
    vel, mag = pg.patchify(ds)
    res = np.stack([np.repeat(np.repeat(np.repeat(v[..., 0], 2, 1), 2, 2), 2, 3) for v in vel], axis=-1).astype(np.float64)
    v = pg._patchup_with_overlap(res[:, :, :, :, 0], pg.nr_x, pg.nr_y, pg.nr_z)
    v = v * ds.venc
    v[np.abs(v) < ds.velocity_per_px] = 0
    out = str(tmp_path / "result.h5")
    pred.save_to_h5(out, "u", np.expand_dims(v, 0), compression='gzip')
    pred.save_to_h5(out, "u", np.expand_dims(v, 0), compression='gzip')        # second row appends along axis 0
    pred.save_to_h5(out, "dx", np.expand_dims(ds.dx / 2, 0), compression='gzip')
    back = h5io.read_all(out)
    check(back["u"][:1], G["postprocess_u"], "postprocess_u")
    assert back["u"].shape == (2, 84, 76, 72) and np.array_equal(back["u"][0], back["u"][1])
    assert back["dx"].shape == (1, 3)


def test_dataset_batches_shuffle_and_ragged_tail():
    idx = data.load_indexes(os.path.join(DATA, "validate.csv"))
    ph = data.PatchHandler3D(DATA, 16, 2, 4, 0.6)
    ds = ph.initialize_dataset(idx, shuffle=True, seed=3, shard=(0, 1))
    assert len(ds) == 3
    batches = list(ds)
    assert [b[0].shape[0] for b in batches] == [4, 4, 2]                      # ragged last batch kept
    assert batches[0][0].shape == (4, 16, 16, 16, 1) and batches[0][6].shape == (4, 32, 32, 32, 1)
    assert batches[0][9].shape == (4,) and batches[0][10].shape == (4, 32, 32, 32)
    again = list(ds)                                                           # new epoch, new order
    cat = lambda bs: np.concatenate([b[0].sum(axis=(1, 2, 3, 4)) for b in bs])
    assert not np.array_equal(cat(batches), cat(again)) and np.allclose(np.sort(cat(batches)), np.sort(cat(again)))
    # two ranks see disjoint halves of every global batch
    d0 = list(ph.initialize_dataset(idx, shuffle=True, seed=3, shard=(0, 2)))
    d1 = list(ph.initialize_dataset(idx, shuffle=True, seed=3, shard=(1, 2)))
    assert [b[0].shape[0] for b in d0] == [4, 2] and [b[0].shape[0] for b in d1] == [4, 0]
    g = list(ph.initialize_dataset(idx, shuffle=True, seed=3, shard=(0, 1)))
    both = np.sort(np.concatenate([cat(d0), cat(d1)]))
    assert np.allclose(both, np.sort(cat(g)))


def test_keras_weight_file_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    layers = [("conv3d", rng.normal(size=(3, 3, 3, 3, 64)).astype(np.float32), rng.normal(size=64).astype(np.float32)),
              ("conv3d_1", rng.normal(size=(3, 3, 3, 64, 64)).astype(np.float32), None)]
    p = str(tmp_path / "w.h5")
    h5io.write_keras_weights(p, layers)
    back = h5io.read_keras_weights(p)
    assert set(back) == {"conv3d", "conv3d_1"}
    assert np.array_equal(back["conv3d"][0], layers[0][1]) and np.array_equal(back["conv3d"][1], layers[0][2])
    assert back["conv3d_1"][1] is None


def test_keras_weight_file_with_shifted_layer_names_loads_by_position(tmp_path):
    """Keras load_weights matches by topological position: a model built second in a process is saved as conv3d_36...;
    such files must load, while a wrong layer count or a bias mismatch must raise."""
    weights_io = importlib.import_module("4dflownet_amd.weights_io")
    rng = np.random.default_rng(1)
    shapes = [((3, 3, 3, 3, 64), True), ((3, 3, 3, 64, 64), False), ((3, 3, 3, 64, 1), True)]

    class L:
        def __init__(self, name, shp, bias):
            self.name, self.w, self.b = name, np.zeros(shp, np.float32), (np.zeros(shp[-1], np.float32) if bias else None)

    class M:
        layers = [L("conv3d" if i == 0 else "conv3d_%d" % i, shp, b) for i, (shp, b) in enumerate(shapes)]

        def set_weights(self, arrays):
            self.got = arrays

    file_layers = [("conv3d_%d" % (36 + i), rng.normal(size=shp).astype(np.float32), rng.normal(size=shp[-1]).astype(np.float32) if b else None)
                   for i, (shp, b) in enumerate(shapes)]
    p = str(tmp_path / "shifted.h5")
    h5io.write_keras_weights(p, file_layers)
    m = M()
    weights_io.load_model_weights(m, p)
    exp = [a for _, k, b in file_layers for a in ((k,) if b is None else (k, b))]
    assert len(m.got) == len(exp) and all(np.array_equal(x, y) for x, y in zip(m.got, exp))
    h5io.write_keras_weights(str(tmp_path / "short.h5"), file_layers[:2])
    with pytest.raises(KeyError):
        weights_io.load_model_weights(M(), str(tmp_path / "short.h5"))
    bad = [file_layers[0], (file_layers[1][0], file_layers[1][1], np.zeros(64, np.float32)), file_layers[2]]
    h5io.write_keras_weights(str(tmp_path / "bias.h5"), bad)
    with pytest.raises(ValueError):
        weights_io.load_model_weights(M(), str(tmp_path / "bias.h5"))


@pytest.mark.skipif(not os.path.exists("/opt/conda/bin/python3.9"), reason="h5py interpreter only exists in the build container")
def test_written_files_are_readable_by_real_h5py(tmp_path):
    import subprocess
    p = str(tmp_path / "x.h5")
    a = np.arange(2 * 3 * 4, dtype=np.float64).reshape(1, 2, 3, 4)
    h5io.append_dataset(p, "u", a, compression='gzip'); h5io.append_dataset(p, "u", a + 1, compression='gzip')
    code = ("import h5py,numpy as np\nf=h5py.File(%r,'r')\nu=f['u']\nassert u.shape==(2,2,3,4) and u.dtype==np.float32 and u.maxshape[0] is None\n"
            "assert u.compression=='gzip'\nassert np.array_equal(u[1], (np.arange(24).reshape(2,3,4)+1).astype('float32'))\nprint('ok')" % p)
    r = subprocess.run(["/opt/conda/bin/python3.9", "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_py_function_bridge_alias():
    ph = data.PatchHandler3D(DATA, 16, 2, 4, 0.6)
    row = G["loader"][0]["csv_row"]
    a = ph.load_data_using_patch_index(row); b = ph.load_patches_from_index_file(row)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_volume_cache_rereads_a_rewritten_file_and_is_bounded(tmp_path):
    """the decoded-volume cache is keyed on (path, mtime, size) and bounded (LRU by decoded bytes)."""
    import time
    data = importlib.import_module("4dflownet_amd.data")
    h5io = importlib.import_module("4dflownet_amd.h5io")
    p = str(tmp_path / "v.h5")
    a = np.arange(24, dtype=np.float32).reshape(1, 2, 3, 4)
    h5io.append_datasets(p, [("u", a), ("v", a + 1)])
    c = data._VolumeCache(max_bytes=a.nbytes + a.nbytes // 2)
    assert np.array_equal(c.get(p, "u"), a) and c.get(p, "u") is c.get(p, "u")        # second access is the cached array
    assert c.get(p, "missing") is None
    c.get(p, "v")                                                                  # over budget: the least recently used entry goes
    assert c._bytes <= c.max_bytes and not any(k[-1] == "u" for k in c._d)
    os.remove(p)
    time.sleep(0.01)
    h5io.append_datasets(p, [("u", a * 2), ("v", a)])                               # same path, new contents
    assert np.array_equal(c.get(p, "u"), a * 2)
    assert all(k[:3] == c._file_id(p) for k in c._d)                                # nothing of the old file is left


@pytest.mark.parametrize("n_parallel,prefetch,pinned", [(None, None, None), (1, 0, False), (3, 2, True), (2, 1, True), (0, 3, False)])
def test_prefetching_loader_yields_the_same_batches(n_parallel, prefetch, pinned):
    """The producer thread, the per-batch worker pool (`n_parallel`, PatchHandler3D.py:32) and the pinned staging ring change WHEN
    a batch is assembled, never its contents or order: every configuration yields the synchronous loader's batches bit for bit
    (each batch is copied when received -- with pinned=True a batch is only valid until two more have been requested)."""
    idx = data.load_indexes(os.path.join(DATA, "validate.csv"))
    ref = [tuple(np.array(a) for a in b) for b in
           data.PatchHandler3D(DATA, 16, 2, 4, 0.6).initialize_dataset(idx, shuffle=True, seed=5, shard=(0, 1), n_parallel=1, prefetch=0)]
    ds = data.PatchHandler3D(DATA, 16, 2, 4, 0.6).initialize_dataset(idx, shuffle=True, seed=5, shard=(0, 1), n_parallel=n_parallel,
                                                                     prefetch=prefetch, pinned=pinned)
    for epoch in range(2):
        got = [tuple(np.array(a) for a in b) for b in ds]
        if epoch == 0:
            assert len(got) == len(ref) == 3
            for g, r in zip(got, ref):
                assert all(np.array_equal(x, y) and x.dtype == y.dtype for x, y in zip(g, r))
    import threading
    it = iter(ds)                                   # an abandoned iterator must not leave its producer thread behind
    next(it)
    it.close()
    assert not any(t.name == "fdn-loader" and t.is_alive() for t in threading.enumerate())


def test_pinned_ring_keeps_a_batch_until_two_more_were_requested():
    """The documented contract of the pinned staging ring (prefetch + 3 slots): the consumer may still hold batch k - 1 (a non-blocking
    copy in flight) while it works on batch k, whatever the producer does meanwhile.  The dataset is walked several times over so that
    the ring wraps; the producer gets time to run ahead before every check."""
    import time
    idx = data.load_indexes(os.path.join(DATA, "validate.csv"))
    idx = np.concatenate([idx] * 4, axis=0)
    ref = [tuple(np.array(a) for a in b) for b in
           data.PatchHandler3D(DATA, 16, 2, 2, 0.6).initialize_dataset(idx, shuffle=False, shard=(0, 1), n_parallel=1, prefetch=0)]
    ds = data.PatchHandler3D(DATA, 16, 2, 2, 0.6).initialize_dataset(idx, shuffle=False, shard=(0, 1), n_parallel=2, prefetch=1, pinned=True)
    held = []
    for k, b in enumerate(ds):
        held.append(b)
        time.sleep(0.05)                            # the producer fills every slot it may
        for j in (k - 1, k):                        # the previous batch and this one are both intact
            if j >= 0:
                assert all(np.array_equal(x, y) for x, y in zip(held[j], ref[j])), (k, j)
    assert len(held) == len(ref) >= 8


def test_loader_errors_surface_in_the_consumer():
    idx = data.load_indexes(os.path.join(DATA, "validate.csv")).copy()
    idx[1, 0] = "no_such_file.h5"
    ds = data.PatchHandler3D(DATA, 16, 2, 4, 0.6).initialize_dataset(idx, shuffle=False, shard=(0, 1), prefetch=2)
    with pytest.raises(Exception):
        list(ds)
