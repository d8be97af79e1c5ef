"""On-device input pipeline == host loader (== reference loader, see test_data_golden.py), bit for bit."""
import importlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")
data = importlib.import_module("4dflownet_amd.data")
ddev = importlib.import_module("4dflownet_amd.data_device")
trainer = importlib.import_module("4dflownet_amd.trainer")


@pytest.mark.parametrize("P,R,thr", [(16, 2, 0.6), (16, 1, 0.6), (12, 2, 0.3)])
def test_device_batches_equal_host_batches(P, R, thr):
    idx = data.load_indexes(os.path.join(DATA, "train.csv"))          # covers un-rotated and all 9 (plane, degree) rotations
    host = data.PatchHandler3D(DATA, P, R, 7, thr)
    dev = ddev.DevicePatchHandler3D(DATA, P, R, 7, thr)
    hb = list(host.initialize_dataset(idx, shuffle=True, seed=4, shard=(0, 1)))
    db = list(dev.initialize_dataset(idx, shuffle=True, seed=4, shard=(0, 1)))
    assert len(hb) == len(db) == 8
    names = ["u", "v", "w", "u_mag", "v_mag", "w_mag", "u_hr", "v_hr", "w_hr", "venc", "mask"]
    for h, d in zip(hb, db):
        for n, a, t in zip(names, h, d):
            assert t.is_cuda and tuple(t.shape) == a.shape, (n, t.shape, a.shape)
            assert np.array_equal(t.cpu().numpy(), a), n           # includes -0.0 == 0.0; check the bits too:
            assert t.cpu().numpy().tobytes() == np.ascontiguousarray(a).tobytes(), n


def test_out_of_bounds_rows_are_rejected_and_training_consumes_device_batches():
    idx = data.load_indexes(os.path.join(DATA, "train.csv"))
    dev = ddev.DevicePatchHandler3D(DATA, 24, 2, 2, 0.6)               # shipped CSV is for patch 16: most rows overflow at 24
    bad = [r for r in idx if int(r[4]) + 24 > 38 or int(r[5]) + 24 > 36]
    with pytest.raises(ValueError):
        dev.load_batch_device(bad[:2])
    good = [r for r in idx if int(r[3]) + 24 <= 42 and int(r[4]) + 24 <= 38 and int(r[5]) + 24 <= 36][:4]
    assert len(good) == 4
    tc = trainer.TrainerController(24, 2, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=1, hi_resblock=1)
    ds = dev.initialize_dataset(np.array(good), shuffle=False, shard=(0, 1))
    losses = [tc.train_step(b) for b in ds]
    assert len(losses) == 2 and all(torch.isfinite(l).all() for l in losses)


@pytest.mark.parametrize("pinned", [True, False])
def test_prefetched_h2d_batches_equal_the_loaders_batches(pinned):
    """TrainerController.device_batches: batch k + 1 is copied on a copy stream (non-blocking from the pinned ring) while step k runs.
    Every yielded batch is on the device and bit-equal to what the loader produced, in order, ragged tail included; device-resident
    batches pass through; train steps fed this way give the losses of steps fed the plain batches."""
    idx = data.load_indexes(os.path.join(DATA, "train.csv"))
    P, R, B = 16, 2, 7
    host = data.PatchHandler3D(DATA, P, R, B, 0.6)
    plain = [tuple(np.array(a) for a in b) for b in host.initialize_dataset(idx, shuffle=True, seed=9, shard=(0, 1), pinned=False)]
    tc = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=1, hi_resblock=1, seed=3)
    got = []
    for b in tc.device_batches(host.initialize_dataset(idx, shuffle=True, seed=9, shard=(0, 1), pinned=pinned)):
        assert all(isinstance(t, torch.Tensor) and t.is_cuda for t in b)
        torch.cuda.current_stream().synchronize()
        got.append(tuple(t.cpu().numpy() for t in b))
        torch.empty(1 << 22, device="cuda").normal_()                  # keep the device busy between requests
    assert len(got) == len(plain) and len(plain[-1][0]) != B           # (the ragged tail is part of the comparison)
    for g, p_ in zip(got, plain):
        for a, b_ in zip(g, p_):
            assert a.tobytes() == np.ascontiguousarray(b_, dtype=np.float32).tobytes()
    dev = ddev.DevicePatchHandler3D(DATA, P, R, B, 0.6)
    db = list(dev.initialize_dataset(idx, shuffle=True, seed=9, shard=(0, 1)))
    assert all(x is y for bb, cc in zip(db, tc.device_batches(db)) for x, y in zip(bb, cc))
    l1 = [float(tc.train_step(b).sum()) for b in tc.device_batches(host.initialize_dataset(idx, shuffle=True, seed=9, shard=(0, 1), pinned=pinned))]
    tc2 = trainer.TrainerController(P, R, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=1, hi_resblock=1, seed=3)
    l2 = [float(tc2.train_step(b).sum()) for b in plain]
    assert l1 == l2
