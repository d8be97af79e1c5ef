"""CPU checks of the integer tricks the fp32 kernels rely on (restated from 4dflownet_amd/csrc/fdn_common.h and
conv64_mfma.hip): multiply-shift division with host-made magics, and the bank layout of the padded LDS rows."""
import numpy as np


def test_magic20_division_is_exact_on_its_domain():
    # fdn_magic20 / fdn_div20: floor(r/d) for 0 <= r < 1024, 1 <= d <= 1024 with magic = ceil(2^20/d)
    r = np.arange(1024, dtype=np.uint64)
    for d in range(1, 1025):
        magic = ((1 << 20) + d - 1) // d
        assert np.array_equal((r * np.uint64(magic)) >> np.uint64(20), r // np.uint64(d)), d


def test_magic40_division_is_exact_for_block_indices():
    # fdn_magic40 / fdn_udiv40: M = ceil(2^40/d) = hi*2^32 + lo;  floor(n/d) = (n*hi + mulhi32(n, lo)) >> 8  for n*d < 2^40
    rng = np.random.default_rng(0)
    ds = np.unique(np.concatenate([np.arange(1, 300), rng.integers(1, 1 << 17, 400), [1 << 17, (1 << 17) - 1]]))
    for d in ds:
        d = int(d)
        m = ((1 << 40) + d - 1) // d
        hi, lo = m >> 32, m & 0xFFFFFFFF
        nmax = min(1 << 20, (1 << 40) // d)
        n = np.unique(np.concatenate([rng.integers(0, nmax, 300), np.arange(min(nmax, 64)), [nmax - 1],
                                      np.clip(np.arange(1, 40) * d - 1, 0, nmax - 1), np.clip(np.arange(1, 40) * d, 0, nmax - 1)]))
        n = n.astype(np.uint64)
        q = (n * np.uint64(hi) + ((n * np.uint64(lo)) >> np.uint64(32))) >> np.uint64(8)
        assert np.all(n * np.uint64(hi) + ((n * np.uint64(lo)) >> np.uint64(32)) < (1 << 32))       # fits the 32-bit add
        assert np.array_equal(q, n // np.uint64(d)), d


def test_padded_lds_rows_are_conflict_free_for_ds_read_b128():
    # MI355X_MICROARCH.md: ds_read_b128 is served in four groups of 16 lanes, 64 banks of 4 B = a 256-B window per group.
    # Lane i of an A-fragment read addresses row (base + i) at the same in-row offset; with a row stride of 144 B (cin halves,
    # 128 B + 16 B pad) or 80 B (cin quarters) the 16 rows of a group must fall on 16 different 16-B slots of the window.
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for stride in (144, 80):
        for base in range(0, 64):
            for off in range(0, stride - 16, 16):
                for g in groups:
                    slots = {((base + i) * stride + off) % 256 // 16 for i in g}
                    assert len(slots) == 16, (stride, base, off)
    # the unpadded 128-B stride is NOT conflict-free (that is why the first version needed an XOR swizzle)
    assert len({(i * 128) % 256 // 16 for i in groups[0]}) < 16


def test_trainer_script_picks_the_host_loader_only_on_request(monkeypatch):
    """scripts/trainer.py feeds train_network from the on-device loader (the measured path) unless FDN_HOST_LOADER is set, in which
    case it builds data.PatchHandler3D and asks for the pinned staging ring.  No GPU needed: the device class is
    stubbed, only the selection logic runs."""
    import importlib
    import importlib.util
    import os
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fdn_trainer_script", os.path.join(root, "scripts", "trainer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                       # __main__ guard: nothing runs
    made = []

    class _Dev:
        def __init__(self, *a):
            made.append(a)
    stub = types.ModuleType("4dflownet_amd.data_device")
    stub.DevicePatchHandler3D = _Dev
    monkeypatch.setitem(sys.modules, "4dflownet_amd.data_device", stub)
    monkeypatch.delenv("FDN_HOST_LOADER", raising=False)
    h, kw = mod.make_handler("/d", 16, 2, 20, 0.6)
    assert isinstance(h, _Dev) and kw == {} and made == [("/d", 16, 2, 20, 0.6)]
    monkeypatch.setenv("FDN_HOST_LOADER", "1")
    h, kw = mod.make_handler("/d", 16, 2, 20, 0.6)
    data = importlib.import_module("4dflownet_amd.data")
    assert type(h) is data.PatchHandler3D and kw == {"pinned": True}
    monkeypatch.setenv("FDN_HOST_LOADER", "0")
    assert isinstance(mod.make_handler("/d", 16, 2, 20, 0.6)[0], _Dev)
