"""The built-in HDF5 reader against a Keras-layout weights file it did NOT write: tests/golden/keras_layout_weights.h5 comes from real
h5py (tests/golden/make_golden_keras_h5.py, run with the image's conda python), laid out as tf.keras' model.save() lays out the
functional model of SR4DFlowNet.py (TrainerController.py:347-363; read back by TrainerController.py:394 and predictor.py:61):
weightless layers in `layer_names`, empty `weight_names`, contiguous datasets `<layer>/<layer>/kernel:0`, root `model_config`."""
import importlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "keras_layout_weights.h5")
fdn = importlib.import_module("4dflownet_amd")
h5io = importlib.import_module("4dflownet_amd.h5io")
network = importlib.import_module("4dflownet_amd.network")


def _spec():
    spec = importlib.util.spec_from_file_location("make_golden_keras_h5_values", os.path.join(HERE, "golden", "make_golden_keras_h5.py"))
    src = open(spec.origin).read().replace("import h5py\n", "")          # the value formulas only; h5py is absent from this interpreter
    ns = {"__file__": spec.origin}
    exec(compile(src, spec.origin, "exec"), ns)
    return ns


def test_reader_takes_a_file_written_by_h5py_in_keras_layout():
    ns = _spec()
    assert [s[:1] + s[1:] for s in ns["SPECS"]] == network.layer_specs(0, 0)        # the fixture's network is the package's
    got = h5io.read_keras_weights(PATH)
    assert sorted(got) == sorted(s[0] for s in ns["SPECS"])                         # weightless layers are skipped, nothing else is
    for li, (name, k, ci, co, ub) in enumerate(ns["SPECS"]):
        kern, bias = got[name]
        assert kern.dtype == np.float32 and kern.shape == (k, k, k, ci, co)
        assert np.array_equal(kern, ns["kernel_values"](li, (k, k, k, ci, co))), name
        assert (bias is None) == (not ub)
        if ub:
            assert np.array_equal(bias, ns["bias_values"](li, co)), name
    # the reader walks the group tree (attribute messages -- layer_names, weight_names, model_config: variable-length strings as h5py
    # writes them -- are skipped, not parsed): the weightless layers are there as empty groups, (h5 groups iterate by name)
    with h5io.open_read(PATH) as f:
        names = list(f["model_weights"].keys())
        assert "input_1" in names and "concatenate" in names and "tf_op_layer_Square" in names and len(names) == len(ns["SPECS"]) + 11
        assert list(f["model_weights"]["input_1"].keys()) == []


@pytest.mark.gpu
def test_model_loads_the_keras_layout_file():
    import torch
    ns = _spec()
    m = network.FlowNetModel(2, low_resblock=0, hi_resblock=0, seed=3)
    before = m.flat_w.clone()
    m.load_weights(PATH)
    assert not torch.equal(before, m.flat_w)
    for li, (L, (name, k, ci, co, ub)) in enumerate(zip(m.layers, ns["SPECS"])):
        assert L.name == name
        assert np.array_equal(L.w.cpu().numpy(), ns["kernel_values"](li, (k, k, k, ci, co)))
        if ub:
            assert np.array_equal(L.b.cpu().numpy(), ns["bias_values"](li, co))
    g = torch.Generator(device="cuda").manual_seed(1)
    pred = m.forward([torch.rand((1, 8, 8, 8, 1), device="cuda", generator=g) for _ in range(6)])
    assert pred.shape == (1, 16, 16, 16, 3) and bool(torch.isfinite(pred).all())
