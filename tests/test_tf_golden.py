"""Consumes tests/golden/tf_golden.npz -- outputs of the REFERENCE ITSELF under a real TensorFlow, written by
tests/golden/make_golden_tf.py -- when it exists: the oracle (CPU) and the HIP path (GPU) must reproduce the reference's
prediction, loss, gradients and first Adam step within the north_star tolerance (1e-3 relative fp32).  TensorFlow is absent
from the build container and the GPU box, so until somebody runs that script elsewhere these tests SKIP and the arithmetic
stays "parity unpinned" (oracle/flownet_oracle.py header)."""
import importlib
import os

import numpy as np
import pytest

from oracle import flownet_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
NPZ = os.path.join(HERE, "golden", "tf_golden.npz")
TOL = 1e-3

needs_npz = pytest.mark.skipif(not os.path.exists(NPZ), reason="parity unpinned: tests/golden/tf_golden.npz absent (run "
                               "tests/golden/make_golden_tf.py where tensorflow + the reference are available)")


def load_cases():
    z = np.load(NPZ, allow_pickle=False)
    cases = []
    for ci in range(int(z["cases"])):
        k = "c%d_" % ci
        P, R, LB, HB, B = [int(v) for v in z[k + "cfg"]]
        names = [str(s) for s in z[k + "tv_names"]]
        c = dict(P=P, R=R, LB=LB, HB=HB, B=B, names=names,
                 batch=tuple(z[k + "in_" + n] for n in ("u", "v", "w", "u_mag", "v_mag", "w_mag", "u_hr", "v_hr", "w_hr", "venc", "mask")),
                 pred=z[k + "pred"], loss=z[k + "loss"], mse=z[k + "mse"], rel_err=z[k + "rel_err"],
                 grads=[z[k + "grad_%03d" % i] for i in range(len(names))], wnew=[z[k + "wnew_%03d" % i] for i in range(len(names))])
        cases.append(c)
    return cases


def by_creation_order(names, arrays):
    """Keras trainable_variables order (depth-sorted layers for a functional model) -> our flat order (layer creation order,
    kernel then bias).  Variable names look like 'conv3d_7/kernel:0'."""
    def key(n):
        layer, var = n.split("/")[0], n.split("/")[1]
        idx = 0 if layer == "conv3d" else int(layer.split("_")[1])
        return (idx, 0 if var.startswith("kernel") else 1)
    order = sorted(range(len(names)), key=lambda i: key(names[i]))
    return [arrays[i] for i in order]


def rel(a, b):
    return float(np.linalg.norm((np.asarray(a, np.float64) - b).ravel()) / max(np.linalg.norm(np.asarray(b, np.float64).ravel()), 1e-30))


def test_generator_script_is_importable_and_names_match_the_oracle():
    """The generator must keep working the day it is needed: it parses, and its layer-name rule equals the oracle's."""
    import ast
    src = open(os.path.join(HERE, "golden", "make_golden_tf.py")).read()
    ast.parse(src)
    ns = {"__file__": os.path.join(HERE, "golden", "make_golden_tf.py"), "__name__": "make_golden_tf_head"}
    exec(compile(src.split("def seeded_weights")[0], "make_golden_tf_head", "exec"), ns)
    for LB, HB in ((1, 1), (8, 4)):
        assert ns["layer_names"](LB, HB) == [p["name"] for p in O.init_params(0, LB, HB)]


def test_keras_layer_order_is_a_permutation_that_only_swaps_equal_depth_layers():
    """network.keras_layer_order: the order Keras lists the conv layers in (model.layers / trainable_variables / optimizer slots)."""
    network = importlib.import_module("4dflownet_amd.network")
    assert network.keras_layer_order(8, 4) == [2, 0, 3, 1] + list(range(4, 30)) + [30, 32, 34, 31, 33, 35]
    for LB, HB in ((1, 1), (2, 0), (0, 1), (8, 4)):
        order = network.keras_layer_order(LB, HB)
        specs = network.layer_specs(LB, HB)
        assert sorted(order) == list(range(len(specs)))
        # layers that change places have identical (K, Cin, Cout, bias): the reason shapes cannot reveal a wrong order
        for pos, i in enumerate(order):
            assert specs[pos][1:] == specs[i][1:] or {pos, i} <= set(range(4)) | set(range(len(specs) - 6, len(specs)))


@needs_npz
def test_keras_variable_order_matches_real_tensorflow_names():
    """The day tf_golden.npz exists: the derived order must be the one a real TensorFlow reports (settles optimizer.pkl)."""
    network = importlib.import_module("4dflownet_amd.network")
    for c in load_cases():
        specs = network.layer_specs(c["LB"], c["HB"])
        expect = []
        for i in network.keras_layer_order(c["LB"], c["HB"]):
            expect.append("%s/kernel:0" % specs[i][0])
            if specs[i][4]:
                expect.append("%s/bias:0" % specs[i][0])
        assert c["names"] == expect, (c["names"][:8], expect[:8])


@needs_npz
def test_reader_takes_the_checkpoint_files_real_keras_wrote():
    """tf_model_c<i>.h5 (model.save) and tf_optimizer_c<i>.pkl (pickled optimizer.get_weights()), written beside the .npz: the built-in
    HDF5 reader returns the weights after the step, and the pickle's slot order is the Keras variable order the restore path assumes."""
    import pickle
    h5io = importlib.import_module("4dflownet_amd.h5io")
    for ci, c in enumerate(load_cases()):
        h5 = os.path.join(HERE, "golden", "tf_model_c%d.h5" % ci)
        pkl = os.path.join(HERE, "golden", "tf_optimizer_c%d.pkl" % ci)
        if not (os.path.exists(h5) and os.path.exists(pkl)):
            pytest.skip("tf_golden.npz predates the checkpoint files: re-run tests/golden/make_golden_tf.py")
        got = h5io.read_keras_weights(h5)
        for n, a in zip(c["names"], c["wnew"]):
            layer, var = n.split("/")
            k, b = got[layer]
            assert np.array_equal(k if var.startswith("kernel") else b, a), n
        ow = pickle.load(open(pkl, "rb"))
        nv = len(c["names"])
        assert len(ow) == 1 + 2 * nv and int(ow[0]) == 1
        assert [tuple(a.shape) for a in ow[1:1 + nv]] == [tuple(a.shape) for a in c["wnew"]]      # m slots in trainable_variables order


@needs_npz
def test_oracle_reproduces_the_tensorflow_reference():
    for c in load_cases():
        params = O.init_params(0, c["LB"], c["HB"], np.float64)
        batch = tuple(np.asarray(a, np.float64) for a in c["batch"])
        out = O.loss_and_grads(params, batch, c["R"], c["LB"], c["HB"], f32_coeffs=True)
        assert rel(out["pred"], c["pred"]) <= TOL
        assert np.abs(out["loss"] - c["loss"]).max() <= TOL * np.abs(c["loss"]).max()
        assert np.abs(out["rel_err"] - c["rel_err"]).max() <= TOL * max(np.abs(c["rel_err"]).max(), 1.0)
        g_ref = np.concatenate([np.asarray(a, np.float64).reshape(-1) for a in by_creation_order(c["names"], c["grads"])])
        assert rel(O.flatten(out["grads"]), g_ref) <= TOL
        state = {}
        O.train_step(params, state, batch, 1e-4, c["R"], c["LB"], c["HB"], f32_coeffs=True)
        w_ref = np.concatenate([np.asarray(a, np.float64).reshape(-1) for a in by_creation_order(c["names"], c["wnew"])])
        assert np.abs(O.flatten(params) - w_ref).max() <= 2.1e-4          # first Adam step: +-lr per element at most


@needs_npz
@pytest.mark.gpu
def test_hip_path_reproduces_the_tensorflow_reference():
    trainer = importlib.import_module("4dflownet_amd.trainer")
    for c in load_cases():
        tc = trainer.TrainerController(c["P"], c["R"], initial_learning_rate=1e-4, quicksave_enable=False,
                                       low_resblock=c["LB"], hi_resblock=c["HB"], seed=0)
        loss = tc.train_step(c["batch"])
        assert np.abs(loss.cpu().numpy() - c["loss"]).max() <= TOL * np.abs(c["loss"]).max()
        g_ref = np.concatenate([np.asarray(a, np.float64).reshape(-1) for a in by_creation_order(c["names"], c["grads"])])
        # the flat gradient buffer holds d(sum_b mse_b); the reference's gradient also carries B * 2 * lambda * w (L2)
        g = tc.model.flat_g.double().cpu().numpy()
        w0 = O.flatten(O.init_params(0, c["LB"], c["HB"], np.float64))
        isk = tc.model.is_kernel.cpu().numpy().astype(bool)
        g = g + np.where(isk, c["B"] * 2 * O.L2_LAMBDA * w0, 0.0)
        assert rel(g, g_ref) <= TOL
