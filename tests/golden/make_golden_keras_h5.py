"""A Keras-layout weights file written by REAL h5py, laid out as tf.keras' `model.save(path)` lays out a functional model
(`keras.engine.saving.save_model_to_hdf5` -> `save_weights_to_hdf5_group`; what TrainerController.py:347-363 writes and
TrainerController.py:394 / predictor.py:61 `load_weights`): fixture for the built-in HDF5 reader (4dflownet_amd/h5io.py), which
otherwise only ever sees files this package wrote itself.

    /opt/conda/bin/python3.9 tests/golden/make_golden_keras_h5.py      (h5py lives in the conda python of the build image)

writes tests/golden/keras_layout_weights.h5:
  * root attributes `keras_version`, `backend`, `model_config` (a JSON string), as model.save() sets them;
  * group `model_weights` with attributes `layer_names` (EVERY layer of the functional graph in `model.layers` order, the weightless
    ones -- InputLayer, TensorFlowOpLayer, Concatenate, LeakyReLU, Add, Lambda -- included), `backend`, `keras_version`;
  * one group per layer with attribute `weight_names` (empty array for weightless layers) and the datasets
    `<layer>/<layer>/kernel:0` (+ `bias:0`), contiguous float32, no chunking / compression -- h5py's defaults, as Keras uses them.
Network: low_resblock = 0, hi_resblock = 0 (6 stem + 6 head convolutions = SR4DFlowNet.py:17-25,39-46).  Values: a closed-form
integer pattern (exact in fp32, compresses well in git): w.flat[i] = ((7 i + 13 layer_index) mod 251 - 125) / 1024, biases
((3 i + layer_index) mod 17 - 8) / 64 -- the test recomputes them, no second fixture needed."""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SPECS = [("conv3d", 3, 3, 64, True), ("conv3d_1", 3, 64, 64, True), ("conv3d_2", 3, 3, 64, True), ("conv3d_3", 3, 64, 64, True),
         ("conv3d_4", 1, 128, 64, True), ("conv3d_5", 3, 64, 64, True)] + \
        [s for i in range(3) for s in (("conv3d_%d" % (6 + 2 * i), 3, 64, 64, True), ("conv3d_%d" % (7 + 2 * i), 3, 64, 1, True))]
# model.layers of the functional graph, weightless layers included (names as tf.keras 2.2 generates them)
WEIGHTLESS_BEFORE = ["input_1", "input_2", "input_3", "input_4", "input_5", "input_6", "tf_op_layer_Square", "tf_op_layer_Sqrt", "concatenate"]
WEIGHTLESS_AFTER = ["concatenate_1", "lambda", "concatenate_2"]


def kernel_values(li, shape):
    n = int(np.prod(shape))
    return (((7 * np.arange(n, dtype=np.int64) + 13 * li) % 251 - 125) / 1024.0).astype(np.float32).reshape(shape)


def bias_values(li, cout):
    return (((3 * np.arange(cout, dtype=np.int64) + li) % 17 - 8) / 64.0).astype(np.float32)


def main():
    dst = os.path.join(HERE, "keras_layout_weights.h5")
    layer_names = WEIGHTLESS_BEFORE[:7] + ["conv3d_2", "conv3d", "conv3d_3", "conv3d_1", "concatenate", "conv3d_4", "conv3d_5"] + \
        ["conv3d_6", "conv3d_8", "conv3d_10", "conv3d_7", "conv3d_9", "conv3d_11"] + WEIGHTLESS_AFTER
    with h5py.File(dst, "w") as f:
        f.attrs["keras_version"] = b"2.3.0-tf"
        f.attrs["backend"] = b"tensorflow"
        f.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {"name": "model", "layers": [{"name": n} for n in layer_names]}}).encode()
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = [n.encode("utf8") for n in layer_names]
        g.attrs["backend"] = b"tensorflow"
        g.attrs["keras_version"] = b"2.3.0-tf"
        by_name = dict((s[0], (i, s)) for i, s in enumerate(SPECS))
        for n in layer_names:
            lg = g.create_group(n)
            if n not in by_name:
                lg.attrs["weight_names"] = np.zeros((0,), dtype="S1")       # Keras: an empty weight_names attribute
                continue
            li, (_, k, ci, co, ub) = by_name[n]
            names = ["%s/kernel:0" % n] + (["%s/bias:0" % n] if ub else [])
            lg.attrs["weight_names"] = [w.encode("utf8") for w in names]
            lg.create_dataset("%s/kernel:0" % n, data=kernel_values(li, (k, k, k, ci, co)))
            if ub:
                lg.create_dataset("%s/bias:0" % n, data=bias_values(li, co))
    print("wrote", dst, os.path.getsize(dst), "bytes, h5py", h5py.__version__)


if __name__ == "__main__":
    main()
