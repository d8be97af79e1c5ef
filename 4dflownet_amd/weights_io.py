"""Model weight files.  `.npz` (native) and Keras-HDF5 `.h5` (model_weights/<layer>/<layer>/{kernel:0,bias:0},
layer names conv3d ... conv3d_35 in creation order -- what model.save()/load_weights() use in
TrainerController.py:356,394 and predictor.py:61)."""
import numpy as np


def _named(model):
    out = []
    for L in model.layers:
        out.append(("%s/kernel:0" % L.name, L.w))
        if L.b is not None:
            out.append(("%s/bias:0" % L.name, L.b))
    return out


def save_model_weights(model, path):
    arrays = dict((n, t.detach().cpu().numpy()) for n, t in _named(model))
    if path.endswith(".npz"):
        np.savez(path, **arrays)
        return
    from . import h5io
    h5io.write_keras_weights(path, [(L.name, L.w.detach().cpu().numpy(), None if L.b is None else L.b.detach().cpu().numpy())
                                    for L in model.layers])


def load_model_weights(model, path):
    if path.endswith(".npz"):
        z = np.load(path)
        model.set_weights([z[n] for n, _ in _named(model)])
        return
    from . import h5io
    by_layer = h5io.read_keras_weights(path)
    arrays = []
    for L in model.layers:
        if L.name not in by_layer:
            raise KeyError("layer %s not found in %s" % (L.name, path))
        k, b = by_layer[L.name]
        arrays.append(k)
        if L.b is not None:
            arrays.append(b)
    model.set_weights(arrays)
