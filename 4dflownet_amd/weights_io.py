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
    names = [L.name for L in model.layers]
    if not all(n in by_layer for n in names):
        # Keras' load_weights matches layers by position, not by name: a model built second in a process is saved as
        # conv3d_36 ... conv3d_71.  Fall back to the file's conv layers in creation order (numeric suffix) -- the shapes
        # are then checked one by one by set_weights.
        conv = sorted((n for n in by_layer if _suffix(n) is not None), key=_suffix)
        if len(conv) != len(names):
            raise KeyError("%s holds %d conv3d layers (%s ...), the model has %d: cannot match by name or by position"
                           % (path, len(conv), ", ".join(conv[:3]), len(names)))
        names = conv
    arrays = []
    for L, n in zip(model.layers, names):
        k, b = by_layer[n]
        if (b is None) != (L.b is None):
            raise ValueError("%s: layer %s %s a bias but the model's %s %s" % (path, n, "has" if b is not None else "lacks", L.name,
                                                                               "has none" if L.b is None else "expects one"))
        arrays.append(k)
        if L.b is not None:
            arrays.append(b)
    model.set_weights(arrays)


def _suffix(name):
    """conv3d -> 0, conv3d_<k> -> k, anything else -> None."""
    if name == "conv3d":
        return 0
    if name.startswith("conv3d_") and name[7:].isdigit():
        return int(name[7:])
    return None
