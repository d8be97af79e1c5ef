"""ONE-COMMAND TensorFlow pin for the arithmetic of the hot path.

TensorFlow is absent from the build container and the GPU box, so today the conv / upsample / loss / Adam arithmetic is
checked against a restatement only ("parity unpinned", oracle/flownet_oracle.py header).  The day a machine with
`tensorflow` (the reference names 2.2, README.md:4; any 2.x with tf.keras works) AND the reference checkout is at hand:

    cd <this repo> && REFERENCE=/path/to/4DFlowNet python tests/golden/make_golden_tf.py

imports the reference's OWN `Network.TrainerController` / `Network.SR4DFlowNet` / `Network.PatchHandler3D` (never copied
here), sets the seeded weights of SURVEY 8(d) by LAYER NAME (conv3d, conv3d_1, ... = creation order), feeds loader batches
of the reference's data/example_data*.h5 rows, and writes tests/golden/tf_golden.npz with, per case:
    inputs (the 11 loader arrays), the weights that were set, pred = model(inputs), the (B,) loss vector incl. L2, mse,
    rel-error, tape.gradient(loss, trainable_variables) (all arrays, in Keras trainable_variables order + their names),
    the weights after ONE optimizer.apply_gradients step, and optimizer.get_weights() after it (iterations, m..., v...);
    beside the .npz, per case: tf_model_c<i>.h5 = model.save() and tf_optimizer_c<i>.pkl = pickle of optimizer.get_weights(), the two
    checkpoint files of TrainerController.py:347-363 as real Keras writes them.
tests/test_tf_golden.py consumes that file when present (CPU: the oracle against it; GPU: the HIP path against it) and
skips with "parity unpinned" when absent.  Only data leaves this script: inputs and outputs, no reference source.

Cases: (P=8,  R=2, LB=1, HB=1, B=2) and (P=16, R=2, LB=1, HB=1, B=2), rows of data/train.csv that fit the patch size."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REFERENCE", "/root/reference")
CASES = [dict(P=8, R=2, LB=1, HB=1, B=2), dict(P=16, R=2, LB=1, HB=1, B=2)]
LR = 1e-4


def layer_names(LB, HB):
    """Keras default names in creation order (SR4DFlowNet.py:17-46): conv3d, conv3d_1, ..."""
    n = 6 + 2 * (LB + HB) + 6
    return ["conv3d" if i == 0 else "conv3d_%d" % i for i in range(n)]


def seeded_weights(LB, HB):
    """Glorot-uniform from default_rng(0), drawn layer by layer in creation order; biases 0.  Must equal
    oracle.flownet_oracle.init_params(0, LB, HB) -- asserted below so the two cannot drift apart."""
    sys.path.insert(0, ROOT)
    from oracle import flownet_oracle as O
    return O.init_params(0, LB, HB, np.float32)


def main():
    import tensorflow as tf                                   # the real one -- this script is useless without it
    sys.path.insert(0, os.path.join(REF, "src"))
    os.chdir(os.path.join(REF, "src"))                        # TrainerController writes ../models relative to src/
    from Network.PatchHandler3D import PatchHandler3D
    from Network.TrainerController import TrainerController
    out = {"tf_version": np.asarray(tf.__version__), "cases": np.asarray(len(CASES))}
    data_dir = os.path.join(REF, "data")
    rows = np.genfromtxt(os.path.join(data_dir, "train.csv"), delimiter=",", skip_header=True, dtype="unicode")   # trainer.py:9
    for ci, c in enumerate(CASES):
        P, R, LB, HB, B = c["P"], c["R"], c["LB"], c["HB"], c["B"]
        tf.keras.backend.clear_session()                      # layer-name counters restart at conv3d
        tc = TrainerController(P, R, initial_learning_rate=LR, quicksave_enable=False, network_name="golden",
                               low_resblock=LB, hi_resblock=HB)
        params = seeded_weights(LB, HB)
        names = layer_names(LB, HB)
        assert [p["name"] for p in params] == names
        for p in params:
            layer = tc.model.get_layer(p["name"])
            cur = layer.get_weights()
            new = [p["w"]] + ([p["b"]] if p["b"] is not None else [])
            assert [a.shape for a in cur] == [a.shape for a in new], (p["name"], [a.shape for a in cur])
            layer.set_weights(new)
        # loader batch: the first B rows whose patch fits (PatchHandler3D.py:49-160 does the slicing / rotation)
        ph = PatchHandler3D(data_dir, P, R, B, mask_threshold=0.6)
        ds = ph.initialize_dataset(rows[:B], shuffle=False, n_parallel=None)
        batch = next(iter(ds))
        u, v, w, u_mag, v_mag, w_mag, u_hr, v_hr, w_hr, venc, mask = batch
        hires = tf.concat((u_hr, v_hr, w_hr), axis=-1)
        tvars = tc.model.trainable_variables
        with tf.GradientTape() as tape:                       # TrainerController.py:213-219, un-jitted
            pred = tc.model([u, v, w, u_mag, v_mag, w_mag], training=True)
            loss_only, mse, divloss = tc.loss_function(hires, pred, mask)
            rel = tc.accuracy_function(hires, pred, mask)
            l2 = tc.calculate_regularizer_loss()
            loss = loss_only + l2                              # :245-249
        grads = tape.gradient(loss, tvars)                    # :223
        tc.optimizer.apply_gradients(zip(grads, tvars))       # :225
        k = "c%d_" % ci
        out[k + "cfg"] = np.asarray([P, R, LB, HB, B])
        for n_, a in zip(("u", "v", "w", "u_mag", "v_mag", "w_mag", "u_hr", "v_hr", "w_hr", "venc", "mask"), batch):
            out[k + "in_" + n_] = a.numpy()
        out[k + "pred"] = pred.numpy()
        out[k + "loss"] = loss.numpy()
        out[k + "mse"] = mse.numpy()
        out[k + "rel_err"] = rel.numpy()
        out[k + "l2"] = np.asarray(l2.numpy())
        out[k + "tv_names"] = np.asarray([t.name for t in tvars])
        for i, (g, t) in enumerate(zip(grads, tvars)):
            out[k + "grad_%03d" % i] = g.numpy()
            out[k + "wnew_%03d" % i] = t.numpy()
        ow = tc.optimizer.get_weights()                       # [iterations, m..., v...] -- settles the optimizer.pkl order
        out[k + "opt_n"] = np.asarray(len(ow))
        for i, a in enumerate(ow):
            out[k + "opt_%03d" % i] = np.asarray(a)
        out[k + "opt_names"] = np.asarray([w_.name for w_ in tc.optimizer.weights])
        # the two files the reference itself writes at a checkpoint (TrainerController.py:347-363): model.save()'s HDF5 and the pickled
        # optimizer.get_weights() -- real Keras files for the built-in reader / restore path (tests/test_tf_golden.py; f1 of SURVEY 8)
        import pickle
        tc.model.save(os.path.join(HERE, "tf_model_c%d.h5" % ci))
        with open(os.path.join(HERE, "tf_optimizer_c%d.pkl" % ci), "wb") as f:
            pickle.dump(tc.optimizer.get_weights(), f)
    dst = os.path.join(HERE, "tf_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, "(%d arrays, TensorFlow %s)" % (len(out), tf.__version__))


if __name__ == "__main__":
    try:
        import tensorflow  # noqa: F401
    except ImportError:
        sys.exit("make_golden_tf.py needs a real `tensorflow` (absent in this container): parity stays unpinned until it runs")
    main()
