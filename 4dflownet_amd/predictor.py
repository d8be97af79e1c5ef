"""Full-volume sliding-window inference, mirroring src/predictor.py: patchify -> batched forward -> stitch ->
de-normalise -> zero sub-pixel velocities -> append u,v,w (+dx/res_increase) to the output HDF5.

With torch.distributed initialised (BASELINE cfg5) the patch list is split contiguously across ranks; every rank
runs its share through the HIP forward and the HR patches are all-gathered so rank 0 stitches and writes."""
import os
import time

import numpy as np
import torch

from . import h5io, parallel
from .data import ImageDataset
from .network import Input, SR4DFlowNet
from .tiler import PatchGenerator


def prepare_network(patch_size, res_increase, low_resblock, hi_resblock, device=None, dtype="float32"):
    """predictor.py:11-29.  dtype="bfloat16" runs the forward with bf16 activation storage (fp32 weights / prediction)."""
    shape = (patch_size, patch_size, patch_size, 1)
    ins = [Input(shape, n) for n in ('u', 'v', 'w', 'u_mag', 'v_mag', 'w_mag')]
    return SR4DFlowNet(res_increase).build_network(*ins, low_resblock, hi_resblock, device=device, dtype=dtype)


def save_to_h5(output_filepath, col_name, dataset, compression=None):
    """src/utils/prediction_utils.py:15-28."""
    h5io.append_dataset(output_filepath, col_name, dataset, compression=compression)


def _predict_pipelined(network, velocities, magnitudes, batch_size, lo, hi):
    """Single-process path: the float64 conversion runs on the device, every batch's result travels to a pinned staging buffer on
    a copy stream while the next batch computes, and the host moves it into the result array in that shadow -- the loop costs the
    forward time only."""
    S = velocities[0].shape[1] * network.res_increase
    res = np.empty((hi - lo, S, S, S, 3), dtype=np.float64)
    key = (batch_size, S)
    st = getattr(network, "_predict_stage", None)
    if st is None or st[0] != key:
        st = (key, [torch.empty((batch_size, S, S, S, 3), dtype=torch.float64).pin_memory() for _ in range(2)],
              torch.cuda.Stream(device=network.device))
        network._predict_stage = st
    _, stage, copy_stream = st
    main = torch.cuda.current_stream(network.device)
    inflight = [None, None]                      # per staging slot: (event, first row, row count, device tensor kept alive)

    def drain(slot):
        if inflight[slot] is not None:
            ev, r0, cnt, _keep = inflight[slot]
            ev.synchronize()
            res[r0:r0 + cnt] = stage[slot][:cnt].numpy()
            inflight[slot] = None

    for k, s in enumerate(range(lo, hi, batch_size)):
        e = min(s + batch_size, hi)
        ins = [velocities[i][s:e] for i in range(3)] + [magnitudes[i][s:e] for i in range(3)]
        out64 = network.forward(ins).double()
        slot = k & 1
        drain(slot)                              # the copy issued two batches ago has long finished; frees the staging slot
        copy_stream.wait_stream(main)
        with torch.cuda.stream(copy_stream):
            stage[slot][:e - s].copy_(out64, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        out64.record_stream(copy_stream)
        inflight[slot] = (ev, s - lo, e - s, out64)
        drain(slot ^ 1)                          # the previous batch's rows, while this batch computes
    drain(0); drain(1)
    return res


def predict_patches(network, velocities, magnitudes, batch_size):
    """The batched predict loop of predictor.py:79-94 (results accumulate in float64 like np.zeros + np.append
    there), with the patch list sharded over ranks when running data-parallel."""
    n = len(velocities[0])
    world, rank = parallel.world_size(), parallel.rank()
    if world == 1:
        return _predict_pipelined(network, velocities, magnitudes, batch_size, 0, n)
    per = (n + world - 1) // world
    lo, hi = min(rank * per, n), min((rank + 1) * per, n)
    outs = []
    for s in range(lo, hi, batch_size):
        e = min(s + batch_size, hi)
        ins = [velocities[i][s:e] for i in range(3)] + [magnitudes[i][s:e] for i in range(3)]
        outs.append(network.forward(ins))
    S = velocities[0].shape[1] * network.res_increase
    mine = torch.cat(outs, 0) if outs else torch.zeros((0, S, S, S, 3), device=network.device)
    pad = torch.zeros((per, S, S, S, 3), device=network.device)
    pad[:mine.shape[0]] = mine
    gathered = parallel.all_gather_equal(pad)
    mine = torch.cat([g[:max(0, min((r + 1) * per, n) - min(r * per, n))] for r, g in enumerate(gathered)], 0)
    return mine.cpu().numpy().astype(np.float64)


def predict_file(network, input_filepath, output_filepath, patch_size, res_increase, batch_size=8,
                 round_small_values=True, verbose=True):
    """predictor.py:67-115 for every row of the input file.  Returns the list of (u,v,w) volumes written."""
    pgen = PatchGenerator(patch_size, res_increase)
    dataset = ImageDataset()
    nr_rows = dataset.get_dataset_len(input_filepath)
    is0 = parallel.rank() == 0
    written = []
    for nrow in range(nr_rows):
        dataset.load_vectorfield(input_filepath, nrow)
        velocities, magnitudes = pgen.patchify(dataset)
        t0 = time.time()
        results = predict_patches(network, velocities, magnitudes, batch_size)
        if verbose and is0:
            print("Processed row %d/%d: %d patches in %.2f secs." % (nrow + 1, nr_rows, len(results), time.time() - t0))
        vols, cols = [], []
        for i in range(3):
            v = pgen._patchup_with_overlap(results[:, :, :, :, i], pgen.nr_x, pgen.nr_y, pgen.nr_z)
            v = v * dataset.venc                                   # de-normalise (:103)
            if round_small_values:
                v[np.abs(v) < dataset.velocity_per_px] = 0        # (:104-107)
            v = np.expand_dims(v, axis=0)
            vols.append(v)
            cols.append((dataset.velocity_colnames[i], v))
        if dataset.dx is not None:
            cols.append((dataset.dx_colname, np.expand_dims(dataset.dx / res_increase, axis=0)))
        if is0:
            h5io.append_datasets(output_filepath, cols, compression='gzip')     # u, v, w (+ dx/R): one pass over the file
        written.append(tuple(vols))
    return written


def main(data_dir='../data', filename='example_data.h5', output_dir="../result", output_filename='example_result.h5',
         model_path="../models/4DFlowNet/4DFlowNet.h5", patch_size=24, res_increase=2, batch_size=8,
         round_small_values=True, low_resblock=8, hi_resblock=4, dtype="float32"):
    """Same hard-coded surface as predictor.py:31-47 (+ dtype)."""
    parallel.init_from_env()
    network = prepare_network(patch_size, res_increase, low_resblock, hi_resblock, dtype=dtype)
    network.load_weights(model_path)
    if not os.path.isdir(output_dir) and parallel.rank() == 0:
        os.makedirs(output_dir)
    predict_file(network, '{}/{}'.format(data_dir, filename), '{}/{}'.format(output_dir, output_filename), patch_size,
                 res_increase, batch_size, round_small_values)
    if parallel.rank() == 0:
        print("Done!")


if __name__ == '__main__':
    main()
