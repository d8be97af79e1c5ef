"""Full-volume sliding-window inference, mirroring src/predictor.py: patchify -> batched forward -> stitch ->
de-normalise -> zero sub-pixel velocities -> append u,v,w (+dx/res_increase) to the output HDF5.

With torch.distributed initialised (BASELINE cfg5) the patch list is split contiguously across ranks; every rank
runs its share through the same pipelined HIP forward loop and sends exactly its rows to rank 0, which stitches and writes."""
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


def _stage(network, batch_size, S):
    """Two pinned float64 staging buffers + a copy stream, kept on the network between calls."""
    key = (batch_size, S)
    st = getattr(network, "_predict_stage", None)
    if st is None or st[0] != key:
        st = (key, [torch.empty((batch_size, S, S, S, 3), dtype=torch.float64).pin_memory() for _ in range(2)],
              torch.cuda.Stream(device=network.device))
        network._predict_stage = st
    return st[1], st[2]


def _drain_to_host(network, chunks, res, batch_size):
    """chunks: iterable of (first row, device tensor (cnt,S,S,S,3) fp32 or fp64) in production order.  Converts to float64 on the
    device, moves every chunk to a pinned staging buffer on a copy stream while the producer of the NEXT chunk runs, and copies it
    into `res` in that shadow: the loop costs the producer's time only."""
    S = res.shape[1]
    stage, copy_stream = _stage(network, batch_size, S)
    main = torch.cuda.current_stream(network.device)
    inflight = [None, None]                      # per staging slot: (event, first row, row count, device tensor kept alive)

    def drain(slot):
        if inflight[slot] is not None:
            ev, r0, cnt, _keep = inflight[slot]
            ev.synchronize()
            res[r0:r0 + cnt] = stage[slot][:cnt].numpy()
            inflight[slot] = None

    for k, (r0, out) in enumerate(chunks):
        out64 = out.double()
        cnt = out64.shape[0]
        slot = k & 1
        drain(slot)                              # the copy issued two chunks ago has long finished; frees the staging slot
        copy_stream.wait_stream(main)
        with torch.cuda.stream(copy_stream):
            stage[slot][:cnt].copy_(out64, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        out64.record_stream(copy_stream)
        inflight[slot] = (ev, r0, cnt, out64)
        drain(slot ^ 1)                          # the previous chunk's rows, while this one is produced
    drain(0); drain(1)
    return res


def _forward_chunks(network, velocities, magnitudes, batch_size, lo, hi, base=0, keep=None):
    """The batched forward loop of predictor.py:79-94 over patches [lo, hi): yields (row - base, prediction) per batch; with `keep`
    (a device buffer) every prediction is also stored at rows [row - lo, ...) of it."""
    for s in range(lo, hi, batch_size):
        e = min(s + batch_size, hi)
        ins = [velocities[i][s:e] for i in range(3)] + [magnitudes[i][s:e] for i in range(3)]
        out = network.forward(ins)
        if keep is not None:
            keep[s - lo:e - lo].copy_(out)
        yield s - base, out


def _predict_pipelined(network, velocities, magnitudes, batch_size, lo, hi):
    """Single-process path: forward + float64 conversion on the device, results to the host behind the next batch's compute."""
    S = velocities[0].shape[1] * network.res_increase
    res = np.empty((hi - lo, S, S, S, 3), dtype=np.float64)
    return _drain_to_host(network, _forward_chunks(network, velocities, magnitudes, batch_size, lo, hi, base=lo), res, batch_size)


def shard_bounds(n, world):
    """Contiguous shards of the patch list: rank r owns rows [bounds[r], bounds[r+1]) (the last ranks may own none)."""
    per = (n + world - 1) // world
    return [min(r * per, n) for r in range(world + 1)]


def predict_patches(network, velocities, magnitudes, batch_size):
    """The batched predict loop of predictor.py:79-94 (results accumulate in float64 like np.zeros + np.append there).

    Data-parallel (BASELINE cfg5): the patch list is sharded contiguously over the ranks, every rank runs the same pipelined loop
    on its shard, and rank 0 -- the only rank that stitches and writes -- receives exactly the rows each rank owns (no padding, no
    copy to ranks that do not need it).  Returns the (n,S,S,S,3) float64 array on rank 0 and None on the other ranks.
      nccl (= RCCL): fp32 predictions travel device-to-device; rank 0 then converts to float64 on the device and drains through the
      pinned staging buffers.
      gloo (CPU tests, two ranks on one device): every rank drains its own shard to the host and sends the float64 rows."""
    n = len(velocities[0])
    world, rank = parallel.world_size(), parallel.rank()
    if world == 1:
        return _predict_pipelined(network, velocities, magnitudes, batch_size, 0, n)
    import torch.distributed as dist
    S = velocities[0].shape[1] * network.res_increase
    bounds = shard_bounds(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    dev = torch.device(network.device)
    if dev.type == "cuda" and not parallel._host_staged():
        if rank == 0:
            full = torch.empty((n, S, S, S, 3), device=dev, dtype=torch.float32)
            res = np.empty((n, S, S, S, 3), dtype=np.float64)
            _drain_to_host(network, _forward_chunks(network, velocities, magnitudes, batch_size, lo, hi), res, batch_size)
            # receives are posted AFTER the own shard (a pending RCCL receive is a kernel spinning on a few CUs); the peers finish
            # their equal shards at about the same time, so only the transfer itself (1.3 MB per patch over xGMI) is exposed
            ops_ = [dist.P2POp(dist.irecv, full[bounds[r]:bounds[r + 1]], r) for r in range(1, world) if bounds[r + 1] > bounds[r]]
            for q in (dist.batch_isend_irecv(ops_) if ops_ else []):     # one grouped RCCL launch for all peers
                q.wait()                              # makes the current stream wait for the transfers; no host sync
            step = max(batch_size, 1)
            _drain_to_host(network, ((s, full[s:min(s + step, n)]) for s in range(hi, n, step)), res, batch_size)
            return res
        if hi > lo:
            mine = torch.empty((hi - lo, S, S, S, 3), device=dev, dtype=torch.float32)
            for _ in _forward_chunks(network, velocities, magnitudes, batch_size, lo, hi, keep=mine):
                pass
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine, 0)]):
                q.wait()
        return None
    # host transport
    if dev.type == "cuda":
        mine = _predict_pipelined(network, velocities, magnitudes, batch_size, lo, hi)
    else:                                             # CPU stand-in network (tests): same loop without the staging machinery
        mine = np.empty((hi - lo, S, S, S, 3), dtype=np.float64)
        for r0, out in _forward_chunks(network, velocities, magnitudes, batch_size, lo, hi, base=lo):
            mine[r0:r0 + out.shape[0]] = out.detach().cpu().numpy().astype(np.float64)
    if rank == 0:
        res = np.empty((n, S, S, S, 3), dtype=np.float64)
        res[lo:hi] = mine
        for r in range(1, world):
            if bounds[r + 1] > bounds[r]:
                dist.recv(torch.from_numpy(res[bounds[r]:bounds[r + 1]]), src=r)     # straight into the result rows
        return res
    if hi > lo:
        dist.send(torch.from_numpy(mine), dst=0)
    return None


def predict_file(network, input_filepath, output_filepath, patch_size, res_increase, batch_size=8,
                 round_small_values=True, verbose=True):
    """predictor.py:67-115 for every row of the input file.  Returns the list of (u,v,w) volumes written (rank 0; the other ranks
    of a data-parallel run compute their shard of every row's patches and return an empty list)."""
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
        if not is0:
            continue                              # rank 0 holds the gathered patches: it alone stitches and writes
        if verbose:
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
