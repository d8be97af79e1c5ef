"""Data-parallel plumbing (new relative to the reference, which is single-process -- SURVEY.md section 8e).

One process per GPU; torch.distributed with backend "nccl" (= RCCL over xGMI on ROCm) for GPU tensors and
"gloo" for the CPU tests.  The only data-path collective is the sum-all-reduce of the flat gradient buffer, issued per
step as a few contiguous buckets while backward is still running: SUM (not mean) reproduces the reference's single-process gradient of sum_b loss_b over the global batch
(tape.gradient of a vector target, TrainerController.py:223,249)."""
import os

import numpy as np
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def rank():
    return dist.get_rank() if is_dist() else 0


def world_size():
    return dist.get_world_size() if is_dist() else 1


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (set by torch.distributed.run).
    Returns (rank, world_size, local_rank).  A single process without those variables stays un-initialised."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lrk = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(lrk)
        dist.init_process_group(backend=backend, rank=rk, world_size=ws)
    return rk, ws, lrk


def _host_staged():
    """True when device tensors have to travel through host memory: the process group is gloo (CPU tests, and the
    single-GPU test harness that runs two ranks on ONE device, where RCCL refuses duplicate GPUs)."""
    return dist.get_backend() == "gloo"


def allreduce_sum_(flat):
    """In-place SUM all-reduce of one flat buffer (the 13.4 MB gradient vector at cfg2).  nccl (= RCCL): straight on
    the device buffer over xGMI.  gloo: device tensors are staged through host memory."""
    if is_dist() and world_size() > 1:
        if flat.is_cuda and _host_staged():
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_sum_start(flat):
    """Start an in-place SUM all-reduce of a (contiguous slice of a) flat device buffer and return a handle for
    allreduce_wait().  nccl (= RCCL): asynchronous on the process group's own stream, ordered after everything enqueued on the
    current stream so far.  gloo (host-staged): done synchronously here, handle None."""
    if not (is_dist() and world_size() > 1):
        return None
    if flat.is_cuda and _host_staged():
        allreduce_sum_(flat)
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_wait(handle):
    """Make the current stream wait for an all-reduce started by allreduce_sum_start (no host synchronisation under nccl)."""
    if handle is not None:
        handle.wait()


def all_gather_equal(t):
    """all_gather of equally shaped tensors -> list of world tensors on t's device (host-staged under gloo)."""
    if not is_dist() or world_size() == 1:
        return [t]
    if t.is_cuda and _host_staged():
        h = t.cpu()
        out = [torch.empty_like(h) for _ in range(world_size())]
        dist.all_gather(out, h)
        return [o.to(t.device) for o in out]
    out = [torch.empty_like(t) for _ in range(world_size())]
    dist.all_gather(out, t)
    return out


def allreduce_sum_host(values, op="sum"):
    """SUM (or "max") all-reduce of a few host scalars (epoch metrics: totals and counts; bench: step time).
    Returns a list of floats."""
    if not is_dist() or world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if not _host_staged():
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return t.cpu().tolist()


def global_batch_size(local_b, device=None):
    """Sum of the per-rank batch sizes (ranks may differ on a ragged last batch)."""
    if not is_dist() or world_size() == 1:
        return int(local_b)
    t = torch.tensor([float(local_b)], device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(t.item()))


def barrier():
    if is_dist():
        dist.barrier()


class ShardedIndexSampler:
    """Global shuffle with a shared seed per epoch (mirrors ds.shuffle(buffer_size=len), PatchHandler3D.py:30),
    then every global batch of batch_size*world rows is split into disjoint per-rank slices of batch_size."""

    def __init__(self, n_rows, batch_size, shuffle, seed=0, rank_=None, world=None):
        self.n = int(n_rows)
        self.bs = int(batch_size)
        self.shuffle = shuffle
        self.seed = seed
        self.rank = rank() if rank_ is None else rank_
        self.world = world_size() if world is None else world
        self.epoch = 0

    def set_epoch(self, e):
        self.epoch = e

    def __len__(self):
        gb = self.bs * self.world
        return (self.n + gb - 1) // gb

    def __iter__(self):
        order = np.arange(self.n)
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(order)
        gb = self.bs * self.world
        for s in range(0, self.n, gb):
            chunk = order[s:s + gb]                       # ragged last global batch is kept (PatchHandler3D.py:33)
            mine = chunk[self.rank * self.bs:(self.rank + 1) * self.bs]
            yield mine                                    # may be empty on the ragged tail
        self.epoch += 1
