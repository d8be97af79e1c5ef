"""Data-parallel path on CPU with gloo, world_size 2: the flat-gradient SUM all-reduce reproduces the
single-process gradient of the global batch, shards are disjoint, ragged tails are handled."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import flownet_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel = importlib.import_module("4dflownet_amd.parallel")
    r, w, lr = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.rank() == rank and parallel.world_size() == world
    P, R, LB, HB = 4, 2, 1, 1
    params = O.init_params(0, LB, HB, np.float64)
    full = O.synthetic_batch(4, P, R, seed=5, dtype=np.float64)
    sampler = parallel.ShardedIndexSampler(4, 2, shuffle=False)
    rows = next(iter(sampler))
    mine = tuple(a[rows] for a in full)
    out = O.loss_and_grads(params, mine, R, LB, HB)
    # the product folds L2 in after the all-reduce with the GLOBAL batch; emulate: strip the local L2 part first
    g = O.flatten(out["grads"]) - len(rows) * 2 * O.L2_LAMBDA * O.flatten([{"w": p["w"], "b": None if p["b"] is None else 0 * p["b"]} for p in params])
    flat = torch.from_numpy(g.copy())
    parallel.allreduce_sum_(flat)
    Bg = parallel.global_batch_size(len(rows))
    # ragged tail: 5 rows, batch 2 x 2 ranks -> global batches of 4 and 1; rank 1's last slice is empty
    s2 = parallel.ShardedIndexSampler(5, 2, shuffle=True, seed=1)
    sizes = [len(x) for x in s2]
    parallel.barrier()
    q.put((rank, rows.tolist(), flat.numpy(), Bg, sizes))
    dist.destroy_process_group()


def test_dp2_sum_allreduce_equals_global_batch_gradient():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1] and res[1][1] == [2, 3]                  # disjoint shards of the global batch
    assert np.array_equal(res[0][2], res[1][2])                          # every rank holds the same reduced gradient
    assert res[0][3] == 4 and res[1][3] == 4
    assert res[0][4] == [2, 1] and res[1][4] == [2, 0]
    # single-process reference: gradient of sum_b loss_b over the whole batch of 4, without the L2 term
    P, R, LB, HB = 4, 2, 1, 1
    params = O.init_params(0, LB, HB, np.float64)
    full = O.synthetic_batch(4, P, R, seed=5, dtype=np.float64)
    out = O.loss_and_grads(params, full, R, LB, HB)
    gref = O.flatten(out["grads"]) - 4 * 2 * O.L2_LAMBDA * O.flatten([{"w": p["w"], "b": None if p["b"] is None else 0 * p["b"]} for p in params])
    np.testing.assert_allclose(res[0][2], gref, rtol=1e-9, atol=1e-12)


def test_sampler_covers_everything_once_per_epoch():
    parallel = importlib.import_module("4dflownet_amd.parallel")
    seen = []
    for r in range(4):
        s = parallel.ShardedIndexSampler(50, 3, shuffle=True, seed=7, rank_=r, world=4)
        assert len(s) == 5
        for rows in s:
            seen.extend(rows.tolist())
    assert sorted(seen) == list(range(50))


class _FakeNet:
    """Stands in for FlowNetModel in predict_patches: a deterministic function of the inputs on CPU tensors."""
    res_increase = 2
    device = torch.device("cpu")

    def forward(self, ins):
        u = torch.as_tensor(np.asarray(ins[0], dtype=np.float32))[..., 0]          # (B,P,P,P)
        up = u.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
        m = torch.as_tensor(np.asarray(ins[3], dtype=np.float32))[..., 0].mean(dim=(1, 2, 3))
        return torch.stack([up, 2 * up, up + m[:, None, None, None]], dim=-1)


def _patches(n, P=4, seed=9):
    rng = np.random.default_rng(seed)
    vel = [rng.normal(size=(n, P, P, P, 1)).astype(np.float32) for _ in range(3)]
    mag = [rng.uniform(size=(n, P, P, P, 1)).astype(np.float32) for _ in range(3)]
    return vel, mag


def _predict_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel = importlib.import_module("4dflownet_amd.parallel")
    predictor = importlib.import_module("4dflownet_amd.predictor")
    parallel.init_from_env(backend="gloo")
    out = []
    for n in CFG5_CASES[world]:
        vel, mag = _patches(n)
        out.append(predictor.predict_patches(_FakeNet(), vel, mag, batch_size=2))
    parallel.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


# patches per case; world 2: shards 3+2, 1+0 (an empty shard), 2+2; world 3: 3+3+1, 1+1+0, 1+0+0, 2+2+2
CFG5_CASES = {2: (5, 1, 4), 3: (7, 2, 1, 6)}


@pytest.mark.parametrize("world", [2, 3])
def test_cfg5_patch_list_sharded_over_ranks_and_gathered_to_root(world):
    """Inference: the patch list is split contiguously over ranks (predictor.shard_bounds), every rank sends
    exactly the rows it owns to rank 0 -- no padding, no copy to ranks that do not stitch -- and rank 0 ends up with the complete,
    correctly ordered float64 result; ragged and empty shards included.  The other ranks return None."""
    predictor = importlib.import_module("4dflownet_amd.predictor")
    assert predictor.shard_bounds(5, 2) == [0, 3, 5] and predictor.shard_bounds(1, 2) == [0, 1, 1]
    assert predictor.shard_bounds(7, 3) == [0, 3, 6, 7] and predictor.shard_bounds(1, 3) == [0, 1, 1, 1]
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_predict_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, n in enumerate(CFG5_CASES[world]):
        vel, mag = _patches(n)
        ref = _FakeNet().forward([v for v in vel] + [m for m in mag]).numpy().astype(np.float64)
        got = res[0][1][k]
        assert got.shape == (n, 8, 8, 8, 3) and got.dtype == np.float64
        np.testing.assert_array_equal(got, ref)
        for r in range(1, world):
            assert res[r][1][k] is None


def _helpers_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel = importlib.import_module("4dflownet_amd.parallel")
    trainer = importlib.import_module("4dflownet_amd.trainer")
    parallel.init_from_env(backend="gloo")
    sums = parallel.allreduce_sum_host([rank + 1.0, 10.0 * (rank + 1)])
    mx = parallel.allreduce_sum_host([float(rank)], op="max")
    gathered = parallel.all_gather_equal(torch.full((2, 3), float(rank)))
    # epoch metrics: (total, count) are combined over ranks -- a rank with an empty shard (count 0) does not dilute the mean
    m = trainer.Mean("x", torch.device("cpu"))
    if rank == 0:
        m.update_state(torch.tensor([1.0, 2.0, 3.0]))
    glob = m.result_global()
    # bucketed gradient all-reduce (trainer.train_step): contiguous slices of one flat buffer, started one after the other, waited for
    # at the end -- together they must equal ONE all-reduce of the whole buffer
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    handles = [parallel.allreduce_sum_start(flat[lo:hi]) for lo, hi in ((6, 10), (2, 6), (0, 2))]
    for h in handles:
        parallel.allreduce_wait(h)
    parallel.barrier()
    q.put((rank, sums, mx, [g.tolist() for g in gathered], m.result(), glob, flat.tolist()))
    dist.destroy_process_group()


def test_host_collectives_and_rank_combined_metrics():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_helpers_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r][1] == [3.0, 30.0] and res[r][2] == [1.0]
        assert res[r][3] == [[[0.0] * 3] * 2, [[1.0] * 3] * 2]
        assert res[r][5] == 2.0                        # global mean of rank 0's three values; rank 1 contributed nothing
        assert res[r][6] == [3.0 * i for i in range(10)]
    assert res[0][4] == 2.0 and res[1][4] == 0.0
