"""Product-level data-parallel tests on the GPU (BASELINE cfg3 / cfg5): two ranks run the REAL
TrainerController.train_step / predictor.predict_file and must reproduce the single-process run on the global batch.

Semantics under test (src/Network/TrainerController.py:223,245-249): tape.gradient of the (B,) loss vector = gradient
of sum_b loss_b with the scalar L2 term counted once per sample -> SUM (not mean) all-reduce of the per-rank
gradients, L2 applied with the GLOBAL batch size, ragged tails kept (PatchHandler3D.py:33): a rank whose shard is empty
contributes zeros and still joins the collective.

With >= 2 GPUs the ranks use one GPU each over nccl (= RCCL); on a 1-GPU box both ranks share cuda:0 and the
collective is gloo with host staging (RCCL refuses two ranks on one device) -- the product code path is the same
apart from the transport inside parallel.allreduce_sum_."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import flownet_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")

P, R, LB, HB = 8, 2, 1, 1
LR = 1e-4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _init(rank, world, port):
    ngpu = torch.cuda.device_count()
    backend = "nccl" if ngpu >= world else "gloo"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank % ngpu), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank % ngpu)
    parallel = importlib.import_module("4dflownet_amd.parallel")
    parallel.init_from_env(backend=backend)
    return parallel


def _global_batches():
    """Step 1: a full global batch of 4 (2 per rank).  Step 2: a ragged one of 2 rows -> rank 0 takes both, rank 1 none."""
    return [O.synthetic_batch(4, P, R, seed=31), O.synthetic_batch(2, P, R, seed=32)]


def _train_worker(rank, world, port, q):
    parallel = _init(rank, world, port)
    trainer = importlib.import_module("4dflownet_amd.trainer")
    tc = trainer.TrainerController(P, R, initial_learning_rate=LR, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=0)
    grads = []
    for gb in _global_batches():
        n = len(gb[0])
        rows = next(iter(parallel.ShardedIndexSampler(n, 2, shuffle=False)))       # this rank's slice of the global batch
        tc.train_step(tuple(a[rows] for a in gb))
        grads.append((len(rows), tc.model.flat_g_ext.cpu().numpy().copy()))
    # validation pass with fewer rows than batch*world (ranks with an empty shard must not crash or hang)
    vb = O.synthetic_batch(1, P, R, seed=33)
    rows = next(iter(parallel.ShardedIndexSampler(1, 2, shuffle=False)))
    tc.test_step(tuple(a[rows] for a in vb))
    res = dict((k, v.result_global()) for k, v in tc.loss_metrics.items())
    torch.cuda.synchronize()
    parallel.barrier()
    q.put((rank, grads, tc.model.flat_w.cpu().numpy().copy(), res))
    torch.distributed.destroy_process_group()


def _run(target, world=2, extra=()):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return res


def test_dp2_train_step_equals_single_process_global_batch(fdn):
    trainer = importlib.import_module("4dflownet_amd.trainer")
    res = _run(_train_worker)
    # single process on the global batches
    tc = trainer.TrainerController(P, R, initial_learning_rate=LR, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=0)
    ref_g = []
    for gb in _global_batches():
        tc.train_step(gb)
        ref_g.append(tc.model.flat_g_ext.cpu().numpy().copy())
    tc.test_step(O.synthetic_batch(1, P, R, seed=33))
    ref_w = tc.model.flat_w.cpu().numpy()
    ref_res = dict((k, v.result()) for k, v in tc.loss_metrics.items())

    assert [n for n, _ in res[0][1]] == [2, 2] and [n for n, _ in res[1][1]] == [2, 0]      # rank 1's second shard is empty
    for step in range(2):
        g0, g1 = res[0][1][step][1], res[1][1][step][1]
        assert np.array_equal(g0, g1)                               # every rank holds the same reduced buffer
        assert g0[-1] == (4.0, 2.0)[step]                           # the batch-size slot carries the GLOBAL batch
        # vs the single-process gradient of the global batch: equal up to fp32 effects -- the partial sums are grouped by
        # shard, and the kernel planner may tile N=2 and N=4 launches differently, so a ReLU unit within an ulp of its kink can
        # land on the other side (each flip moves the gradient by ~1/(B*V) relative; test_gpu_train_step.kink_sides).
        # north_star tolerance: 1e-3 relative.
        ref = ref_g[step][:-1].astype(np.float64)
        d = g0[:-1].astype(np.float64) - ref
        assert np.linalg.norm(d) <= 1e-3 * np.linalg.norm(ref)
        assert np.abs(d).max() <= 1e-3 * np.abs(ref).max()
    # The collective itself is exact: step 1's reduced buffer == fp32 sum of the two shard gradients computed one after the
    # other in THIS process with the same N=2 launches (a 2-rank SUM is one fp32 add per element), batch slot 2 + 2.
    shard_g = []
    gb = _global_batches()[0]
    for rows in ([0, 1], [2, 3]):
        t = trainer.TrainerController(P, R, initial_learning_rate=LR, quicksave_enable=False, low_resblock=LB, hi_resblock=HB, seed=0)
        t.train_step(tuple(a[rows] for a in gb))
        shard_g.append(t.model.flat_g_ext.cpu().numpy().copy())
    assert np.array_equal(res[0][1][0][1], shard_g[0] + shard_g[1])
    # weights after two Adam steps: identical on both ranks; equal to the single-process run.  Adam moves a weight by
    # ~lr*sign(g) in its first steps, so an element whose gradient is summation-order noise may differ by up to 2*lr per
    # step; every well-conditioned element must agree to 1e-6.
    assert np.array_equal(res[0][2], res[1][2])
    dw = np.abs(res[0][2].astype(np.float64) - ref_w)
    assert dw.max() <= 4.2 * LR
    g = np.abs(ref_g[0][:-1])
    good = g >= 1e-3 * g.max()
    assert good.sum() > 0.2 * good.size
    assert dw[good].max() <= 1e-6
    # epoch metrics combine (total, count) over ranks: same numbers as the single process
    for k in ("train_loss", "train_mse", "train_accuracy", "val_loss", "val_accuracy"):
        for r in range(2):
            assert abs(res[r][3][k] - ref_res[k]) <= 1e-4 * max(abs(ref_res[k]), 1e-6), (k, res[r][3][k], ref_res[k])
    # l2_reg_loss is logged once per step and rank that had samples (3 entries here vs 2 in the single process)
    assert abs(res[0][3]["l2_reg_loss"] - ref_res["l2_reg_loss"]) <= 1e-2 * ref_res["l2_reg_loss"]


def _predict_worker(rank, world, port, q, outdir):
    parallel = _init(rank, world, port)
    predictor = importlib.import_module("4dflownet_amd.predictor")
    net = predictor.prepare_network(24, 2, 1, 1)
    out = os.path.join(outdir, "dp_result.h5")
    vols = predictor.predict_file(net, os.path.join(DATA, "example_data.h5"), out, 24, 2, batch_size=4, verbose=False)
    torch.cuda.synchronize()
    parallel.barrier()
    q.put((rank, [np.asarray(v, np.float32) for v in vols[0]] if vols else None))
    torch.distributed.destroy_process_group()


def test_dp2_predict_file_equals_single_process(tmp_path):
    """cfg5: the 12 patches of example_data.h5 are split 6 + 6 over two ranks, gathered to rank 0, stitched and written by rank 0."""
    predictor = importlib.import_module("4dflownet_amd.predictor")
    h5io = importlib.import_module("4dflownet_amd.h5io")
    res = _run(_predict_worker, extra=(str(tmp_path),))
    net = predictor.prepare_network(24, 2, 1, 1)
    ref = predictor.predict_file(net, os.path.join(DATA, "example_data.h5"), str(tmp_path / "single.h5"), 24, 2, batch_size=4,
                                 verbose=False)
    back = h5io.read_all(str(tmp_path / "dp_result.h5"))          # written once, by rank 0
    single = h5io.read_all(str(tmp_path / "single.h5"))
    for i, name in enumerate("uvw"):
        assert back[name].shape == (1, 84, 76, 72)
        scale = np.abs(ref[0][i]).max()
        assert np.abs(res[0][1][i] - np.asarray(ref[0][i], np.float32)).max() <= 1e-5 * scale
        assert res[1][1] is None                       # rank 1 computed its 6 patches and sent them; it neither stitches nor writes
        assert np.abs(back[name] - single[name]).max() <= 1e-5 * scale


def test_bench_self_spawns_ranks(tmp_path):
    """`python bench.py --gpus 2` without torchrun must launch two ranks itself (VERDICT r1 missing #1).  On a 1-GPU box
    it needs --oversubscribe (two ranks on one device, gloo): the line is then marked and is no scaling number."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--patch", "8",
            "--low", "1", "--hi", "1", "--batch", "2", "--no-cpu-baseline", "--no-secondary"]
    if torch.cuda.device_count() < 2:
        bad = subprocess.run(args, env=env, capture_output=True, text=True, timeout=600)
        assert bad.returncode != 0 and "GPU" in (bad.stderr + bad.stdout)          # hard failure, not a silent 1-rank run
        args.append("--oversubscribe")
    r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["collective_ranks"] == 2 and line["config"]["global_batch"] == 4
    if torch.cuda.device_count() < 2:
        # two ranks on ONE device over gloo: the line must not read as a 2-GPU RCCL run
        assert line["n_gpus"] == 1 and line["backend"] == "gloo" and line["rccl_ranks"] == 0 and line["oversubscribed"] is True
    else:
        assert line["n_gpus"] == 2 and line["backend"] == "rccl" and line["rccl_ranks"] == 2
    assert line["config"]["parallelism"] == "dp2" and line["value"] > 0


def _rccl_worker(rank, world, port, q):
    """ONE rank on cuda:0 in a world-1 nccl (= RCCL) group, with parallel.world_size patched to 2 so that train_step takes the
    data-parallel branch: every gradient bucket goes through a real asynchronous RCCL all-reduce on the process group's stream
    (the identity for one rank), started from inside backward() and waited for before the Adam launch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    parallel = importlib.import_module("4dflownet_amd.parallel")
    trainer = importlib.import_module("4dflownet_amd.trainer")
    parallel.world_size = lambda: 2
    started = []
    start = parallel.allreduce_sum_start
    def counting_start(flat):
        h = start(flat)
        started.append((flat.numel(), h is not None))
        return h
    parallel.allreduce_sum_start = counting_start
    tc = trainer.TrainerController(P, R, initial_learning_rate=LR, quicksave_enable=False, low_resblock=2, hi_resblock=HB, seed=0)
    out = []
    for seed in (41, 42):
        tc.train_step(O.synthetic_batch(2, P, R, seed=seed))
        out.append(tc.model.flat_g_ext.cpu().numpy().copy())
    torch.cuda.synchronize()
    q.put((0, out, tc.model.flat_w.cpu().numpy().copy(), started, list(tc.model.grad_buckets)))
    torch.distributed.destroy_process_group()


def test_bucketed_rccl_allreduce_inside_backward_is_transparent(fdn):
    """The asynchronous per-bucket RCCL path (what 8 GPUs run) on one GPU: same gradient buffer and same weights, bit for bit,
    as the plain single-process steps; the buckets tile the extended gradient buffer exactly and in completion order."""
    trainer = importlib.import_module("4dflownet_amd.trainer")
    (_, grads, w, started, buckets), = _run(_rccl_worker, world=1)
    tc = trainer.TrainerController(P, R, initial_learning_rate=LR, quicksave_enable=False, low_resblock=2, hi_resblock=HB, seed=0)
    n = tc.model.n_params
    assert len(buckets) == 3 and buckets[0][1] == n + 1 and buckets[-1][0] == 0
    assert all(buckets[i][0] == buckets[i + 1][1] for i in range(2))
    assert started == [(hi - lo, True) for (lo, hi) in buckets] * 2           # 3 asynchronous collectives per step
    for k, seed in enumerate((41, 42)):
        tc.train_step(O.synthetic_batch(2, P, R, seed=seed))
        assert np.array_equal(tc.model.flat_g_ext.cpu().numpy(), grads[k])
    assert np.array_equal(tc.model.flat_w.cpu().numpy(), w)


def _rccl_delayed_worker(rank, world, port, q, bucketed):
    """As _rccl_worker, but every collective is (a) issued behind a ~100 ms spin kernel on a side stream, so it completes long
    after backward has finished and the host has reached the Adam launch, and (b) preceded on that stream by `bucket += 1`,
    so its completion is visible in the data.  If train_step launched Adam without making the compute stream wait for the
    handles, Adam would consume the un-incremented gradients."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    parallel = importlib.import_module("4dflownet_amd.parallel")
    trainer = importlib.import_module("4dflownet_amd.trainer")
    parallel.world_size = lambda: 2
    side = torch.cuda.Stream()
    started = []
    # calibrate the spin kernel (its cycle counter's rate differs between devices): aim at ~60 ms per collective
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1_000_000); torch.cuda.synchronize()
    e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
    spin = int(10_000_000 * 60.0 / max(e0.elapsed_time(e1), 1e-3))

    def delayed_start(flat):
        side.wait_stream(torch.cuda.current_stream())            # the bucket's producers
        with torch.cuda.stream(side):
            torch.cuda._sleep(spin)
            flat.add_(1.0)
            h = torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, async_op=True)
        started.append(flat.numel())
        return h
    parallel.allreduce_sum_start = delayed_start
    tc = trainer.TrainerController(P, R, initial_learning_rate=LR, quicksave_enable=False, low_resblock=2, hi_resblock=HB, seed=0,
                                   bucketed_allreduce=bucketed)
    tc.profile_allreduce = True
    import time
    host_ms = []
    # batches staged on the device first: a pageable host->device copy inside train_step would queue behind the blocked compute
    # stream and stall the host for a reason that has nothing to do with the collective
    batches = [tuple(torch.as_tensor(a, device="cuda") for a in O.synthetic_batch(2, P, R, seed=seed)) for seed in (41, 42)]
    torch.cuda.synchronize()
    for b in batches:
        t0 = time.perf_counter()
        tc.train_step(b)
        host_ms.append((time.perf_counter() - t0) * 1e3)         # the host must NOT have blocked on the collectives
    torch.cuda.synchronize()
    waits = [e0.elapsed_time(e1) for e0, e1 in tc.allreduce_wait_events]
    q.put((0, tc.model.flat_w.cpu().numpy().copy(), tc.model.flat_g_ext.cpu().numpy().copy(), started, waits, host_ms))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("bucketed", [True, False])
def test_adam_cannot_overtake_a_slow_allreduce(fdn, bucketed):
    """Stress form of the test above: collectives that finish ~0.1 s late must still be ordered before the
    Adam launch on the compute stream -- for the bucketed path and for the single post-backward all-reduce."""
    trainer = importlib.import_module("4dflownet_amd.trainer")
    ops = importlib.import_module("4dflownet_amd.ops")
    (_, w, g, started, waits, host_ms), = _run(_rccl_delayed_worker, world=1, extra=(bucketed,))
    # reference: plain single-process steps whose gradient buffer (incl. the batch slot) is incremented right before Adam
    tc = trainer.TrainerController(P, R, initial_learning_rate=LR, quicksave_enable=False, low_resblock=2, hi_resblock=HB, seed=0)
    n = tc.model.n_params
    adam = ops.adam_step

    def adam_after_increment(*a, **k):
        tc.model.flat_g_ext.add_(1.0)
        return adam(*a, **k)
    ops.adam_step = adam_after_increment
    try:
        for seed in (41, 42):
            tc.train_step(O.synthetic_batch(2, P, R, seed=seed))
    finally:
        ops.adam_step = adam
    assert started == ([hi - lo for lo, hi in tc.model.grad_buckets] * 2 if bucketed else [n + 1] * 2)
    assert np.array_equal(g, tc.model.flat_g_ext.cpu().numpy())
    assert np.array_equal(w, tc.model.flat_w.cpu().numpy())
    # the delay really was exposed on the compute stream and the host never blocked on it.  (Judged on the second step: in the
    # first one the host is slow -- communicator set-up inside the first collective -- and may reach the wait after the GPU is done.)
    assert len(waits) == 2 and waits[1] > 20.0, waits
    assert host_ms[1] < 0.5 * waits[1], (host_ms, waits)


def _rccl_two_gpu_worker(rank, world, port, q):
    parallel = _init(rank, world, port)
    assert torch.distributed.get_backend() == "nccl"
    trainer = importlib.import_module("4dflownet_amd.trainer")
    handles = []
    start = parallel.allreduce_sum_start

    def counting_start(flat):
        h = start(flat)
        handles.append((flat.numel(), h is not None))
        return h
    parallel.allreduce_sum_start = counting_start
    tc = trainer.TrainerController(P, R, initial_learning_rate=LR, quicksave_enable=False, low_resblock=2, hi_resblock=HB, seed=0)
    gb = O.synthetic_batch(4, P, R, seed=51)
    rows = [2 * rank, 2 * rank + 1]
    for _ in range(3):
        tc.train_step(tuple(a[rows] for a in gb))
    torch.cuda.synchronize()
    parallel.barrier()
    q.put((rank, tc.model.flat_g_ext.cpu().numpy().copy(), tc.model.flat_w.cpu().numpy().copy(), handles))
    torch.distributed.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_bucketed_rccl_allreduce_two_gpus(fdn):
    """The asynchronous bucketed path with a REAL peer over xGMI: only runs where two devices are visible (the driver's
    multi-GPU node); self-skips on the 1-GPU box."""
    trainer = importlib.import_module("4dflownet_amd.trainer")
    res = _run(_rccl_two_gpu_worker)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert all(is_async for _, is_async in res[0][3]) and len(res[0][3]) == 9
    assert res[0][1][-1] == 4.0
    tc = trainer.TrainerController(P, R, initial_learning_rate=LR, quicksave_enable=False, low_resblock=2, hi_resblock=HB, seed=0)
    gb = O.synthetic_batch(4, P, R, seed=51)
    for _ in range(3):
        tc.train_step(gb)
    ref = tc.model.flat_g_ext.cpu().numpy()[:-1].astype(np.float64)
    d = res[0][1][:-1].astype(np.float64) - ref
    assert np.linalg.norm(d) <= 1e-3 * np.linalg.norm(ref)
