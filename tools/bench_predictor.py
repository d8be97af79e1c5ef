"""cfg5: full-volume sliding-window inference throughput (patch 24, res x2, batch 8, 8+4 ResBlocks, fp32).
Times predictor.predict_patches (tiler -> batched HIP forward -> gather) on the shipped example volume (12 patches) and on a
synthetic 100x100x100 volume (125 patches); prints patches/s.  Launch under torch.distributed.run to shard patches over ranks."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
parallel = importlib.import_module("4dflownet_amd.parallel")
predictor = importlib.import_module("4dflownet_amd.predictor")
data = importlib.import_module("4dflownet_amd.data")
tiler = importlib.import_module("4dflownet_amd.tiler")


class _Vol:
    pass


def main():
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    dtype = "bfloat16" if "--bf16" in sys.argv else "float32"
    net = predictor.prepare_network(24, 2, 8, 4, dtype=dtype)
    cases = []
    ds = data.ImageDataset()
    ds.load_vectorfield(os.path.join(ROOT, "tests", "golden", "data", "example_data.h5"), 0)
    cases.append(("example_data.h5 (42x38x36)", ds))
    rng = np.random.default_rng(0)
    v = _Vol()
    for n in ("u", "v", "w"):
        setattr(v, n, rng.uniform(-1, 1, (100, 100, 100)).astype(np.float32))
    for n in ("mag_u", "mag_v", "mag_w"):
        setattr(v, n, rng.uniform(0, 0.016, (100, 100, 100)).astype(np.float32))
    cases.append(("synthetic 100^3", v))
    for name, vol in cases:
        pg = tiler.PatchGenerator(24, 2)
        vel, mag = pg.patchify(vol)
        predictor.predict_patches(net, vel, mag, 8)            # warm-up
        torch.cuda.synchronize(); parallel.barrier()
        t0 = time.perf_counter()
        res = predictor.predict_patches(net, vel, mag, 8)
        torch.cuda.synchronize(); parallel.barrier()
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        out = pg.unpatchify(res)
        ts = time.perf_counter() - t1
        if rank == 0:
            print("%-28s %4d patches on %d GPU(s), %s: forward+gather %.3f s = %.1f patches/s; stitch %.3f s -> %s"
                  % (name, len(res), world, dtype, dt, len(res) / dt, ts, out[0].shape))


if __name__ == "__main__":
    main()
