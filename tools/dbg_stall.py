"""Where does the one 2.5-s step of bench.py's `sustained` leg come from?  Per-step HIP-event times + allocator statistics."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
trainer = importlib.import_module("4dflownet_amd.trainer")
dev = torch.device("cuda", 0)
for leg in range(2):
    tc = trainer.TrainerController(24, 2, initial_learning_rate=1e-4, quicksave_enable=False, low_resblock=8, hi_resblock=4, device=dev, seed=0)
    batch = bench.synthetic_batch(8, 24, 2, 1234, dev)
    for _ in range(5): tc.train_step(batch)
    torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    evs[0].record()
    for i in range(n):
        t0 = time.perf_counter(); tc.train_step(batch); host.append(time.perf_counter() - t0); evs[i + 1].record()
    torch.cuda.synchronize()
    per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n)])
    st1 = torch.cuda.memory_stats()
    print("leg %d overlap=%s: median %.2f max %.1f ms at step %d; host-side max %.1f ms at step %d; device mallocs %d -> %d, retries %d -> %d, reserved %.2f GB" % (
        leg, tc.model.overlap_wgrad, np.median(per), per.max(), per.argmax(), max(host) * 1e3, int(np.argmax(host)),
        st0["num_device_alloc"], st1["num_device_alloc"], st0["num_alloc_retries"], st1["num_alloc_retries"], st1["reserved_bytes.all.current"] / 2**30), flush=True)
    del tc, batch, evs
    torch.cuda.empty_cache()
