"""Per-step HIP-event times of `bench.py`'s timed loop for a given warmup (is the device still warming up at step 0?).
usage: step_trend.py <warmup steps> <timed steps> [<ms of unrelated device work before the warmup>]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
import bench

warm, steps = int(sys.argv[1]), int(sys.argv[2])
spin_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["c2"]
pool, query, x, dy = bench.make_inputs(cfg, dev)
params = [query] + list(pool.parameters())
if spin_ms > 0:
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < spin_ms:
        for _ in range(10):
            a @ a
        torch.cuda.synchronize()
for _ in range(warm):
    bench.step(pool, query, x, dy, params, False)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for i in range(steps):
    bench.step(pool, query, x, dy, params, False)
    ev[i + 1].record()
torch.cuda.synchronize()
print(warm, spin_ms, [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(steps)])
