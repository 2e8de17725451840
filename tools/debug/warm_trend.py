"""Step time of a host-bound shard (configs[2], 8192 rows) over a long run: on this pool it drops from ~0.30 to ~0.24 ms after
~0.7 s -- is that the step count, wall time with the CPU busy, or wall time?  usage: warm_trend.py [spin|sleep|none] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch, bench
dev = torch.device("cuda:0")
cfg = bench.CONFIGS[os.environ.get("CFG", "c3")]
pool, query, x, dy = bench.make_inputs(cfg, dev)
params = [query] + list(pool.parameters())
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
for _ in range(20):
    bench.step(pool, query, x, dy, params, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
if mode == "spin":
    k = 0
    while time.perf_counter() - t0 < secs:
        k += 1
elif mode == "sleep":
    time.sleep(secs)
out = []
for rep in range(16):
    t0 = time.perf_counter()
    for _ in range(200):
        bench.step(pool, query, x, dy, params, False)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 200 * 1e3, 3))
print(mode, out, flush=True)
