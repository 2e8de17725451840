"""Where the host time of one eager step goes (cProfile over N steps of bench.step at a small shard, device kept busy)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
import bench

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
cfg = bench.CONFIGS[cfgname]
pool, query, x, dy = bench.make_inputs(cfg, dev)
params = [query] + list(pool.parameters())
for _ in range(50):
    bench.step(pool, query, x, dy, params, False)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(n):
    bench.step(pool, query, x, dy, params, False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time per step {1e3 * (t1 - t0) / n:.3f} ms; with the device drained {1e3 * (t2 - t0) / n:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    bench.step(pool, query, x, dy, params, False)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
