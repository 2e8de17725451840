"""cProfile of the host side of the bench step (where the ~0.4 ms of python/torch/ctypes time per step goes)."""
import cProfile, pstats, sys, torch
sys.path.insert(0, ".")
import bench
cfg = bench.CONFIGS["tiny"]
pool, query, x, dy = bench.make_inputs(cfg, torch.device("cuda:0"))
params = [query] + list(pool.parameters())
for _ in range(20):
    bench.step(pool, query, x, dy, params, False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    bench.step(pool, query, x, dy, params, False)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
