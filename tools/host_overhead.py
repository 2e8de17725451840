"""Host-side cost of one bench step (python + torch + ctypes launch path) against its GPU time: issues K steps without
synchronising and reports when the host finished issuing vs when the GPU finished.   usage: python tools/host_overhead.py"""
import sys, time, torch
sys.path.insert(0, ".")
import bench

for name in ("c2", "c3", "tiny"):
    cfg = bench.CONFIGS[name]
    pool, query, x, dy = bench.make_inputs(cfg, torch.device("cuda:0"))
    params = [query] + list(pool.parameters())
    for _ in range(10):
        bench.step(pool, query, x, dy, params, False)
    torch.cuda.synchronize()
    K = 50
    t0 = time.perf_counter()
    for _ in range(K):
        bench.step(pool, query, x, dy, params, False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host issue {1e3 * (t1 - t0) / K:.3f} ms/step, total {1e3 * (t2 - t0) / K:.3f} ms/step")
