"""Eager step time (fwd + bwd through the Python layer) by call size with the library's graph replay forced off / on:
run once per setting, AECF_DEBUG=graph=0 | graph=1 (the knob is read once per process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
import bench

dev = torch.device("cuda:0")
out = []
for (B, M, E, H) in ((64, 3, 512, 8), (512, 3, 512, 8), (2048, 3, 512, 8), (4096, 2, 256, 4), (8192, 2, 768, 8), (8192, 3, 512, 8),
                     (16384, 3, 512, 8), (16384, 4, 1024, 8)):
    cfg = (B, M, E, H, torch.bfloat16, 0.15)
    pool, query, x, dy = bench.make_inputs(cfg, dev)
    params = [query] + list(pool.parameters())
    for _ in range(60):
        bench.step(pool, query, x, dy, params, False)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            bench.step(pool, query, x, dy, params, False)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
    out.append(f"{B}x{M}x{E}: {best:.3f}")
print(os.environ.get("AECF_DEBUG", "default"), " | ".join(out))
