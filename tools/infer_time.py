"""Forward-only (eval mode, no autograd) throughput of the pool at the headline shape and at small batches: what a serving
process that fuses resident embeddings sees.  usage: infer_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
import bench

dev = torch.device("cuda:0")
for B in (65536, 8192, 1024, 64):
    cfg = (B, 3, 512, 8, torch.bfloat16, 0.15)
    pool, query, x, dy = bench.make_inputs(cfg, dev)
    pool.eval()
    x = x.detach()
    with torch.no_grad():
        for _ in range(50):
            out, info = pool(query.expand(B, -1, -1), x, return_info=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(200):
            out, info = pool(query.expand(B, -1, -1), x, return_info=True)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 200 * 1e3
    ms = e0.elapsed_time(e1) / 200
    print(f"B={B}: {ms:.4f} ms per forward ({wall:.4f} wall) = {B / ms / 1e3:.1f} M samples/s", flush=True)
