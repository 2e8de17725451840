"""Per-stage time against batch size at d=512 / M=3 / 8 heads (bf16): the intercept of the line is what a stage pays per
launch whatever the batch (launch, weight prologue, first tile, tail)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("AECF_DEBUG", "graph=0")
import torch
import bench

dev = torch.device("cuda:0")
res = {}
for B in (8192, 16384, 32768, 65536, 131072):
    cfg = (B, 3, 512, 8, torch.bfloat16, 0.15)
    pool, query, x, dy = bench.make_inputs(cfg, dev)
    params = [query] + list(pool.parameters())
    for _ in range(10):
        bench.step(pool, query, x, dy, params, False)
    st = bench.StageTimer()
    for _ in range(30):
        st.arm(pool); bench.step(pool, query, x, dy, params, False); st.disarm(pool)
        torch.cuda.synchronize(); st.collect()
    res[B] = st.median_ms()
    del pool, query, x, dy, params
    torch.cuda.empty_cache()
names = [k for k in res[65536] if res[65536][k] > 0.02]
print("stage".ljust(14) + "".join(str(B).rjust(9) for B in res) + "   intercept(us)  us/64k")
for k in names:
    ys = [res[B][k] * 1e3 for B in res]
    # fit on the two largest sizes
    (b1, y1), (b2, y2) = (65536, res[65536][k] * 1e3), (131072, res[131072][k] * 1e3)
    slope = (y2 - y1) / (b2 - b1)
    print(k.ljust(14) + "".join(("%.1f" % y).rjust(9) for y in ys) + "   %8.1f  %8.1f" % (y1 - slope * b1, slope * 65536))
