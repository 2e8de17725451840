"""What the literal "1e-3 relative, bf16" costs: step time of the float32-store verification form (AECF_PRECISE,
layer.precise_forward_backward) at the headline shape [B=65536, M=3, d=512, 8 heads], next to the production bf16 step.
usage: precise_time.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
import bench
from aecf_amd.layer import precise_forward_backward

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
cfg = (B, 3, 512, 8, torch.bfloat16, 0.15)
pool, query, x, dy = bench.make_inputs(cfg, dev)
a = pool.attention
args = (x.detach(), query.detach(), a.in_proj_weight.detach(), a.in_proj_bias.detach(), a.out_proj.weight.detach(),
        a.out_proj.bias.detach(), 8, dy.detach())


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ms[len(ms) // 2]


params = [query] + list(pool.parameters())
t_prod = timed(lambda: bench.step(pool, query, x, dy, params, False), 40)
t_prec = timed(lambda: precise_forward_backward(*args))
print(f"B={B} M=3 d=512 H=8 bf16 inputs: production bf16-store step {t_prod:.3f} ms ({B / t_prod / 1e3:.1f} M samples/s); "
      f"AECF_PRECISE float32-store form (forward + backward, no masking / entropy loss) {t_prec:.3f} ms "
      f"({B / t_prec / 1e3:.1f} M samples/s) = {t_prec / t_prod:.2f}x the time")
