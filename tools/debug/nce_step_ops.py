"""Which torch operators run inside one `bench.py --config c3 --contrastive` step besides the library's kernels (torch.profiler)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                # noqa: E402
from aecf_amd import losses                                 # noqa: E402
dev = torch.device("cuda:0")
B, M, E, H, dtype, p = bench.CONFIGS["c3"]
pool, query, x, dy = bench.make_inputs((B, M, E, H, dtype, p), dev)
params = [query] + list(pool.parameters())
keys = losses.l2_normalize(torch.randn(65536, E, device=dev).to(dtype)).detach()
zb = torch.randn(B, E, device=dev).to(dtype)


def step():
    out, info = pool(query.expand(B, -1, -1), x, return_info=True)
    nb = losses.l2_normalize(zb)
    keys[3 * B:4 * B] = nb
    loss = losses.gathered_contrastive_entropy_loss(out.squeeze(1), keys, 3 * B, pool.curriculum_masking, info["entropy"], temperature=0.07)
    x.grad = None
    for q in params:
        q.grad = None
    loss.backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::mul", "aten::copy_") and e.device_time_total > 8:
        print(e.name, e.input_shapes, round(e.device_time_total, 1), "us")
