"""One-off check at B = 2^20 (not part of the test suite: ~15 GB of HBM): finite outputs, batch-slice bit-equality with a
4096-row run, run-to-run bit-equality of dW_in.   usage: python tools/big_batch_check.py"""
import sys, time, torch
sys.path.insert(0, ".")
import aecf_amd
from aecf_amd import layer
dev = torch.device("cuda:0")
B, M, E, H = 1 << 20, 3, 512, 8
torch.manual_seed(0)
q, pool = aecf_amd.create_fusion_pool(E, M, mask_prob=0.15, num_heads=H)
pool = pool.to(dev, torch.bfloat16).train()
q = torch.nn.Parameter(q.detach().to(dev, torch.bfloat16))
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, M, E, device=dev, generator=g).to(torch.bfloat16).requires_grad_(True)
dy = torch.randn(B, 1, E, device=dev, generator=g).to(torch.bfloat16)
U = torch.rand(B, 1, M, device=dev, generator=g)
def run(n):
    xs = x[:n].detach().requires_grad_(True)
    for p in list(pool.parameters()) + [q]: p.grad = None
    out, info = pool(q.expand(n, -1, -1), xs, return_info=True, uniforms=U[:n])
    out.backward(dy[:n])
    torch.cuda.synchronize()
    return out.detach(), info["masked_attention_weights"].detach(), xs.grad, pool.attention.in_proj_weight.grad.clone()
t0 = time.time(); big = run(B); t1 = time.time()
small = run(4096)
print("B", B, "sec", round(t1 - t0, 3), "mem GB", round(torch.cuda.max_memory_allocated() / 2**30, 1))
print("finite", all(bool(torch.isfinite(t.float()).all()) for t in big))
print("slice bit-equal", [bool(torch.equal(b[:4096], s)) for b, s in zip(big[:3], small[:3])])
# second run: determinism of the parameter gradient at this size
again = run(B)
print("dw_in deterministic", bool(torch.equal(again[3], big[3])))
