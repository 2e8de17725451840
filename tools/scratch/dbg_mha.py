import sys, torch
sys.path.insert(0, '.')
import aecf_amd
from oracle import aecf_oracle as O
DEV='cuda:0'
torch.set_printoptions(linewidth=200, precision=4)
for (E,H,B,T,S,dtype) in [(32,4,5,2,5,torch.float32)]:
    g = torch.Generator().manual_seed(E + T + S)
    pool = aecf_amd.MultimodalAttentionPool(E, num_heads=H)
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(dtype).float()
    q, k, v, dy, dw = rnd(B, T, E), rnd(B, S, E), rnd(B, S, E), rnd(B, T, E), rnd(B, T, S)
    a = pool.attention
    w = [t_.detach().clone() for t_ in (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias)]
    f = O.mha_forward(q, k, v, w[0], w[1], w[2], w[3], H, None)
    b = O.mha_backward(q, k, v, w[0], w[1], w[2], H, f, dy, dw)
    pool = pool.to(DEV, dtype).eval()
    qd, kd, vd = (t_.to(DEV, dtype).requires_grad_(True) for t_ in (q, k, v))
    y, info = pool(qd, kd, vd, return_info=True)
    ((y.float() * dy.to(DEV)).sum() + (info["attention_weights"].float() * dw.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    got = pool.attention.in_proj_bias.grad.cpu().view(3, E)
    want = b["db_in"].view(3, E)
    for i in range(3):
        print("part", i, "\n got ", got[i], "\n want", want[i])
