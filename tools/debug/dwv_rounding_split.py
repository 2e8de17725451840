"""Where the float32-stored dW_v error of the bf16 path comes from (CPU, fp32 math on bf16 inputs): the rounding of do = dy W_o,
the rounding of the pooled rows, both, and do carried as hi + lo with the pooled rows still rounded."""
import torch, math
torch.manual_seed(0)
B, M, E, H = 8192, 3, 512, 8
hd = E // H
bf = lambda t: t.to(torch.bfloat16).float()
x = bf(torch.randn(B, M, E)); dy = bf(torch.randn(B, E))
w_o = bf(torch.randn(E, E) / E ** 0.5)
p = torch.softmax(torch.randn(B, H, M), -1)
do = dy @ w_o                                    # exact (fp32 math on bf16 inputs)
pe = p.repeat_interleave(hd, 1)                  # [B, E, M] weights per output row j
def dwv(do_, round_pooled):
    out = torch.zeros(E, E)
    for h in range(H):
        pooled = torch.einsum("bm,bmk->bk", p[:, h], x)      # [B, E]
        if round_pooled: pooled = bf(pooled)
        out[h * hd:(h + 1) * hd] = do_[:, h * hd:(h + 1) * hd].t() @ pooled
    return out
ref = dwv(do, False)
rel = lambda a: ((a - ref).abs().max() / ref.abs().max()).item()
print("do bf16, pooled exact :", rel(dwv(bf(do), False)))
print("do exact, pooled bf16 :", rel(dwv(do, True)))
print("both bf16             :", rel(dwv(bf(do), True)))
lo = bf(do - bf(do))
print("do hi+lo, pooled bf16 :", rel(dwv(bf(do) + lo, True)))
