import torch, time
dev = torch.device("cuda:0")
a = torch.ones((), device=dev); b = torch.ones((), device=dev, dtype=torch.bfloat16)
big = torch.empty(1 << 29, device=dev, dtype=torch.bfloat16)
def t(fn, pre=None, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(n):
        if pre: pre()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
    return tot / n * 1e3
print("f32 + bf16 scalar: %.1f us" % t(lambda: a + b))
print("f32 + f32 scalar: %.1f us" % t(lambda: a + a))
print("f32 + bf16 scalar after a 1 GB write: %.1f us" % t(lambda: a + b, pre=lambda: big.fill_(1.0)))
print("f32 + f32 scalar after a 1 GB write: %.1f us" % t(lambda: a + a, pre=lambda: big.fill_(1.0)))
