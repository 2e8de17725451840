"""Standalone timing of the optimiser step of the example model: FusedAdamW (aecf_adamw_step) against torch's fused AdamW."""
import torch
from aecf_amd.optim import FusedAdamW
from aecf_amd.xray import AECFModel

dev = torch.device("cuda:0")


def run(make, label):
    torch.manual_seed(0)
    model = AECFModel(512, 512, 15).to(dev)
    params = list(model.parameters())
    for p in params:
        p.grad = torch.randn_like(p)
    opt = make(params)
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            opt.step()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(label, "us per step (graph of 20 steps):", round(e0.elapsed_time(e1) * 1e3 / 200, 2), "params", sum(p.numel() for p in params),
          "tensors", len(params))


run(lambda ps: FusedAdamW(ps, lr=1e-4, weight_decay=0.01), "aecf FusedAdamW")
run(lambda ps: torch.optim.AdamW(ps, lr=1e-4, weight_decay=0.01, fused=True, capturable=True), "torch fused capturable")


def sweep(make, label):
    for n in (65536, 262144, 1048576, 4194304, 16777216):
        for parts in (1, 16):
            ps = [torch.randn(n // parts, device=dev).requires_grad_() for _ in range(parts)]
            for p in ps:
                p.grad = torch.randn_like(p)
            opt = make(ps)
            for _ in range(3):
                opt.step()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    opt.step()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 100
            print(f"{label} n={n} tensors={parts}: {us:.1f} us  {28 * n / us / 1e6:.2f} TB/s")


sweep(lambda ps: FusedAdamW(ps, lr=1e-4, weight_decay=0.01), "aecf")
sweep(lambda ps: torch.optim.AdamW(ps, lr=1e-4, weight_decay=0.01, fused=True, capturable=True), "torch")
