"""How long does the vendor GEMM (torch.nn.functional.linear -> hipBLASLt) take for the path's two plain products?"""
import torch, time
dev = torch.device("cuda:0")
for (R, N, K) in [(65536, 512, 512), (16384, 1024, 1024), (8192, 768, 768), (65536 * 3, 512, 512)]:
    a = torch.randn(R, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for _ in range(20):
        torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50):
        torch.nn.functional.linear(a, w, b)
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / 50 * 1e3
    print(R, N, K, "%.1f us" % us, "%.0f TFLOP/s" % (2.0 * R * N * K / us / 1e6))
