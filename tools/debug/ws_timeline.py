"""Phase timeline of vproj_slab_kernel at C2 from in-kernel shader-clock stamps (experiment build -DAECF_WS_TIMELINE:
a scratch checkout with tools/micro/variants/aecf_gemm_ws_ablations.hip (the round-4 source that carries the stamps) in place of aecf_gemm_ws.hip, built with -DAECF_WS_TIMELINE; AECF_LIB_PATH=build/var/wtl/libaecf_hip.so).
Stamps per 16-sample step: 0 top, 1 after (lgkmcnt + barrier), 2 after the softmax, 3 after the MFMA loop, 4 after vmcnt(0),
5 after the next step's partial scores, 6 after the stores."""
import ctypes
import sys

import numpy as np
import torch

import aecf_amd
from aecf_amd import _lib

dev = torch.device("cuda:0")
B, M, E, H = 65536, 3, 512, 8
torch.manual_seed(0)
query, pool = aecf_amd.create_fusion_pool(E, M, mask_prob=0.15, num_heads=H)
pool = pool.to(dev, torch.bfloat16).train()
query = torch.nn.Parameter(query.detach().to(dev, torch.bfloat16))
x = torch.randn(B, M, E, device=dev).to(torch.bfloat16).requires_grad_()
for _ in range(3):
    out, info = pool(query.expand(B, -1, -1), x, return_info=True)
    out.float().square().mean().backward()
torch.cuda.synchronize()
lib = _lib.load()
n = 2 * 8 * 64 * 8
buf = (ctypes.c_ulonglong * n)()
if not hasattr(lib, "aecf_debug_ws_timeline"):
    sys.exit("library built without -DAECF_WS_TIMELINE")
fn = lib.aecf_debug_ws_timeline
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = fn(buf, n)
tall = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(2, 8, 64, 8)
# ---- dsu_ws_kernel: stamps 0 top, 1 after barrier A, 2 P + dot done, 3 after (lgkm + barrier B), 4 softmax backward done,
#      5 after (lgkm + barrier C), 6 u product issued, 7 after vmcnt(0)
dn = ["barrier A", "copy issue + P MFMAs + dot", "lgkm + barrier B", "softmax backward", "lgkm + barrier C", "u product", "wait vmcnt"]
for sel in range(8):
    tt = tall[1][sel]
    steps = int((tt[:, 0] > 0).sum())
    if steps < 4:
        continue
    tt = tt[:steps]
    d = np.diff(tt, axis=1)
    step_len = tt[1:, 0] - tt[:-1, 0]
    print(f"dsu_ws block sel {sel // 4} wave {2 * (sel % 4)}: {steps} steps, mean step {step_len[1:].mean():.0f} cycles:",
          " | ".join(f"{dn[i]} {d[2:-1, i].mean():.0f}" for i in range(7)))
t = tall[0]
names = ["lgkm + barrier", "softmax", "MFMA loop", "wait vmcnt", "next scores", "stores"]
for sel in range(8):
    tt = t[sel]
    steps = int((tt[:, 0] > 0).sum())
    if steps < 4:
        continue
    tt = tt[:steps]
    d = np.diff(tt[:, :7], axis=1)
    step_len = tt[1:, 0] - tt[:-1, 0]
    print(f"vproj_slab block sel {sel // 4} wave {2 * (sel % 4) + 1}: {steps} steps, mean step {step_len[1:].mean():.0f} cycles:",
          " | ".join(f"{names[i]} {d[2:-1, i].mean():.0f}" for i in range(6)), f"| loop back {(tt[1:, 0] - tt[:-1, 6])[1:].mean():.0f}")
print("rc", rc, "first->last stamp, block 0 wave 1:", int(t[0, :, 6].max() - t[0, 0, 0]))
