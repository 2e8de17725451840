"""Phase timeline of gemm_tn_tr_wide_kernel (dW_v) at C2 from in-kernel shader-clock stamps (experiment build
-DAECF_TN_TIMELINE: a scratch checkout with tools/micro/variants/aecf_gemm_tn_tr_ablations.hip (the round-4 source that carries the stamps) in place of aecf_gemm_tn_tr.hip, built with -DAECF_TN_TIMELINE; run with
AECF_LIB_PATH=build/var/tl/libaecf_hip.so).  Stamps per step: 0 top, 1 after vmcnt(0), 2 probs in LDS (lgkmcnt), 3 after barrier 1,
4 pooling issued, 5 pooled stores done (lgkmcnt), 6 after barrier 2, 7 after the MFMA phase."""
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
n = 8 * 64 * 8
buf = (ctypes.c_ulonglong * n)()
fn = lib.aecf_debug_tn_timeline if hasattr(lib, "aecf_debug_tn_timeline") else None
if fn is None:
    sys.exit("library built without -DAECF_TN_TIMELINE")
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = fn(buf, n)
t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(8, 64, 8)
names = ["wait vmcnt", "probs->LDS", "barrier 1", "issue+unpack+pool", "wait pooled stores", "barrier 2", "tr reads + MFMA", "loop back"]
for sel in range(8):
    tt = t[sel]
    steps = int((tt[:, 0] > 0).sum())
    if steps < 3:
        continue
    tt = tt[:steps]
    d = np.diff(tt, axis=1)                                  # [steps, 7]
    back = tt[1:, 0] - tt[:-1, 7]
    step_len = tt[1:, 0] - tt[:-1, 0]
    print(f"block sel {sel // 4} wave {5 * (sel % 4)}: {steps} steps, mean step {step_len[1:].mean():.0f} cycles; phases (mean over steps 2..):",
          " | ".join(f"{names[i]} {d[2:, i].mean():.0f}" for i in range(7)), f"| loop back {back[1:].mean():.0f}")
print("rc", rc, "total cycles first->last stamp, block 0 wave 0:", int(t[0, :, 7].max() - t[0, 0, 0]))
