"""AECF_HILO_GRADS: errors of the float32-stored parameter gradients against float32 math, per block of dW_in (q / k / v rows),
with the flag off and on, at the headline shape; and the step time both ways.  usage: python tools/debug/hilo_errors.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aecf_amd                                        # noqa: E402
from aecf_amd import layer                             # noqa: E402
from oracle import aecf_oracle as O                    # noqa: E402  (diagnostic tool, not a product path)
from tests.helpers import hot_shape_inputs, rel_err    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
d = hot_shape_inputs(71, B=B)
E, H, M = d["E"], d["H"], d["M"]
qe = d["query"].expand(B, -1, -1)
f = O.mha_forward(qe, d["x"], d["x"], d["w_in"], d["b_in"], d["w_out"], d["b_out"], H)
b = O.mha_backward(qe, d["x"], d["x"], d["w_in"], d["b_in"], d["w_out"], H, f, d["dy"], d["dwbar"])


def run(hilo):
    pool = aecf_amd.MultimodalAttentionPool(E, num_heads=H)
    with torch.no_grad():
        pool.attention.in_proj_weight.copy_(d["w_in"]); pool.attention.in_proj_bias.copy_(d["b_in"])
        pool.attention.out_proj.weight.copy_(d["w_out"]); pool.attention.out_proj.bias.copy_(d["b_out"])
    pool = pool.to(dev).train()                        # float32 master parameters, bf16 activations
    pool.options.hilo_grads = hilo
    x = d["x"].to(dev, torch.bfloat16).requires_grad_(True)
    q0 = d["query"].to(dev).requires_grad_(True)
    y, info = pool(q0.to(torch.bfloat16).expand(B, -1, -1), x, return_info=True)
    ((y.float() * d["dy"].to(dev)).sum() + (info["attention_weights"].float() * d["dwbar"].to(dev)).sum()).backward()
    torch.cuda.synchronize()
    a = pool.attention
    g = lambda t_: t_.detach().float().cpu()
    dw = g(a.in_proj_weight.grad)
    out = dict(dW_q=rel_err(dw[:E], b["dw_in"][:E]), dW_k=rel_err(dw[E:2 * E], b["dw_in"][E:2 * E]),
               dW_v=rel_err(dw[2 * E:], b["dw_in"][2 * E:]), dw_in=rel_err(dw, b["dw_in"]),
               db_in=rel_err(g(a.in_proj_bias.grad), b["db_in"]), dw_out=rel_err(g(a.out_proj.weight.grad), b["dw_out"]),
               db_out=rel_err(g(a.out_proj.bias.grad), b["db_out"]), dquery=rel_err(g(q0.grad), b["dquery"].sum(0, keepdim=True)),
               dx=rel_err(g(x.grad), b["dkey"] + b["dvalue"]), y=rel_err(g(y), f["y"]))
    return {k: float("%.3g" % v) for k, v in out.items()}


for hilo in (False, True):
    print("hilo" if hilo else "default", run(hilo), flush=True)
