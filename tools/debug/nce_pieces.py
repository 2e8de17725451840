"""Checks every piece of the tile-GEMM InfoNCE against torch on the GPU: E, sums, W, da, db (debug aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aecf_amd import _lib
from aecf_amd.layer import _ptr, _stream

def rel(g, w):
    return float((g.double() - w.double()).abs().max() / w.double().abs().max().clamp_min(1e-30))

def main():
    rows, cols, off, d = (int(x) for x in (sys.argv[1:5] + ["256", "256", "0", "128"][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(1)
    nrm = lambda t: (t / t.norm(dim=-1, keepdim=True)).to(torch.bfloat16)
    b = nrm(torch.randn(cols, d, device=dev, generator=g))
    a = nrm(0.8 * b[off:off + rows].float() + 0.6 * nrm(torch.randn(rows, d, device=dev, generator=g)).float())
    T = 0.07
    coef = 0.5 / cols
    Rp, Cp = (rows + 255) // 256 * 256, (cols + 255) // 256 * 256
    ws_bytes = lib.aecf_nce_sym_workspace_bytes(rows, cols, d)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    cs = torch.empty(cols, **f32)
    _lib.check(lib.aecf_nce_sym_pass1(rows, cols, d, T, _ptr(a), _ptr(b), _ptr(ws), ws_bytes, _ptr(cs), _stream()), "p1")
    torch.cuda.synchronize()
    E = ws[:Rp * Cp * 2].view(torch.bfloat16).view(Rp // 256, Cp // 64, 256, 64).permute(0, 2, 1, 3).reshape(Rp, Cp).clone()
    S = a.float() @ b.float().T
    Eref = torch.exp((S - 1.0) / T)
    print("E valid block rel err", rel(E[:rows, :cols].float(), Eref), " padding max", float(E[rows:].abs().max()) if Rp > rows else 0.0,
          float(E[:, cols:].abs().max()) if Cp > cols else 0.0)
    bad = ((E[:rows, :cols].float() - Eref).abs() > 0.02 * Eref.abs().max()).nonzero()
    print("  bad E entries", bad.shape[0], bad[:8].tolist())
    print("col sums rel err", rel(cs, Eref.sum(0)))
    lr, da, db = torch.empty(rows, **f32), torch.empty(rows, d, **f32), torch.empty(cols, d, **f32)
    _lib.check(lib.aecf_nce_sym_loss(rows, cols, off, d, T, _ptr(a), _ptr(b), _ptr(cs), _ptr(ws), ws_bytes, _ptr(lr), 0, 2, 0.0, None,
                                     1.0, None, None, _stream()), "loss")
    _lib.check(lib.aecf_nce_sym_grads(rows, cols, off, d, T, coef, _ptr(a), _ptr(b), _ptr(ws), ws_bytes, None, _lib.AECF_F32, _ptr(da),
                                      _ptr(db), _stream()), "grads")
    torch.cuda.synchronize()
    W = ws[:Rp * Cp * 2].view(torch.bfloat16).view(Rp // 256, Cp // 64, 256, 64).permute(0, 2, 1, 3).reshape(Rp, Cp).clone().float()
    l, c = Eref.sum(1), Eref.sum(0)
    Wref = Eref * (1 / l[:, None] + 1 / c[None, :])
    idx = torch.arange(rows, device=dev)
    Wref[idx, off + idx] -= 2.0
    Wref *= coef / T
    print("W rel err", rel(W[:rows, :cols], Wref))
    da_w = W[:rows, :cols] @ b.float()
    db_w = W[:rows, :cols].T @ a.float()
    print("da vs (stored W) @ b", rel(da, da_w), "   db vs (stored W)^T @ a", rel(db, db_w))
    e = (da - da_w).abs()
    print("  da worst rows", e.max(1).values.topk(5).indices.tolist(), "worst cols", e.max(0).values.topk(5).indices.tolist())
    e = (db - db_w).abs()
    print("  db worst rows", e.max(1).values.topk(5).indices.tolist(), "worst cols", e.max(0).values.topk(5).indices.tolist())
    # structure of the da error: which (row % 16, col % 16) are off
    e = ((da - da_w).abs() > 1e-2 * da_w.abs().max()).float()
    print("  da wrong fraction", float(e.mean()), "by col%16", [round(float(e[:, k::16].mean()), 2) for k in range(16)])
    print("  da wrong by row%16", [round(float(e[k::16].mean()), 2) for k in range(16)])

main()
