"""CPU oracle for the AECF fusion hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain PyTorch-CPU tensor arithmetic: matmul, exp,
sum, where -- no nn.MultiheadAttention, no autograd) of the algorithm the reference
runs for the path named in BASELINE.json.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``aecf_amd`` never does (it fails loudly when the HIP library is missing).

Parity status: PINNED.  The reference ships no golden vectors of its own (SURVEY.md
section 4), so the oracle is pinned against outputs of the reference itself, imported in
the build container from /root/reference and run on CPU:  ``tests/golden/make_golden.py``
generates ``tests/golden/*.npz`` from the reference and
``tests/test_oracle_golden.py`` checks every function below against them.

Every function cites the reference file:line it follows.  ``ref:`` = /root/reference,
``torch:`` = the installed torch 2.10 sources the reference delegates the attention
arithmetic to (third-party dependency, torch>=2.0 unpinned in ref:requirements.txt:1).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# A4: nn.MultiheadAttention arithmetic (ref: aecf/AECFLayer.py:399-407, 515-521)
# --------------------------------------------------------------------------------------
def split_in_proj(w_in: Tensor, b_in: Optional[Tensor]):
    """torch:nn/functional.py:5836-5852 -- packed [3E,E] weight, q/k/v order."""
    E = w_in.shape[1]
    wq, wk, wv = w_in[:E], w_in[E:2 * E], w_in[2 * E:]
    if b_in is None:
        z = torch.zeros(E, dtype=w_in.dtype)
        return wq, wk, wv, z, z, z
    return wq, wk, wv, b_in[:E], b_in[E:2 * E], b_in[2 * E:]


def mha_forward(
    query: Tensor,            # [B,T,E]  (batch-first)
    key: Tensor,              # [B,S,E]
    value: Tensor,            # [B,S,E]
    w_in: Tensor,             # [3E,E]
    b_in: Optional[Tensor],   # [3E]
    w_out: Tensor,            # [E,E]
    b_out: Optional[Tensor],  # [E]
    num_heads: int,
    key_padding_mask: Optional[Tensor] = None,  # [B,S] bool, True = ignore
    attn_mask: Optional[Tensor] = None,          # [T,S] or [B*H,T,S]; bool (True = blocked) or float additive
    dropout_u: Optional[Tensor] = None,          # [B,H,T,S] uniforms: keep = (u >= dropout_p)   (torch:6591-6592)
    dropout_p: float = 0.0,
) -> Dict[str, Tensor]:
    """Need-weights branch of multi_head_attention_forward, torch:nn/functional.py:6576-6612.

    Returns y [B,T,E], head-averaged weights wbar [B,T,S], per-head probs p [B,H,T,S]
    and the intermediates the explicit backward needs.
    """
    B, T, E = query.shape
    S = key.shape[1]
    H = num_heads
    hd = E // H
    wq, wk, wv, bq, bk, bv = split_in_proj(w_in, b_in)
    qp = query @ wq.T + bq          # torch:5836-5841  (F1)
    kp = key @ wk.T + bk            # torch:5842       (F2)
    vp = value @ wv.T + bv
    # head split, torch:6504-6519 (F4)
    qh = qp.reshape(B, T, H, hd).permute(0, 2, 1, 3)   # [B,H,T,hd]
    kh = kp.reshape(B, S, H, hd).permute(0, 2, 1, 3)
    vh = vp.reshape(B, S, H, hd).permute(0, 2, 1, 3)
    scale = math.sqrt(1.0 / float(hd))                  # torch:6577-6578 (F5)
    scores = (qh * scale) @ kh.transpose(-1, -2)        # torch:6589 (F6)  [B,H,T,S]
    neg_inf = float("-inf")
    if attn_mask is not None:                           # torch:6584-6587
        am = attn_mask.reshape(1, 1, T, S) if attn_mask.dim() == 2 else attn_mask.reshape(B, H, T, S)
        if attn_mask.dtype == torch.bool:
            scores = scores.masked_fill(am, neg_inf)
        else:
            scores = scores + am.to(scores.dtype)
    if key_padding_mask is not None:                    # torch:6554-6566
        scores = scores.masked_fill(key_padding_mask.reshape(B, 1, 1, S), neg_inf)
    m = scores.max(dim=-1, keepdim=True).values
    e = torch.exp(scores - m)
    p = e / e.sum(dim=-1, keepdim=True)                 # torch:6590 (F7)
    keep = None
    pd = p
    if dropout_p > 0.0:                                 # torch:6591-6592: the weights are dropped out, then used AND returned
        keep = (dropout_u.reshape(B, H, T, S) >= dropout_p).to(p.dtype) / (1.0 - dropout_p)
        pd = p * keep
    oh = pd @ vh                                        # torch:6594 (F9)  [B,H,T,hd]
    o = oh.permute(0, 2, 1, 3).reshape(B, T, E)         # torch:6596-6599 (F10)
    y = o @ w_out.T                                     # torch:6600 (F11)
    if b_out is not None:
        y = y + b_out
    wbar = pd.mean(dim=1)                               # torch:6604-6606 (F12) [B,T,S]
    return dict(y=y, wbar=wbar, p=p, pd=pd, keep=keep, o=o, qp=qp, kp=kp, vp=vp)


def mha_backward(
    query: Tensor, key: Tensor, value: Tensor,
    w_in: Tensor, b_in: Optional[Tensor], w_out: Tensor,
    num_heads: int, fwd: Dict[str, Tensor],
    dy: Tensor,                      # [B,T,E]
    dwbar: Optional[Tensor] = None,  # [B,T,S] gradient on info['attention_weights']
) -> Dict[str, Tensor]:
    """Closed-form transpose of mha_forward (what autograd does for the reference, A10).

    Returns dquery, dkey, dvalue (callers add dkey+dvalue when key is value), dw_in, db_in,
    dw_out, db_out.
    """
    B, T, E = query.shape
    S = key.shape[1]
    H = num_heads
    hd = E // H
    wq, wk, wv, _, _, _ = split_in_proj(w_in, b_in)
    scale = math.sqrt(1.0 / float(hd))
    p, o, qp, kp, vp = fwd["p"], fwd["o"], fwd["qp"], fwd["kp"], fwd["vp"]
    dy2 = dy.reshape(B * T, E)
    dw_out = dy2.T @ o.reshape(B * T, E)
    db_out = dy2.sum(0)
    do = dy @ w_out                                        # [B,T,E]
    doh = do.reshape(B, T, H, hd).permute(0, 2, 1, 3)      # [B,H,T,hd]
    vh = vp.reshape(B, S, H, hd).permute(0, 2, 1, 3)
    kh = kp.reshape(B, S, H, hd).permute(0, 2, 1, 3)
    qh = qp.reshape(B, T, H, hd).permute(0, 2, 1, 3)
    dp = doh @ vh.transpose(-1, -2)                        # [B,H,T,S]
    if dwbar is not None:
        dp = dp + dwbar.unsqueeze(1) / H                   # mean over heads
    pd, keep = fwd.get("pd", p), fwd.get("keep")
    dvh = pd.transpose(-1, -2) @ doh                       # [B,H,S,hd]
    if keep is not None:
        dp = dp * keep                                     # back through the dropout
    ds = p * (dp - (p * dp).sum(-1, keepdim=True))         # softmax backward
    dqh = (ds @ kh) * scale
    dkh = ds.transpose(-1, -2) @ (qh * scale)
    dqp = dqh.permute(0, 2, 1, 3).reshape(B * T, E)
    dkp = dkh.permute(0, 2, 1, 3).reshape(B * S, E)
    dvp = dvh.permute(0, 2, 1, 3).reshape(B * S, E)
    dquery = (dqp @ wq).reshape(B, T, E)
    dkey = (dkp @ wk).reshape(B, S, E)
    dvalue = (dvp @ wv).reshape(B, S, E)
    dw_in = torch.cat([dqp.T @ query.reshape(B * T, E),
                       dkp.T @ key.reshape(B * S, E),
                       dvp.T @ value.reshape(B * S, E)], 0)
    db_in = torch.cat([dqp.sum(0), dkp.sum(0), dvp.sum(0)], 0)
    return dict(dquery=dquery, dkey=dkey, dvalue=dvalue, dw_in=dw_in, db_in=db_in,
                dw_out=dw_out, db_out=db_out)


# --------------------------------------------------------------------------------------
# A5: CurriculumMasking.forward (ref: aecf/AECFLayer.py:130-283)
# --------------------------------------------------------------------------------------
def entropy_rows(w: Tensor) -> Tensor:
    """ref: aecf/AECFLayer.py:113-128 -- -sum xlogy(w,w), clamped to [0, log L]."""
    L = w.shape[-1]
    safe = torch.where(w == 0, torch.ones_like(w), w)
    xlogx = torch.where(w == 0, torch.zeros_like(w), w * torch.log(safe))
    xlogx = torch.where(torch.isnan(w), w, xlogx)          # xlogy propagates NaN in x
    h = -xlogx.sum(-1)
    return h.clamp(0.0, math.log(L))


def curriculum_mask_train(
    weights: Tensor,          # [..., L]
    uniforms: Tensor,         # [..., L] float32 U[0,1): the draws torch.bernoulli consumes (ref :204)
    base_mask_prob: float = 0.15,
    entropy_target: float = 0.7,
    min_active: int = 1,
    eps: float = 1e-8,
) -> Dict[str, Tensor]:
    """Training-mode branch, ref: aecf/AECFLayer.py:158-283, with the Bernoulli draw made
    explicit: torch.bernoulli(p) on CPU == (rand_like(p, float32) < p).
    Ties in the min-active top-k resolve to the lowest index (what torch.topk does on CPU
    for k=1; k>=2 ties are unspecified in torch and not pinned).
    """
    L = weights.shape[-1]
    dt = weights.dtype
    if L <= 1:                                             # ref :160-167
        z = torch.zeros(weights.shape[:-1], dtype=dt)
        return dict(masked=weights, mask=torch.ones_like(weights), entropy=z, mask_rate=z,
                    target_entropy=z, weights_norm=weights)
    w = weights
    sums = w.sum(-1, keepdim=True)                         # ref :170
    finite = torch.isfinite(w)
    if not bool(finite.all()):                             # ref :173-176
        w = torch.where(finite, w, torch.zeros_like(w))
        sums = w.sum(-1, keepdim=True)
    needs_norm = sums < eps                                # ref :178
    w = torch.where(needs_norm, torch.full_like(w, 1.0 / L), w / sums)   # ref :179-184
    ent = entropy_rows(w)                                  # ref :190
    max_ent = math.log(float(L))
    norm_ent = (ent / max_ent).clamp(0.0, 1.0)             # ref :192
    keep = (1.0 - base_mask_prob * norm_ent).unsqueeze(-1).clamp(0.0, 1.0)   # ref :197-201
    mask = (uniforms < keep.to(torch.float32)).to(dt)      # ref :204
    k = min(min_active, L)                                 # ref :207
    needs_more = mask.sum(-1) < k                          # ref :208-209
    if bool(needs_more.any()):                             # ref :211-260
        # stable descending sort == lowest index first among equal values
        order = torch.sort(w, dim=-1, descending=True, stable=True).indices[..., :k]
        min_mask = torch.zeros_like(w).scatter(-1, order, 1.0)
        mask = torch.where(needs_more.unsqueeze(-1), min_mask, mask)
    masked = w * mask                                      # ref :263
    s = masked.sum(-1, keepdim=True)                       # ref :264
    final = torch.where(s > eps, masked / s, w)            # ref :267-272
    mask_rate = 1.0 - mask.float().mean(-1)                # ref :275 (always float32)
    target = torch.full_like(ent, max_ent * entropy_target)   # ref :280
    return dict(masked=final, mask=mask, entropy=ent, mask_rate=mask_rate,
                target_entropy=target, weights_norm=w, keep=keep.squeeze(-1))


def curriculum_mask_train_backward(weights: Tensor, mask: Tensor, d_masked: Tensor, eps: float = 1e-8) -> Tensor:
    """Gradient of curriculum_mask_train()['masked'] w.r.t. ``weights`` (what autograd gives the
    reference for the stand-alone module, ref :170-184, :263-272; entropy/keep-prob carry no gradient:
    the Bernoulli draw cuts the graph and the info entries are detached, ref :277-281)."""
    finite = torch.isfinite(weights)
    w = torch.where(finite, weights, torch.zeros_like(weights))
    s = w.sum(-1, keepdim=True)
    needs_norm = s < eps
    wn = w / s
    ms = (wn * mask).sum(-1, keepdim=True)
    final = wn * mask / ms
    dot = (d_masked * final).sum(-1, keepdim=True)
    dwn = torch.where(ms > eps, mask * (d_masked - dot) / ms, d_masked)
    dot2 = (dwn * wn).sum(-1, keepdim=True)
    dw = (dwn - dot2) / s
    return torch.where(needs_norm | ~finite, torch.zeros_like(dw), dw)


def curriculum_mask_eval(weights: Tensor) -> Dict[str, Tensor]:
    """Eval-mode branch, ref: aecf/AECFLayer.py:150-156: weights unchanged, no target key."""
    ent = entropy_rows(weights)
    return dict(masked=weights, entropy=ent, mask_rate=torch.zeros_like(ent))


def entropy_rows_backward(w: Tensor, dent: Tensor) -> Tensor:
    """d/dw of entropy_rows (eval mode keeps the entropy attached, ref :150-156):
    -(log w + 1) where w>0 and the clamp is inactive, else 0 (xlogy grad at 0 is -inf*0 in
    torch; rows with exact zeros are not pinned)."""
    L = w.shape[-1]
    raw = -torch.where(w == 0, torch.zeros_like(w), w * torch.log(torch.where(w == 0, torch.ones_like(w), w))).sum(-1)
    live = ((raw >= 0.0) & (raw <= math.log(L))).to(w.dtype).unsqueeze(-1)
    return -(torch.log(w) + 1.0) * dent.unsqueeze(-1) * live


# --------------------------------------------------------------------------------------
# A6: CurriculumMasking.entropy_loss (ref: aecf/AECFLayer.py:285-314)
# --------------------------------------------------------------------------------------
def entropy_loss(entropy: Tensor, last_seq_len: int = 2, entropy_target: float = 0.7) -> Tensor:
    if not bool(torch.isfinite(entropy).all()):            # ref :295-296
        entropy = torch.nan_to_num(entropy, nan=0.0, posinf=1.0, neginf=0.0)
    max_ent = math.log(float(last_seq_len)) if last_seq_len > 1 else 0.0   # ref :307
    target = max_ent * entropy_target
    d = entropy - target
    return (d * d).mean().clamp(min=0.0)                   # ref :311-314


def entropy_loss_backward(entropy: Tensor, last_seq_len: int = 2, entropy_target: float = 0.7,
                          dloss: float = 1.0) -> Tensor:
    finite = torch.isfinite(entropy)
    e = torch.nan_to_num(entropy, nan=0.0, posinf=1.0, neginf=0.0) if not bool(finite.all()) else entropy
    max_ent = math.log(float(last_seq_len)) if last_seq_len > 1 else 0.0
    g = 2.0 * (e - max_ent * entropy_target) / entropy.numel() * dloss
    return torch.where(finite, g, torch.zeros_like(g))     # nan_to_num has zero grad at replaced entries


# --------------------------------------------------------------------------------------
# A7: projection-free single-head attention (ref: aecf/AECFLayer.py:556-581)
# --------------------------------------------------------------------------------------
def sdpa(query: Tensor, key: Tensor, value: Tensor, scale: Optional[float] = None) -> Tensor:
    if scale is None:
        scale = query.shape[-1] ** -0.5                    # ref :573-574
    scores = (query @ key.transpose(-2, -1)) * scale       # ref :577
    m = scores.max(-1, keepdim=True).values
    e = torch.exp(scores - m)
    p = e / e.sum(-1, keepdim=True)                        # ref :578
    return p @ value                                       # ref :581


def sdpa_backward(query: Tensor, key: Tensor, value: Tensor, dout: Tensor,
                  scale: Optional[float] = None) -> Tuple[Tensor, Tensor, Tensor]:
    if scale is None:
        scale = query.shape[-1] ** -0.5
    scores = (query @ key.transpose(-2, -1)) * scale
    p = torch.softmax(scores, -1)
    dv = p.transpose(-2, -1) @ dout
    dp = dout @ value.transpose(-2, -1)
    ds = p * (dp - (p * dp).sum(-1, keepdim=True)) * scale
    return ds @ key, ds.transpose(-2, -1) @ query, dv


# --------------------------------------------------------------------------------------
# A3: MultimodalAttentionPool.forward composed (ref: aecf/AECFLayer.py:409-547)
# --------------------------------------------------------------------------------------
def pool_forward_train(query, x, w_in, b_in, w_out, b_out, num_heads, uniforms,
                       base_mask_prob=0.15, entropy_target=0.7, min_active=1,
                       key_padding_mask=None):
    """pool(query, x, return_info=True) in train mode with curriculum masking attached."""
    f = mha_forward(query, x, x, w_in, b_in, w_out, b_out, num_heads, key_padding_mask)
    m = curriculum_mask_train(f["wbar"], uniforms, base_mask_prob, entropy_target, min_active)
    return f, m


# --------------------------------------------------------------------------------------
# A9: contrastive term named by north_star.  NOT in the reference:
# build-defined symmetric InfoNCE; PARITY UNPINNED (no reference to pin against).
# --------------------------------------------------------------------------------------
def info_nce(za: Tensor, zb: Tensor, temperature: float = 0.07) -> Tensor:
    """Symmetric InfoNCE on L2-normalised rows: 0.5*(CE(za zb^T / t) + CE(zb za^T / t))."""
    na = za / za.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    nb = zb / zb.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    logits = na @ nb.T / temperature
    idx = torch.arange(za.shape[0])
    lse_r = torch.logsumexp(logits, 1)
    lse_c = torch.logsumexp(logits, 0)
    diag = logits[idx, idx]
    return 0.5 * ((lse_r - diag).mean() + (lse_c - diag).mean())


def info_nce_backward(za: Tensor, zb: Tensor, temperature: float = 0.07) -> Tuple[Tensor, Tensor]:
    n = za.shape[0]
    ra = za.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    rb = zb.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    na, nb = za / ra, zb / rb
    logits = na @ nb.T / temperature
    pr = torch.softmax(logits, 1)
    pc = torch.softmax(logits, 0)
    g = (0.5 / n) * (pr + pc - 2.0 * torch.eye(n, dtype=za.dtype)) / temperature   # dL/dlogits... scaled
    dna = g @ nb
    dnb = g.T @ na
    dza = (dna - na * (dna * na).sum(-1, keepdim=True)) / ra
    dzb = (dnb - nb * (dnb * nb).sum(-1, keepdim=True)) / rb
    return dza, dzb
