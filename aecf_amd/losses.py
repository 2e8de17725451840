"""The loss side of the AECF objective: the entropy regulariser of the reference
(``CurriculumMasking.entropy_loss``, ref aecf/AECFLayer.py:285-314) plus the contrastive term that
BASELINE.json's north_star names.

The contrastive term does NOT exist in the reference; it is build-defined here as
the symmetric InfoNCE of two views' fused embeddings, L2-normalised, with cross-batch negatives all-gathered
over the data-parallel group:

    L = 0.5 / B_all * sum_i [ CE(za_i . zb_all / T, i) + CE(zb_i . za_all / T, i) ]

All arithmetic (normalisation, logits GEMM, softmax / loss rows, both gradient GEMMs) runs in libaecf_hip.so;
the exchange steps are ``dp.all_gather_rows`` (forward) and its reduce-scatter (backward).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib, dp
from .layer import _DTYPES, _ptr, _require_device, _stream, CurriculumMasking


def _plus(total: torch.Tensor, term: torch.Tensor) -> torch.Tensor:
    """total + term for two loss scalars whose dtypes may differ (a float32 InfoNCE term, a bf16 entropy regulariser): the
    addend is brought to the sum's dtype first.  torch's mixed-dtype elementwise kernel (the run-time cast variant) takes ~41 us
    for ONE element on this ROCm build against ~7 us for a same-dtype add (tools/debug/scalar_add_time.py) -- 1.4 % of the
    configs[2] step for a scalar."""
    if term.dtype != total.dtype and term.dim() == 0 and total.dim() == 0:
        term = term.to(torch.promote_types(total.dtype, term.dtype))
        total = total.to(term.dtype)
    return total + term


class _L2Norm(torch.autograd.Function):
    """aecf_l2norm_forward / _backward: rows -> unit norm."""

    @staticmethod
    def forward(ctx, z, eps):
        lib = _lib.load()
        n, d = z.shape
        zc = z.contiguous()
        zn = torch.empty_like(zc)
        inv = torch.empty(n, dtype=torch.float32, device=z.device)
        _lib.check(lib.aecf_l2norm_forward(n, d, _DTYPES[z.dtype], eps, _ptr(zc), _ptr(zn), _ptr(inv), _stream()),
                   "aecf_l2norm_forward")
        ctx.save_for_backward(zn, inv)
        return zn

    @staticmethod
    def backward(ctx, dzn):
        lib = _lib.load()
        zn, inv = ctx.saved_tensors
        n, d = zn.shape
        g = dzn.to(torch.float32).contiguous()
        dz = torch.empty_like(zn)
        _lib.check(lib.aecf_l2norm_backward(n, d, _DTYPES[zn.dtype], _ptr(zn), _ptr(inv), _ptr(g), _ptr(dz), _stream()),
                   "aecf_l2norm_backward")
        return dz, None


class _NceDirection(torch.autograd.Function):
    """aecf_nce_fwd_bwd: sum_i [logsumexp_j(q_i.k_j/T) - q_i.k_{off+i}/T] * coef for local unit-norm q against all k.
    Forward and both gradients come out of the same call (the gradients are linear in the upstream scalar)."""

    @staticmethod
    def forward(ctx, q, k_all, row_offset, temperature, coef, low_memory=False):
        lib = _lib.load()
        rows, d = q.shape
        cols = k_all.shape[0]
        dt = q.dtype
        qc, kc = q.contiguous(), k_all.to(dt).contiguous()
        dev = q.device
        loss_rows = torch.empty(rows, dtype=torch.float32, device=dev)
        dq = torch.empty(rows, d, dtype=torch.float32, device=dev)
        dk = torch.empty(cols, d, dtype=torch.float32, device=dev)
        # the workspace handed over selects the implementation (include/aecf_hip.h): rows x cols bf16 -> tile GEMMs,
        # O(rows d) -> the streaming form
        ws_bytes = lib.aecf_nce_stream_workspace_bytes(rows, cols, d, _DTYPES[dt]) if low_memory else 0
        if ws_bytes == 0:
            if low_memory:
                raise NotImplementedError(f"aecf_amd: no streaming InfoNCE for dtype {dt}, d = {d}")
            ws_bytes = lib.aecf_nce_workspace_bytes(rows, cols, d, _DTYPES[dt])
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.aecf_nce_fwd_bwd(rows, cols, row_offset, d, _DTYPES[dt], temperature, coef, _ptr(qc), _ptr(kc),
                                        _ptr(loss_rows), _ptr(dq), _ptr(dk), _ptr(ws), ws_bytes, _stream()),
                   "aecf_nce_fwd_bwd")
        ctx.save_for_backward(dq, dk)
        ctx.dtypes = (q.dtype, k_all.dtype)
        return loss_rows.sum() * coef

    @staticmethod
    def backward(ctx, dloss):
        dq, dk = ctx.saved_tensors
        g = dloss.to(torch.float32)
        return (dq * g).to(ctx.dtypes[0]), (dk * g).to(ctx.dtypes[1]), None, None, None, None


class _NceSymmetric(torch.autograd.Function):
    """aecf_nce_sym_pass1 / _loss / _grads: BOTH directions of the symmetric InfoNCE from one block of logits (local rows of view a
    against the gathered rows of view b).  The column sums of the exponentials are the one thing ranks exchange (one
    all-reduce of `cols` floats between pass 1 and the loss); the gradient on the gathered keys is this rank's share (the
    caller's all-gather backward reduce-scatters it).  The forward runs the logits pass and the loss; the two gradient products
    run in the backward, scaled on the device by the gradient that arrives there and written in the inputs' dtype (no float32
    [cols, d] intermediate, no multiply / cast passes).  Optionally carries CurriculumMasking.entropy_loss (ref
    aecf/AECFLayer.py:285-314) in the loss launch.  Returns (this rank's rows' share of the loss, entropy loss)."""

    @staticmethod
    def forward(ctx, a, b_all, entropy, row_offset, temperature, coef, group, last_seq_len, entropy_target):
        lib = _lib.load()
        rows, d = a.shape
        cols = b_all.shape[0]
        dev = a.device
        ac, bc = a.detach().to(torch.bfloat16).contiguous(), b_all.detach().to(torch.bfloat16).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        ws_bytes = lib.aecf_nce_sym_workspace_bytes(rows, cols, d)
        if ws_bytes == 0:
            raise NotImplementedError(f"aecf_amd: symmetric InfoNCE needs d % 64 == 0, got d = {d}")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        col_sums = torch.empty(cols, **f32)
        _lib.check(lib.aecf_nce_sym_pass1(rows, cols, d, temperature, _ptr(ac), _ptr(bc), _ptr(ws), ws_bytes, _ptr(col_sums),
                                          _stream()), "aecf_nce_sym_pass1")
        if dp.world_info(group)[1] > 1:
            torch.distributed.all_reduce(col_sums, group=group)
        loss_rows = torch.empty(rows, **f32)
        if entropy is not None:
            ent = entropy.detach().to(torch.float32).contiguous().reshape(-1)
            ent_loss, dent = torch.zeros(1, **f32), torch.empty(ent.numel(), **f32)
            n_ent, p_ent, p_el, p_de = ent.numel(), _ptr(ent), _ptr(ent_loss), _ptr(dent)
        else:
            ent_loss, dent, n_ent, p_ent, p_el, p_de = torch.zeros(1, **f32), None, 0, None, None, None
        _lib.check(lib.aecf_nce_sym_loss(rows, cols, row_offset, d, temperature, _ptr(ac), _ptr(bc), _ptr(col_sums), _ptr(ws),
                                         ws_bytes, _ptr(loss_rows), n_ent, last_seq_len, entropy_target, p_ent, 1.0, p_el, p_de,
                                         _stream()), "aecf_nce_sym_loss")
        ctx.save_for_backward(ac, bc, ws, *([dent] if dent is not None else []))
        ctx.meta = (a.dtype, b_all.dtype, None if entropy is None else (entropy.dtype, entropy.shape),
                    (rows, cols, int(row_offset), d, float(temperature), float(coef), ws_bytes))
        return loss_rows.sum() * coef, ent_loss.reshape(())

    @staticmethod
    def backward(ctx, d_nce, d_ent):
        lib = _lib.load()
        ac, bc, ws = ctx.saved_tensors[:3]
        ad, bd, em, (rows, cols, row_offset, d, temperature, coef, ws_bytes) = ctx.meta
        if getattr(ctx, "_spent", False):
            raise RuntimeError("aecf_amd: the symmetric InfoNCE backward runs once per forward (it consumes the stored logits)")
        ctx._spent = True
        gdt = torch.bfloat16 if (ad == torch.bfloat16 and bd == torch.bfloat16) else torch.float32
        da = torch.empty(rows, d, dtype=gdt, device=ac.device)
        db = torch.empty(cols, d, dtype=gdt, device=ac.device)
        up = d_nce.detach().to(torch.float32).reshape(1).contiguous()
        _lib.check(lib.aecf_nce_sym_grads(rows, cols, row_offset, d, temperature, coef, _ptr(ac), _ptr(bc), _ptr(ws), ws_bytes,
                                          _ptr(up), _DTYPES[gdt], _ptr(da), _ptr(db), _stream()), "aecf_nce_sym_grads")
        g_ent = None
        if em is not None:
            g_ent = (ctx.saved_tensors[3] * d_ent.to(torch.float32)).reshape(em[1]).to(em[0])
        return da.to(ad), db.to(bd), g_ent, None, None, None, None, None, None


def _sym_supported(z: torch.Tensor, temperature: float, cols: Optional[int] = None) -> bool:
    """The tile-GEMM form applies (bf16, d % 64 == 0, 1/T a safe exponent shift) AND its workspace -- rows x cols bf16 -- fits
    comfortably in what the device has free; otherwise the callers take the streaming kernels (O(rows d) workspace)."""
    if not (z.dtype == torch.bfloat16 and z.shape[1] % 64 == 0 and temperature >= 0.025):
        return False
    if cols is not None and z.is_cuda:
        need = _lib.load().aecf_nce_sym_workspace_bytes(z.shape[0], cols, z.shape[1])
        free, _ = torch.cuda.mem_get_info(z.device)
        if need > 0.6 * free:
            return False
    return True


def _stream_form(z: torch.Tensor, cols: int) -> bool:
    """True when the streaming InfoNCE kernels (no [rows, cols] logits in memory) exist for z's dtype and width."""
    if z.dtype not in _DTYPES or not z.is_cuda:
        return False
    return _lib.load().aecf_nce_stream_workspace_bytes(z.shape[0], cols, z.shape[1], _DTYPES[z.dtype]) > 0


class _LossDirection(torch.autograd.Function):
    """aecf_loss_fwd_bwd: ONE call for one InfoNCE direction (streaming form: no [rows, cols] logits) AND the entropy
    regulariser of the reference (CurriculumMasking.entropy_loss, ref aecf/AECFLayer.py:285-314) with their gradients
    -- BASELINE.json north_star's "second fused kernel".  Returns (contrastive share, entropy loss)."""

    @staticmethod
    def forward(ctx, q, k_all, entropy, row_offset, temperature, coef, last_seq_len, entropy_target):
        lib = _lib.load()
        rows, d = q.shape
        cols = k_all.shape[0]
        dev = q.device
        qc, kc = q.detach().to(torch.bfloat16).contiguous(), k_all.detach().to(torch.bfloat16).contiguous()
        ent = entropy.detach().to(torch.float32).contiguous().reshape(-1)
        f32 = dict(dtype=torch.float32, device=dev)
        loss_rows, dq, dk = torch.empty(rows, **f32), torch.empty(rows, d, **f32), torch.empty(cols, d, **f32)
        ent_loss, dent = torch.empty(1, **f32), torch.empty(ent.numel(), **f32)
        # this operator is the fallback of the symmetric tile-GEMM form (too little free memory for rows x cols bf16, or a
        # temperature its constant-shift softmax does not take): hand over the STREAMING workspace (O(rows d)) where that
        # kernel exists, so that the fallback never allocates more than the form it replaces
        ws_bytes = lib.aecf_nce_stream_workspace_bytes(rows, cols, d, _lib.AECF_BF16)
        if ws_bytes == 0:
            ws_bytes = lib.aecf_nce_workspace_bytes(rows, cols, d, _lib.AECF_BF16)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.aecf_loss_fwd_bwd(rows, cols, row_offset, d, temperature, coef, _ptr(qc), _ptr(kc), _ptr(loss_rows),
                                         _ptr(dq), _ptr(dk), ent.numel(), last_seq_len, entropy_target, _ptr(ent), 1.0,
                                         _ptr(ent_loss), _ptr(dent), _ptr(ws), ws_bytes, _stream()), "aecf_loss_fwd_bwd")
        ctx.save_for_backward(dq, dk, dent)
        ctx.meta = (q.dtype, k_all.dtype, entropy.dtype, entropy.shape)
        return loss_rows.sum() * coef, ent_loss.reshape(())

    @staticmethod
    def backward(ctx, d_nce, d_ent):
        dq, dk, dent = ctx.saved_tensors
        qd, kd, ed, eshape = ctx.meta
        g = d_nce.to(torch.float32)
        return ((dq * g).to(qd), (dk * g).to(kd), (dent * d_ent.to(torch.float32)).reshape(eshape).to(ed),
                None, None, None, None, None)


def contrastive_entropy_loss(za: torch.Tensor, zb: torch.Tensor, masking: CurriculumMasking, entropy: torch.Tensor,
                             temperature: float = 0.07, entropy_weight: float = 0.01,
                             contrastive_weight: float = 1.0) -> torch.Tensor:
    """``contrastive_weight * info_nce(za, zb) + entropy_weight * masking.entropy_loss(entropy)`` (single rank, bf16
    embeddings) with the entropy regulariser riding in the launch of the first InfoNCE direction (aecf_loss_fwd_bwd)."""
    _require_device(za, "za")
    if za.shape != zb.shape or za.dim() != 2:
        raise ValueError(f"expected two [b, d] tensors of equal shape, got {tuple(za.shape)} and {tuple(zb.shape)}")
    na, nb = l2_normalize(za), l2_normalize(zb)
    coef = 0.5 / float(za.shape[0])
    seq_len = masking._last_seq_len if hasattr(masking, "_last_seq_len") else 2
    if _sym_supported(za, temperature, za.shape[0]):
        l_nce, l_ent = _NceSymmetric.apply(na, nb, entropy, 0, float(temperature), coef, None, int(seq_len),
                                           float(masking.entropy_target))
        return _plus(contrastive_weight * l_nce, entropy_weight * l_ent.to(za.dtype))
    l_ab, l_ent = _LossDirection.apply(na, nb, entropy, 0, float(temperature), coef, int(seq_len),
                                       float(masking.entropy_target))
    l_ba = _NceDirection.apply(nb, na, 0, float(temperature), coef, _stream_form(nb, na.shape[0]))
    return _plus(contrastive_weight * (l_ab + l_ba), entropy_weight * l_ent.to(za.dtype))


def gathered_contrastive_entropy_loss(za: torch.Tensor, nb_all: torch.Tensor, row_offset: int, masking: CurriculumMasking,
                                      entropy: torch.Tensor, temperature: float = 0.07, entropy_weight: float = 0.01,
                                      contrastive_weight: float = 1.0, group=None) -> torch.Tensor:
    """The loss side of a data-parallel step in ONE operator: this rank's rows ``za`` [b_local, d] (bf16, not yet normalised)
    against the unit-norm rows of the other view from EVERY rank ``nb_all`` [b_all, d] (``dp.all_gather_rows(l2_normalize(zb))``:
    its backward reduce-scatters the share of the gradient this call returns), positives at ``row_offset + i``; both InfoNCE
    directions from the one block of logits (``aecf_nce_sym_pass1`` / ``_loss`` / ``_grads``; the column sums are all-reduced over ``group`` between
    the passes) plus ``entropy_weight * masking.entropy_loss(entropy)`` riding in the same call.  Returns this rank's share of
    ``contrastive_weight * L_nce`` (coef = 0.5 / b_all) plus the entropy term."""
    _require_device(za, "za")
    if not _sym_supported(za, temperature):
        raise NotImplementedError("aecf_amd: the gathered contrastive loss needs bfloat16 rows with d % 64 == 0 and temperature >= 0.025")
    na = l2_normalize(za)
    coef = 0.5 / float(nb_all.shape[0])
    seq_len = masking._last_seq_len if hasattr(masking, "_last_seq_len") else 2
    l_nce, l_ent = _NceSymmetric.apply(na, nb_all, entropy, int(row_offset), float(temperature), coef, group, int(seq_len),
                                       float(masking.entropy_target))
    return _plus(contrastive_weight * l_nce, entropy_weight * l_ent.to(za.dtype))


def l2_normalize(z: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    _require_device(z, "z")
    return _L2Norm.apply(z, float(eps))


def info_nce(za: torch.Tensor, zb: torch.Tensor, temperature: float = 0.07, group=None) -> torch.Tensor:
    """Symmetric InfoNCE between the local rows of two views with negatives from every rank of ``group``.
    ``za``, ``zb``: [b_local, d] on a ROCm device.  bfloat16 with d % 64 == 0 (temperature >= 0.025) runs the symmetric tile-GEMM
    form: both directions from ONE block of logits, exponentials kept as bf16 [b_local, b_all] (any row counts); float32:
    d % 64 == 0 and total rows over ranks % 64 == 0."""
    _require_device(za, "za")
    _require_device(zb, "zb")
    if za.shape != zb.shape or za.dim() != 2:
        raise ValueError(f"info_nce expects two [b, d] tensors of equal shape, got {tuple(za.shape)} and {tuple(zb.shape)}")
    if za.dtype not in _DTYPES:
        raise NotImplementedError(f"aecf_amd: dtype {za.dtype} is not supported (bfloat16 / float32 only)")
    rank, world = dp.world_info(group)
    na, nb = l2_normalize(za), l2_normalize(zb)
    nb_all = dp.all_gather_rows(nb, group) if world > 1 else nb
    b_all = nb_all.shape[0]
    sym = _sym_supported(za, temperature, b_all)
    if world > 1:                                   # every rank must take the same form (they exchange different things)
        flag = torch.tensor([1 if sym else 0], device=za.device)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=group)
        sym = bool(int(flag.item()))
    if world > 1:
        sizes = torch.tensor([za.shape[0]], device=za.device)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        torch.distributed.all_gather(all_sizes, sizes, group=group)
        offset = int(sum(int(s.item()) for s in all_sizes[:rank]))
    else:
        offset = 0
    coef = 0.5 / float(b_all)
    if sym:
        # both directions from the one block of logits this rank owns (its rows of view a against every row of view b):
        # view a is never gathered, one all-reduce of b_all floats replaces the second direction's pass
        share, _ = _NceSymmetric.apply(na, nb_all, None, offset, float(temperature), coef, group, 2, 0.0)
    else:
        na_all = dp.all_gather_rows(na, group) if world > 1 else na
        # (the symmetric form was refused -- memory, temperature or dtype: the streaming kernels where they exist, never
        #  a second rows x cols allocation per direction)
        low = _stream_form(na, b_all)
        l_ab = _NceDirection.apply(na, nb_all, offset, float(temperature), coef, low)
        l_ba = _NceDirection.apply(nb, na_all, offset, float(temperature), coef, low)
        share = l_ab + l_ba              # this rank's rows' share of the global objective
    if world == 1:
        return share
    # Data-parallel convention (dp.FlatGradBucket.all_reduce(average=True)): gradients are AVERAGED over ranks, so
    # the local term carries a factor `world`; the returned VALUE is the global loss on every rank.
    total = share.detach().clone()
    torch.distributed.all_reduce(total, group=group)
    scaled = share * world
    return scaled + (total - scaled.detach())


def fusion_objective(task_loss: torch.Tensor, masking: Optional[CurriculumMasking], entropy: Optional[torch.Tensor],
                     za: Optional[torch.Tensor] = None, zb: Optional[torch.Tensor] = None, entropy_weight: float = 0.01,
                     contrastive_weight: float = 1.0, temperature: float = 0.07, group=None) -> torch.Tensor:
    """task + entropy_weight * entropy_loss(entropy) [ref README.md:205-208] + contrastive_weight * info_nce(za, zb)."""
    total = task_loss
    if masking is not None and entropy is not None:
        total = _plus(total, entropy_weight * masking.entropy_loss(entropy))
    if za is not None and zb is not None:
        total = _plus(total, contrastive_weight * info_nce(za, zb, temperature, group))
    return total
