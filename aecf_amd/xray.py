"""The fusion path's only end-to-end caller in the reference, rebuilt around device-side routing
(behaviour of ``AECFModel`` in ``xrays/train_xrays_example.py:108-237`` and of one
optimisation step of ``train_both_models``, ``:360-377``).

What is kept from the reference: the module/attribute names and construction order (so a reference state_dict loads and a
seeded construction draws the same parameters) and the forward contract ``model(images, texts, return_info) -> logits
[, info]``.  What is different is how rows travel.  The reference masks, ``torch.where``-s, stacks and index-assigns
three row subsets (six host synchronisations per step).  Here one routing table is built on the device from the presence
bits (``aecf_route_build``), its four class sizes are the single value read back, and rows move through two kernels:
``aecf_rows_gather`` (compact the rows a branch needs) and ``aecf_rows_select`` (write every fused row exactly once from the
branch that owns it, zeros for rows with no modality).  Each is the other's backward.
"""
from __future__ import annotations

import copy
import ctypes
import os
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .layer import CurriculumMasking, MultimodalAttentionPool, _DTYPES, _ptr, _stream
from . import _lib, dp

BOTH, ONLY_A, ONLY_B, NONE = 0, 1, 2, 3


def _need_device(t: torch.Tensor) -> None:
    if t.device.type != "cuda":
        raise RuntimeError("aecf_amd: the HIP path needs tensors on a ROCm device (no CPU fallback is provided)")


def modality_frontend(features: torch.Tensor, drop: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Missing-modality front-end of one modality in ONE kernel pass: rows with
    ``drop[r]`` set are zeroed (ref ``xrays/train_xrays_example.py:173-176``) and ``present[r] = ||row|| > 1e-6`` of the
    row as written (ref ``:202-203``).  Returns ``(features_out, present_u8)``; without ``drop`` the input tensor itself
    is returned.  Features are data (precomputed CLIP embeddings in the reference): no gradient flows to them."""
    _need_device(features)
    if features.dim() != 2 or features.dtype not in _DTYPES:
        raise ValueError("modality_frontend expects a [rows, dim] float32 or bfloat16 tensor")
    lib = _lib.load()
    f = features.detach().contiguous()
    rows, dim = f.shape
    present = torch.empty(rows, dtype=torch.uint8, device=f.device)
    d8 = None if drop is None else drop.to(device=f.device, dtype=torch.uint8).contiguous()
    out = f if drop is None else torch.empty_like(f)
    _lib.check(lib.aecf_modality_frontend(rows, dim, _DTYPES[f.dtype], _ptr(f), _ptr(d8),
                                          None if drop is None else _ptr(out), _ptr(present), _stream()),
               "aecf_modality_frontend")
    return out, present


def front_pair(image: torch.Tensor, text: torch.Tensor, uniforms: Optional[torch.Tensor] = None, missing_prob: float = 0.3,
               drop: Optional[Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]] = None):
    """Both front-ends, the missing-modality decisions and the row classes in ONE launch (``aecf_front_pair``; static routing).
    ``uniforms`` [3, rows] float32 = the draws of ``AECFModel.draw_missing`` (ref xrays/train_xrays_example.py:156-172), or
    ``drop`` = (drop_a, drop_b) decided by the caller, or neither.  Returns ``(image_out, text_out, present_a, present_b, cls)``:
    rows of an absent modality (dropped, zero norm, NaN) come out as zeros; ``cls`` in {BOTH, ONLY_A, ONLY_B, NONE}."""
    _need_device(image)
    if image.dim() != 2 or text.dim() != 2 or image.dtype not in _DTYPES or text.dtype != image.dtype \
            or text.shape[0] != image.shape[0]:
        raise ValueError("front_pair expects two [rows, dim] tensors of one dtype (float32 or bfloat16)")
    lib = _lib.load()
    a, b = image.detach().contiguous(), text.detach().contiguous()
    rows, dev = a.shape[0], a.device
    u = None
    if uniforms is not None:
        u = uniforms.detach().to(device=dev, dtype=torch.float32).contiguous()
        if u.shape != (3, rows):
            raise ValueError("front_pair: uniforms must be [3, rows]")
    da = db = None
    if drop is not None:
        if u is not None:
            raise ValueError("front_pair: uniforms and drop are alternatives")
        da = None if drop[0] is None else drop[0].to(device=dev, dtype=torch.uint8).contiguous()
        db = None if drop[1] is None else drop[1].to(device=dev, dtype=torch.uint8).contiguous()
    out_a, out_b = torch.empty_like(a), torch.empty_like(b)
    marks = torch.empty(2, rows, dtype=torch.uint8, device=dev)
    cls = torch.empty(rows, dtype=torch.int32, device=dev)
    _lib.check(lib.aecf_front_pair(rows, a.shape[1], b.shape[1], _DTYPES[a.dtype], _ptr(a), _ptr(b), _ptr(u), float(missing_prob),
                                   _ptr(da), _ptr(db), _ptr(out_a), _ptr(out_b), marks[0].data_ptr(), marks[1].data_ptr(),
                                   _ptr(cls), _stream()), "aecf_front_pair")
    return out_a, out_b, marks[0], marks[1], cls


class Route:
    """Routing table of one batch: ``cls[r]`` in {BOTH, ONLY_A, ONLY_B, NONE}, ``slot[r]`` = position of row r inside its
    class, ``index[c]`` = the rows of class c in ascending order, ``counts`` = the four class sizes (host integers: the one
    device->host copy of a step)."""

    def __init__(self, present_a: torch.Tensor, present_b: torch.Tensor):
        _need_device(present_a)
        lib = _lib.load()
        rows = present_a.numel()
        dev = present_a.device
        pa = present_a.to(torch.uint8).contiguous()
        pb = present_b.to(device=dev, dtype=torch.uint8).contiguous()
        self.rows = rows
        self.cls = torch.empty(rows, dtype=torch.int32, device=dev)
        self.slot = torch.empty(rows, dtype=torch.int32, device=dev)
        self._index = torch.empty(3, rows, dtype=torch.int32, device=dev)
        counts = torch.empty(4, dtype=torch.int32, device=dev)
        _lib.check(lib.aecf_route_build(rows, _ptr(pa), _ptr(pb), _ptr(self.cls), _ptr(self.slot), _ptr(self._index),
                                        _ptr(counts), _stream()), "aecf_route_build")
        self.counts = tuple(int(v) for v in counts.tolist())

    def index(self, c: int) -> torch.Tensor:
        return self._index[c, :self.counts[c]]


def _arr(ctype, values):
    return (ctype * len(values))(*values)


def _rows_gather(jobs: Sequence[Tuple[torch.Tensor, int, torch.Tensor, int, int, int]], row_bytes: int) -> None:
    """jobs: (src tensor, src pitch bytes, index, dst address, dst pitch bytes, n)."""
    jobs = [j for j in jobs if j[5] > 0]
    if not jobs:
        return
    lib = _lib.load()
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    _lib.check(lib.aecf_rows_gather(
        len(jobs), _arr(vp, [j[0].data_ptr() for j in jobs]), _arr(i64, [j[1] for j in jobs]),
        _arr(vp, [j[2].data_ptr() for j in jobs]), _arr(i64, [j[5] for j in jobs]), _arr(vp, [j[3] for j in jobs]),
        _arr(i64, [j[4] for j in jobs]), row_bytes, _stream()), "aecf_rows_gather")


def _rows_select(route: Route, srcs: Sequence[Optional[Tuple[int, int]]], dst: torch.Tensor, row_bytes: int) -> None:
    """srcs[c] = (address, pitch bytes) of class c's compact rows or None; dst [rows, *] contiguous, fully written."""
    lib = _lib.load()
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    ptrs = [None if (s is None or route.counts[c] == 0) else s[0] for c, s in enumerate(srcs)]
    pitches = [0 if s is None else s[1] for s in srcs]
    _lib.check(lib.aecf_rows_select(route.rows, row_bytes, _ptr(route.cls), _ptr(route.slot), _arr(vp, ptrs),
                                    _arr(i64, pitches), _ptr(dst), dst.stride(0) * dst.element_size(), _stream()),
               "aecf_rows_select")


class _PairGather(torch.autograd.Function):
    """[n_both, 2, E] pool input from the two encoder outputs (what ref :213-214 builds with where + stack)."""

    @staticmethod
    def forward(ctx, a, b, route):
        a, b = a.contiguous(), b.contiguous()
        n, (rows, E), es = route.counts[BOTH], a.shape, a.element_size()
        out = torch.empty(n, 2, E, dtype=a.dtype, device=a.device)
        idx = route.index(BOTH)
        _rows_gather([(a, E * es, idx, out.data_ptr(), 2 * E * es, n),
                      (b, E * es, idx, out.data_ptr() + E * es, 2 * E * es, n)], E * es)
        ctx.route = route
        return out

    @staticmethod
    def backward(ctx, d_out):
        route = ctx.route
        d_out = d_out.contiguous()
        n, _, E = d_out.shape
        es = d_out.element_size()
        da = torch.empty(route.rows, E, dtype=d_out.dtype, device=d_out.device)
        db = torch.empty_like(da)
        _rows_select(route, [(d_out.data_ptr(), 2 * E * es), None, None], da, E * es)
        _rows_select(route, [(d_out.data_ptr() + E * es, 2 * E * es), None, None], db, E * es)
        return da, db, None


class _ClassGather(torch.autograd.Function):
    """Compact rows of one class (the input of a single-modality projection, ref :229-234)."""

    @staticmethod
    def forward(ctx, src, route, cls):
        src = src.contiguous()
        n, W, es = route.counts[cls], src.shape[1], src.element_size()
        out = torch.empty(n, W, dtype=src.dtype, device=src.device)
        _rows_gather([(src, W * es, route.index(cls), out.data_ptr(), W * es, n)], W * es)
        ctx.route, ctx.cls = route, cls
        return out

    @staticmethod
    def backward(ctx, d_out):
        route, cls = ctx.route, ctx.cls
        d_out = d_out.contiguous()
        W, es = d_out.shape[1], d_out.element_size()
        d_src = torch.empty(route.rows, W, dtype=d_out.dtype, device=d_out.device)
        srcs = [None, None, None]
        srcs[cls] = (d_out.data_ptr(), W * es)
        _rows_select(route, srcs, d_src, W * es)
        return d_src, None, None


class _BranchSelect(torch.autograd.Function):
    """fused[r] = the row its branch produced (both -> fusion_proj, only-a -> image_proj, only-b -> text_proj, none -> 0):
    the three index-assignments of ref :209-234 as one pass that writes every row once."""

    @staticmethod
    def forward(ctx, both, only_a, only_b, route, width):
        parts = [p.contiguous() for p in (both, only_a, only_b)]
        dt, dev = parts[0].dtype, parts[0].device
        es = parts[0].element_size()
        out = torch.empty(route.rows, width, dtype=dt, device=dev)
        _rows_select(route, [(p.data_ptr(), width * es) for p in parts], out, width * es)
        ctx.route = route
        return out

    @staticmethod
    def backward(ctx, d_out):
        route = ctx.route
        d_out = d_out.contiguous()
        W, es = d_out.shape[1], d_out.element_size()
        grads = [torch.empty(route.counts[c], W, dtype=d_out.dtype, device=d_out.device) for c in range(3)]
        _rows_gather([(d_out, W * es, route.index(c), grads[c].data_ptr(), W * es, route.counts[c]) for c in range(3)],
                     W * es)
        return grads[0], grads[1], grads[2], None, None


class _StaticSelect(torch.autograd.Function):
    """Static routing: every branch produced a row for EVERY sample; fused[r] = the row of the branch that owns r (zeros for a
    row with no modality) in one pass, and the gradient of fused[r] goes to that branch alone (zeros to the others) in one
    more -- what three ``torch.where`` + two adds and their backward nodes do in the tensor formulation."""

    @staticmethod
    def forward(ctx, both, only_a, only_b, cls):
        parts = [p.contiguous() for p in (both, only_a, only_b)]
        rows, width = parts[0].shape
        es = parts[0].element_size()
        out = torch.empty_like(parts[0])
        lib = _lib.load()
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        _lib.check(lib.aecf_rows_select(rows, width * es, _ptr(cls), None, _arr(vp, [p.data_ptr() for p in parts]),
                                        _arr(i64, [width * es] * 3), _ptr(out), width * es, _stream()), "aecf_rows_select")
        ctx.save_for_backward(cls)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (cls,) = ctx.saved_tensors
        d_out = d_out.contiguous()
        rows, width = d_out.shape
        grads = [torch.empty_like(d_out) for _ in range(3)]
        lib = _lib.load()
        _lib.check(lib.aecf_rows_split(rows, width * d_out.element_size(), _ptr(cls), _ptr(d_out),
                                       _arr(ctypes.c_void_p, [g.data_ptr() for g in grads]), _stream()), "aecf_rows_split")
        return grads[0], grads[1], grads[2], None


class AECFModel(nn.Module):
    """Image + text multi-label classifier around the fusion pool (behaviour of ref xrays/train_xrays_example.py:108-237;
    module names and construction order as there, so seeds and checkpoints carry over)."""

    def __init__(self, image_dim: int = 512, text_dim: int = 512, num_classes: int = 80, hidden_dim: int = 256):
        super().__init__()
        self.name = "AECF_Model"
        self.hidden_dim = hidden_dim
        self.curriculum_enabled = False
        self.missing_modality_training = False
        # static_routing: every branch processes EVERY row and a per-row select keeps the owner's result -- no class sizes are
        # read back, every shape is fixed by the batch size, so a whole training step can be captured as one HIP graph
        # (GraphedTrainStep below).  Same logits, loss and parameter gradients as the compact routing (rows outside a branch
        # get a zero upstream gradient); info tensors then cover all rows (info["both"] marks the rows that have both).
        self.static_routing = False
        enc = lambda d: nn.Sequential(nn.Linear(d, hidden_dim), nn.ReLU(), nn.Dropout(0.1))
        self.image_encoder = enc(image_dim)
        self.text_encoder = enc(text_dim)
        self.curriculum_masking = CurriculumMasking(base_mask_prob=0.15)
        self.attention_pool = MultimodalAttentionPool(embed_dim=hidden_dim, num_heads=4, curriculum_masking=None,
                                                      batch_first=True)
        self.fusion_query = nn.Parameter(torch.randn(1, 1, hidden_dim) * 0.02)
        self.image_proj = nn.Linear(hidden_dim, hidden_dim * 2)
        self.text_proj = nn.Linear(hidden_dim, hidden_dim * 2)
        self.fusion_proj = nn.Linear(hidden_dim, hidden_dim * 2)
        self.classifier = nn.Sequential(nn.Linear(hidden_dim * 2, hidden_dim), nn.ReLU(), nn.Dropout(0.1),
                                        nn.Linear(hidden_dim, num_classes))

    def toggle_curriculum(self, enabled: bool) -> None:
        self.curriculum_enabled = enabled
        self.attention_pool.curriculum_masking = self.curriculum_masking if enabled else None

    def draw_missing(self, batch_size: int, device, missing_prob: float = 0.3, generator=None):
        """Which rows lose which modality in this step (ref :156-172): each modality independently with probability
        ``missing_prob``; a row that would lose both keeps one of them, chosen by a fair coin.  Branch-free on the device:
        three uniform vectors are always drawn (the reference draws the third only for the rows that need it, which costs it a
        host synchronisation; the distribution is the same, the default generator advances by 3B instead of 2B + k)."""
        u = torch.rand(3, batch_size, device=device, generator=generator)
        drop_a, drop_b = u[0] < missing_prob, u[1] < missing_prob
        clash, keep_a = drop_a & drop_b, u[2] > 0.5
        return drop_a & ~(clash & keep_a), drop_b & ~(clash & ~keep_a)

    def forward(self, image_features: torch.Tensor, text_features: torch.Tensor, return_info: bool = False, *,
                mask_uniforms: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                missing: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """``mask_uniforms`` / ``generator`` are handed to the pool's curriculum masking (``[n_both, 1, 2]``).  ``missing`` =
        (drop_a, drop_b) for this call's rows, drawn by the caller: a data-parallel trainer draws them for the GLOBAL batch from
        a shared-seed generator (``draw_missing``) and hands every rank its rows, so that which rows lose a modality does not
        depend on the number of ranks (as for the mask uniforms); without it they are drawn here."""
        drop_a = drop_b = None
        simulate = self.training and self.missing_modality_training
        if self.static_routing:
            # one launch for both front-ends, the decisions and the row classes; the draws are draw_missing's three uniform
            # vectors (one torch.rand node: its offset advances under graph replay)
            u = None
            if simulate and missing is None:
                u = torch.rand(3, image_features.size(0), device=image_features.device, generator=generator)
            xa, xb, _, _, cls = front_pair(image_features, text_features, u, 0.3, missing if simulate else None)
            return self._forward_static(xa, xb, cls, return_info, mask_uniforms, generator)
        if simulate:
            if missing is not None:
                drop_a, drop_b = missing
            else:
                drop_a, drop_b = self.draw_missing(image_features.size(0), image_features.device, generator=generator)
        image_features, has_a = modality_frontend(image_features, drop_a)
        text_features, has_b = modality_frontend(text_features, drop_b)
        route = Route(has_a, has_b)
        enc_a = self.image_encoder(image_features)
        enc_b = self.text_encoder(text_features)
        width = 2 * self.hidden_dim

        info: Dict[str, torch.Tensor] = {}
        if route.counts[BOTH] > 0:
            if mask_uniforms is not None and mask_uniforms.shape[0] == route.rows != route.counts[BOTH]:
                # uniforms given per BATCH row (a data-parallel step hands every rank its rows of one global tensor):
                # compact them like the features, so that a row's mask does not depend on how the batch was sharded
                mu = mask_uniforms.detach().to(device=enc_a.device, dtype=torch.float32).reshape(route.rows, -1).contiguous()
                cu = torch.empty(route.counts[BOTH], mu.shape[1], dtype=torch.float32, device=mu.device)
                _rows_gather([(mu, mu.shape[1] * 4, route.index(BOTH), cu.data_ptr(), mu.shape[1] * 4, route.counts[BOTH])],
                             mu.shape[1] * 4)
                mask_uniforms = cu
            pairs = _PairGather.apply(enc_a, enc_b, route)
            pooled, pool_info = self.attention_pool(self.fusion_query.expand(route.counts[BOTH], -1, -1), pairs, pairs,
                                                    return_info=True, uniforms=mask_uniforms, generator=generator)
            from_both = self.fusion_proj(pooled.squeeze(1))
            if return_info:
                info.update(pool_info)
        else:
            from_both = enc_a.new_zeros(0, width)
        from_a = self.image_proj(_ClassGather.apply(enc_a, route, ONLY_A)) if route.counts[ONLY_A] else enc_a.new_zeros(0, width)
        from_b = self.text_proj(_ClassGather.apply(enc_b, route, ONLY_B)) if route.counts[ONLY_B] else enc_a.new_zeros(0, width)
        fused = _BranchSelect.apply(from_both, from_a, from_b, route, width)
        logits = self.classifier(fused)
        return (logits, info) if return_info else logits


def _forward_static(self, image_features, text_features, cls, return_info, mask_uniforms, generator):
    """Every row runs every branch; ``image_features`` / ``text_features`` come from ``front_pair``: an ABSENT modality's rows
    are zeros there, so its features never reach a weight (compact routing never touches them; presence = norm > 1e-6, ref
    :202-203, is False for NaN / Inf rows, and 0 * NaN = NaN would poison the batch sums of every parameter gradient)."""
    rows = image_features.size(0)
    enc_a = self.image_encoder(image_features)
    enc_b = self.text_encoder(text_features)
    pairs = torch.stack([enc_a, enc_b], dim=1)                                   # [rows, 2, E] (what ref :213-214 stacks)
    pooled, pool_info = self.attention_pool(self.fusion_query.expand(rows, -1, -1), pairs, pairs, return_info=True,
                                            uniforms=mask_uniforms, generator=generator)
    fused = _StaticSelect.apply(self.fusion_proj(pooled.squeeze(1)), self.image_proj(enc_a), self.text_proj(enc_b), cls)
    logits = self.classifier(fused)
    if not return_info:
        return logits
    info = dict(pool_info)
    info["both"] = cls == BOTH
    return logits, info


AECFModel._forward_static = _forward_static


class GraphedTrainStep:
    """One optimisation step of ``AECFModel`` (zero_grad, forward, BCE, backward, AdamW: ref xrays/train_xrays_example.py:360-377)
    captured ONCE as a HIP graph and replayed per batch.  The eager step is ~50 launches of a few microseconds of device work
    each plus one device->host read: at the reference's batch of 64 the host's launch rate is the whole step time (1.9 ms,
    profiles/r03_c4_b64_bench.json).  Static routing (no class sizes read back) makes every shape a function of the batch
    size alone; the batch, labels and the returned loss live in fixed buffers that a replay reads and writes.

    Randomness (curriculum mask, missing-modality draws) comes from the device's default generator, whose offset a replay
    advances; the optimizer must be constructed ``capturable=True``.  One rank only: the captured step has no collective
    (data-parallel callers capture per rank and all-reduce between backward and step themselves)."""

    def __init__(self, model: AECFModel, optimizer: torch.optim.Optimizer, criterion: nn.Module, batch: int, image_dim: int,
                 text_dim: int, num_classes: int, device, dtype=torch.float32, warmup: int = 3, tune_gemm: bool = False):
        """Side effects, all deliberate: ``model.static_routing`` is set to True and stays so (the captured shapes depend on
        it); the device generator advances by the warm-up and capture draws.  NOT a side effect: the warm-up and capture
        steps run real optimisation steps on noise, so the model's parameters and buffers and the optimizer's state
        (moments, step counters) are snapshotted before them and restored afterwards -- the first replay starts from exactly
        the state the caller handed over.

        ``tune_gemm``: the nn.Linear layers around the pool run on torch's BLAS dispatch, whose default pick for a batch of 64
        is a 256 x 64 macro-tile kernel of 26-33 us per GEMM (profiles/r04_c4_notes.md); with this flag torch's TunableOp times
        the rocBLAS / hipBLASLt candidates for each shape during the warm-up steps; tuning is off during the capture (nothing is
        timed under capture), the picks are baked into the captured graph, and the process-wide switches are handed back as they
        were found afterwards."""
        model.static_routing = True
        if tune_gemm:
            import torch.cuda.tunable as tunable
            prev_tunable = (tunable.is_enabled(), tunable.tuning_is_enabled())
            tunable.enable(True)
            tunable.tuning_enable(True)
            if hasattr(tunable, "write_file_on_exit"):
                tunable.write_file_on_exit(False)
            else:                                             # (this torch writes its picks at exit: into a file of this process's own)
                import tempfile
                fd, name = tempfile.mkstemp(prefix="aecf_tunableop_", suffix=".csv")
                os.close(fd)
                tunable.set_filename(name)
        self.model, self.optimizer, self.criterion = model, optimizer, criterion
        self.image = torch.zeros(batch, image_dim, device=device, dtype=dtype)
        self.text = torch.zeros(batch, text_dim, device=device, dtype=dtype)
        self.labels = torch.zeros(batch, num_classes, device=device, dtype=dtype)
        self.image.normal_()
        self.text.normal_()
        model_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        had_state = len(optimizer.state) > 0
        opt_state = copy.deepcopy(optimizer.state_dict()) if had_state else None
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                        # warm-up on a side stream (allocator, lazily built state)
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        if tune_gemm:
            torch.cuda.synchronize(device)
            tunable.tuning_enable(False)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._step()
        if tune_gemm:                                        # the picks are baked into the graph: hand the process-wide switches back
            tunable.tuning_enable(prev_tunable[1])
            tunable.enable(prev_tunable[0])
        # restore IN PLACE (the graph holds the addresses of the parameters and of the optimizer's state tensors)
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(model_state[k])
            if had_state:
                saved = opt_state["state"]
                live = optimizer.state_dict()["state"]
                for idx, st in live.items():
                    for name, val in st.items():
                        if torch.is_tensor(val):
                            if idx in saved and name in saved[idx]:
                                val.copy_(saved[idx][name])
                            else:                            # state the warm-up steps created (the caller had none for it)
                                val.zero_()
            else:                                            # fresh optimizer: moments and step counters back to zero
                for st in optimizer.state.values():
                    for val in st.values():
                        if torch.is_tensor(val):
                            val.zero_()

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)
        logits = self.model(self.image, self.text)
        loss = self.criterion(logits, self.labels)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, images: torch.Tensor, texts: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        self.image.copy_(images, non_blocking=True)
        self.text.copy_(texts, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        self.graph.replay()
        return self.loss


def train_step(model: AECFModel, optimizer: torch.optim.Optimizer, criterion: nn.Module, images: torch.Tensor,
               texts: torch.Tensor, labels: torch.Tensor, bucket: Optional[dp.FlatGradBucket] = None,
               loss_scale: float = 1.0) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """One optimisation step (ref :360-377: zero_grad, forward with info, BCE, backward, step).  Data parallel: pass the
    model's FlatGradBucket -- all gradients of all ranks are then averaged by one all-reduce before the optimizer step;
    ``loss_scale`` = ``world * b_local / B_global`` makes that average the gradient of the global-batch mean loss when
    the shards are uneven (1.0 for even shards)."""
    if bucket is None:
        optimizer.zero_grad(set_to_none=True)
    else:
        bucket.zero()
    logits, info = model(images, texts, return_info=True)
    loss = criterion(logits, labels)
    (loss * loss_scale if loss_scale != 1.0 else loss).backward()
    if bucket is not None:
        bucket.all_reduce(average=True)
    optimizer.step()
    return loss.detach(), info
