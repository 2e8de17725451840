"""Host-side mirror of the reference's operator interface for the fusion path.

Same names, signatures, validation messages, ``info`` dictionary and state_dict keys as
``aecf/AECFLayer.py`` of leochlon/aecf (cited as ``ref:`` below); the arithmetic is done by
hand-written HIP kernels in ``libaecf_hip.so`` reached through ctypes (``_lib.py``).  PyTorch is
used for device memory, streams, autograd bookkeeping and parameter storage only.

There is no CPU path and no PyTorch fallback: tensors must live on a ROCm device and the library
must be present, otherwise a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import math
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import _lib

__all__ = ['CurriculumMasking', 'MultimodalAttentionPool', 'multimodal_attention_pool', 'create_fusion_pool']

_DTYPES = {torch.bfloat16: _lib.AECF_BF16, torch.float32: _lib.AECF_F32}


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """hipStream_t of the current device's current stream (the raw-handle call is ~20x cheaper than building a
    torch.cuda.Stream object, and this runs three times per training step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _draw_uniforms(shape, device, uniforms: Optional[torch.Tensor] = None,
                   generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The float32 uniforms of the Bernoulli draw (ref :204), one per weight element, row-major.

    ``uniforms`` given: used as they are (any shape with the right element count) -- this is how a data-parallel
    step feeds every rank ITS rows of one global tensor (``dp.global_uniforms`` + ``dp.shard_batch``) so that the
    N-rank masks equal the 1-rank masks bit for bit, and how a caller replays a recorded draw.  Otherwise they are
    drawn from ``generator`` (default: the device's default generator, which then advances by exactly
    ``numel`` draws as ``torch.bernoulli`` would)."""
    if uniforms is not None:
        n = 1
        for d in shape:
            n *= int(d)
        if uniforms.numel() != n:
            raise ValueError(f"uniforms has {uniforms.numel()} elements, the mask needs {n} (shape {tuple(shape)})")
        return uniforms.detach().to(device=device, dtype=torch.float32).reshape(shape).contiguous()
    return torch.rand(shape, dtype=torch.float32, device=device, generator=generator)


def _philox_draw(n: int, device: torch.device, generator: Optional[torch.Generator] = None, element0: int = 0):
    """The generator call ``torch.rand(n, device=device, generator=generator)`` WITHOUT making it: returns
    ``(seed, offset, threads, element0)`` -- the generator's state before the call and the thread count of torch's launch
    (ATen/native/cuda/DistributionTemplates.h: blocks of 256, grid = min(ceil(n / 256), CUs * max_threads_per_CU / 256)) --
    and advances the generator by what that call consumes (4 per 4 * threads elements, one iteration here).  The statistics
    kernel evaluates the same Philox4x32-10 stream element by element (AECF_DRAW_UNIFORMS, include/aecf_hip.h), so the masks are
    those of the tensor path bit for bit (``tests/test_pool_gpu.py::test_in_kernel_uniforms_equal_torch_rand``).
    ``element0``: the caller's rows start at this element of the n-element draw (a data-parallel shard, ABI v9).
    Returns None where the offset cannot be read on the host: while the stream is being captured into a graph (a replay
    must see a fresh offset: torch's graph-safe generator state does that for ``torch.rand``), for a non-device generator, or
    when this torch build's ``torch.rand`` does not have the launch geometry written down here (checked once per device)."""
    if device.type != "cuda" or torch.cuda.is_current_stream_capturing():
        return None
    index = device.index if device.index is not None else torch.cuda.current_device()
    gen = generator if generator is not None else torch.cuda.default_generators[index]
    if gen.device.type != "cuda":
        return None
    if not _draw_geometry_ok(index):
        return None
    seed, offset, threads, increment = _philox_geometry(n, index, gen)
    gen.set_offset(offset + increment)
    return seed, offset, threads, int(element0)


def _philox_geometry(n: int, index: int, gen: torch.Generator):
    props = _device_props(index)
    blocks = min((n + 255) // 256, props[0] * (props[1] // 256))
    threads = 256 * blocks
    increment = ((n - 1) // (threads * 4) + 1) * 4
    return int(gen.initial_seed()) & 0xFFFFFFFFFFFFFFFF, int(gen.get_offset()), int(threads), int(increment)


_geometry_ok: Dict[int, bool] = {}


def _draw_geometry_ok(index: int) -> bool:
    """ATen's launch geometry of ``torch.rand`` (block 256, unroll 4, the grid cap, the offset increment) is private to torch;
    the in-kernel draw is only used where it reproduces THIS torch build's tensor: once per device, a draw below and one beyond
    the grid cap are made both ways from a scratch generator and compared (values and generator advance).  A mismatch falls
    back to the tensor path for the life of the process (one ``torch.rand`` launch more per step; the masks stay valid)."""
    ok = _geometry_ok.get(index)
    if ok is None:
        ok = True
        try:
            lib = _lib.load()
            dev = torch.device("cuda", index)
            props = _device_props(index)
            for n in (1000, 256 * props[0] * (props[1] // 256) * 4 + 333):
                gen = torch.Generator(device=dev).manual_seed(0x5EED + n)
                torch.rand(8, device=dev, generator=gen)                      # a non-zero offset to start from
                seed, offset, threads, increment = _philox_geometry(n, index, gen)
                want = torch.rand(n, device=dev, generator=gen)
                got = torch.empty(n, dtype=torch.float32, device=dev)
                _lib.check(lib.aecf_philox_uniforms(n, seed, offset, threads, 0, _ptr(got), _stream()), "aecf_philox_uniforms")
                if gen.get_offset() != offset + increment or not torch.equal(got, want):
                    ok = False
                    break
        except Exception:                                                     # noqa: BLE001 -- any surprise: tensor path
            ok = False
        _geometry_ok[index] = ok
    return ok


_props_cache: Dict[int, Tuple[int, int]] = {}


def _device_props(index: int) -> Tuple[int, int]:
    p = _props_cache.get(index)
    if p is None:
        dp_ = torch.cuda.get_device_properties(index)
        p = (int(dp_.multi_processor_count), int(dp_.max_threads_per_multi_processor))
        _props_cache[index] = p
    return p


class DpState:
    """Data-parallel state of ONE pool module (``aecf_amd.dp.attach`` makes it; nothing here is process-global).

    ``grad_scale``: factor the backward folds into the five parameter gradients as it stores them (1 / world: the gradient
    average is then ONE sum all-reduce, no divide launch).  ``scaled``: the leaf tensors whose gradients carry that factor -- the
    module's parameters and every leaf fusion query its forward has seen (weak references) -- which is how
    ``dp.all_reduce_grads`` knows not to divide them again, whatever autograd did with the tensors in between (two pool
    applications in one backward are summed into fresh allocations).
    ``keep_f32``: bf16 parameters -- the backward writes its float32 batch sums, the collective averages THOSE and the bf16
    gradient is rounded once from the mean.  ``defer_rounding``: with ``keep_f32`` the bf16 tensors autograd receives stay
    UNINITIALISED until ``dp.all_reduce_grads`` / ``GradOverlap.finish`` writes the rounded mean into them (no cast launch;
    reading ``p.grad`` before the collective is an error).  ``hook``: a ``dp.GradOverlap`` while one is active.
    ``runs``: (flat, flat32, version) of the fused backward calls since the last collective -- where the float32 sums behind a
    gradient run are found.  The tensors are held (a few MB each, at most 8), so a pointer compared against them cannot have
    been freed and reused."""
    __slots__ = ("world", "grad_scale", "keep_f32", "defer_rounding", "hook", "runs", "scaled")

    def __init__(self, world: int = 1, grad_scale: float = 1.0, keep_f32: bool = True, defer_rounding: bool = False):
        self.world, self.grad_scale, self.keep_f32, self.defer_rounding = int(world), float(grad_scale), bool(keep_f32), bool(defer_rounding)
        self.hook = None
        self.runs = []
        self.scaled = {}                           # id(tensor) -> weak reference (identity, never tensor equality)

    def add_scaled(self, t: torch.Tensor) -> None:
        import weakref
        if len(self.scaled) > 64:
            self.scaled = {k: r for k, r in self.scaled.items() if r() is not None}
        self.scaled[id(t)] = weakref.ref(t)

    def is_scaled(self, t: torch.Tensor) -> bool:
        r = self.scaled.get(id(t))
        return r is not None and r() is t

    def record(self, flat, flat32) -> None:
        if len(self.runs) >= 8:
            del self.runs[1:-1]
        self.runs.append((flat, flat32, flat._version))

    def take(self, flat: Optional[torch.Tensor]):
        """The float32 sums behind the gradient run ``flat`` (as dp.flat_grad_alias returns it), if ``flat`` IS a run one of this
        module's backward calls wrote since the last collective and nothing has written to it since (an accumulation moves its
        version counter); else None.  Consumes the records."""
        runs, self.runs = self.runs, []
        if flat is None:
            return None
        for f, w, version in runs:
            if (w is not None and f._version == version and f.data_ptr() == flat.data_ptr() and f.numel() == flat.numel()
                    and f.dtype == flat.dtype):
                return w
        return None


class PoolOptions:
    """Per-module switches of the fused path (``MultimodalAttentionPool.options``); tests and bench.py set them on the module
    they drive, nothing is process-global.

    ``hilo_grads``: the weight-gradient products of the backward on bf16 hi + lo operand pairs (AECF_HILO_GRADS,
    include/aecf_hip.h) -- None (default) = whenever the parameter gradients are float32-STORED under bf16 activations (float32
    master weights: there the operand roundings of the default products are visible, 1.3e-3 .. 2.3e-3 of fp32 math at the
    headline shape; a bf16-stored gradient hides them behind its own rounding), True / False = forced.  Shapes the flag is not
    built for run the default products.
    ``draw_in_kernel``: the curriculum mask's uniforms are drawn by the statistics kernel (no torch.rand launch, no [B, M]
    tensor).  ``share_prep``: the forward's preparation launch also produces what the backward derives from the parameters
    alone.  ``stage_events``: (forward, backward) arrays of hipEvent_t the library records at its stage boundaries (bench.py).
    ``dp``: DpState of a data-parallel run."""
    __slots__ = ("hilo_grads", "draw_in_kernel", "share_prep", "stage_events", "dp")

    def __init__(self):
        self.hilo_grads: Optional[bool] = None
        self.draw_in_kernel = True
        self.share_prep = True
        self.stage_events = None
        self.dp: Optional[DpState] = None


_DEFAULT_OPTIONS = PoolOptions()           # direct users of _PoolFunction.apply (never modified)


def _require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"aecf_amd: {what} must be on a ROCm device (got {t.device}); this package has no CPU path")


# ----------------------------------------------------------------------------------------------
# autograd bridges (one per C-ABI operator)
# ----------------------------------------------------------------------------------------------
# what the library answers per call shape (status of aecf_pool_check, workspace sizes, whether the backward wants V):
# pure functions of the description, asked once per shape instead of on every call (each ctypes round trip is ~1-2 us of a
# host-bound step at small batches)
_shape_facts: Dict[Tuple, Tuple] = {}


def _pool_facts(lib, desc, key):
    f = _shape_facts.get(key)
    if f is None:
        ref = ctypes.byref(desc)
        status = lib.aecf_pool_check(ref)
        f = (status,) if status != 0 else (0, lib.aecf_pool_fwd_workspace_bytes(ref), lib.aecf_pool_bwd_workspace_bytes(ref),
                                           lib.aecf_pool_prep_bytes(ref), bool(lib.aecf_pool_wants_saved_v(ref)),
                                           lib.aecf_pool_hilo_bwd_workspace_bytes(ref))
        if len(_shape_facts) > 256:
            _shape_facts.clear()
        _shape_facts[key] = f
    return f


class _PoolFunction(torch.autograd.Function):
    """aecf_pool_forward / aecf_pool_backward.  Outputs: y [B,E], attn_w [B,M], masked_w [B,M], entropy [B],
    mask_rate [B], all in the activation dtype (the library writes the info tensors in that dtype itself; the
    float32 attn_w it also writes is kept for the backward)."""

    @staticmethod
    def forward(ctx, x, q, w_in, b_in, w_out, b_out, kpm, uniforms, num_heads, mask_mode, min_active,
                base_mask_prob, entropy_target, eps, f32_info=False, target_value=None, casts=None, side=None, philox=None,
                opts=None):
        lib = _lib.load()
        opts = _DEFAULT_OPTIONS if opts is None else opts
        ctx.set_materialize_grads(False)       # unused outputs arrive as None, not as zero tensors (fill + cast launches)
        B, M, E = x.shape
        dt = x.dtype
        desc = _lib.PoolDesc(B, M, E, num_heads, _DTYPES[dt], mask_mode, min_active, base_mask_prob,
                             entropy_target, eps)
        facts = _pool_facts(lib, desc, (B, M, E, num_heads, dt, mask_mode))
        _lib.check(facts[0], "aecf_pool_check")
        _, fwd_ws_bytes, bwd_ws_bytes, prep_bytes, wants_v, hilo_bytes = facts
        xc = x.contiguous()
        if casts is not None and len(casts) > 4 and casts[4] is not None:
            qc = casts[4].reshape(E)           # (cast by the module's one multi-tensor launch, with the weights)
        else:
            qc = q.detach().reshape(E).to(dt).contiguous()
        if casts is not None:                  # activation-dtype copies of master weights, cached by the module
            w_in_c, b_in_c, w_out_c, b_out_c = casts[:4]
        else:
            w_in_c = w_in.detach().to(dt).contiguous()
            w_out_c = w_out.detach().to(dt).contiguous()
            b_in_c = None if b_in is None else b_in.detach().to(dt).contiguous()
            b_out_c = None if b_out is None else b_out.detach().to(dt).contiguous()
        dev = x.device
        y = torch.empty(B, E, dtype=dt, device=dev)
        attn_w = torch.empty(B, M, dtype=torch.float32, device=dev)
        probs = torch.empty(B, num_heads, M, dtype=torch.float32, device=dev)
        saved_o = torch.empty(B, E, dtype=dt, device=dev)
        # per-modality value projections, kept only when a backward will follow (B*M*E elements)
        need_bwd = any(t is not None and t.requires_grad for t in (x, q, w_in, b_in, w_out, b_out))
        saved_v = torch.empty(B, M, E, dtype=dt, device=dev) if (need_bwd and wants_v) else None
        # what the backward derives from the parameters alone is produced by the forward's preparation launch
        saved_prep = torch.empty(prep_bytes, dtype=torch.uint8, device=dev) if (need_bwd and opts.share_prep) else None
        prep_ready = False
        if side is not None and "prep_cache" in side and opts.share_prep:
            # the module vouches that its parameters stand still (eval mode / no gradient recording, version counters and storage
            # unchanged): ONE preparation buffer serves every call until they move (AECF_PREP_READY skips the launch)
            cached, prep_ready = side["prep_cache"]
            if cached is not None and cached.numel() == prep_bytes and cached.device == dev:
                saved_prep = cached
            else:
                saved_prep = torch.empty(prep_bytes, dtype=torch.uint8, device=dev)
                prep_ready = False
            side["prep_cache"] = (saved_prep, True)
        if mask_mode != 0:
            masked_w = torch.empty(B, M, dtype=torch.float32, device=dev)
            entropy = torch.empty(B, dtype=torch.float32, device=dev)
            mask_rate = torch.empty(B, dtype=torch.float32, device=dev)
        else:
            masked_w = entropy = mask_rate = None
        if dt != torch.float32 and not f32_info:   # info copies in the activation dtype, written by the gate kernel
            i_attn_w = torch.empty(B, M, dtype=dt, device=dev)
            i_masked_w = None if masked_w is None else torch.empty(B, M, dtype=dt, device=dev)
            i_entropy = None if entropy is None else torch.empty(B, dtype=dt, device=dev)
            # training-mode mask_rate stays float32 (ref :275); eval mode returns it in the input dtype (ref :150-156)
            i_mask_rate = None if (mask_rate is None or mask_mode == 1) else torch.empty(B, dtype=dt, device=dev)
        else:
            i_attn_w = i_masked_w = i_entropy = i_mask_rate = None
        # info['target_entropy'] (ref :273), filled by the kernel that writes the other info tensors
        i_target = torch.empty(B, dtype=dt, device=dev) if (mask_mode == 1 and target_value is not None) else None
        # partial sums of the entropy regulariser over this call's rows, left behind by the kernel that writes the entropies
        # (training mode): CurriculumMasking.entropy_loss(info['entropy']) is then one small launch (side = a dict of the caller)
        # ... and, ABI v8, the loss itself: the out-projection launch's first block adds the partials up (no launch at all)
        # bf16 only: there the sums ride in kernels that run anyway; the float32 kernels would need two launches of their own for
        # them, which a caller that never asks for the regulariser (the example model's step) pays for nothing
        ent_partial = ent_loss = None
        if side is not None and mask_mode == 1 and target_value is not None and dt == torch.bfloat16:
            ent_partial = torch.empty((B + 255) // 256, dtype=torch.float32, device=dev)
            if not f32_info:
                ent_loss = torch.empty(1, dtype=dt, device=dev)
            side["ent_partial"] = (ent_partial, B, float(target_value), ent_loss)
        flags = 0
        saved_o_lo = None
        hilo_ws = 0
        hilo = opts.hilo_grads
        if hilo is None:                           # automatic: the parameter gradients will be STORED in float32 (layer.PoolOptions)
            hilo = any(t is not None and t.dtype != dt for t in (q, w_in, b_in, w_out, b_out))
        if hilo and need_bwd and dt == torch.bfloat16 and hilo_bytes > 0:
            hilo_ws = hilo_bytes
            saved_o_lo = torch.empty(B, E, dtype=dt, device=dev)
            flags |= _lib.AECF_HILO_GRADS
        ph_seed = ph_off = ph_threads = ph_elem0 = 0
        if philox is not None and mask_mode == 1 and uniforms is None:
            ph_seed, ph_off, ph_threads, ph_elem0 = philox
            flags |= _lib.AECF_DRAW_UNIFORMS
        if prep_ready:
            flags |= _lib.AECF_PREP_READY
        ws_bytes = fwd_ws_bytes
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        args = _lib.PoolFwdArgs(
            _ptr(xc), _ptr(qc), _ptr(w_in_c), _ptr(b_in_c), _ptr(w_out_c), _ptr(b_out_c), _ptr(kpm),
            _ptr(uniforms), _ptr(y), _ptr(attn_w), _ptr(masked_w), _ptr(entropy), _ptr(mask_rate),
            _ptr(probs), _ptr(saved_o), _ptr(saved_v), _ptr(ws), ws_bytes,
            None if opts.stage_events is None else ctypes.addressof(opts.stage_events[0]),
            _ptr(i_attn_w), _ptr(i_masked_w), _ptr(i_entropy), _ptr(i_mask_rate), _ptr(saved_prep),
            _ptr(i_target), 0.0 if target_value is None else float(target_value), flags, _ptr(ent_partial),
            ph_seed, ph_off, ph_threads, _ptr(ent_loss), _ptr(saved_o_lo), ph_elem0)
        _lib.check(lib.aecf_pool_forward(ctypes.byref(desc), ctypes.byref(args), _stream()), "aecf_pool_forward")
        ctx.save_for_backward(xc, qc, w_in_c, b_in_c, w_out_c, probs, saved_o, attn_w, saved_v, saved_prep, saved_o_lo)
        ctx.desc = desc
        ctx.opts = opts
        ctx.q_leaf = bool(q.is_leaf)
        if opts.dp is not None and q.is_leaf and q.requires_grad:
            opts.dp.add_scaled(q)                  # its gradient will carry grad_scale (dp.all_reduce_grads asks)
        ctx.bwd_ws_bytes = hilo_ws if saved_o_lo is not None else bwd_ws_bytes
        ctx.q_shape = q.shape
        ctx.param_dtypes = (q.dtype, w_in.dtype, None if b_in is None else b_in.dtype, w_out.dtype,
                            None if b_out is None else b_out.dtype)
        ctx.has_bias = (b_in is not None, b_out is not None)
        if i_attn_w is not None:
            outs = (y, i_attn_w, i_masked_w, i_entropy, mask_rate if mask_mode == 1 else i_mask_rate, i_target)
        else:
            outs = (y, attn_w, masked_w, entropy, mask_rate, i_target)
        nondiff = [t for t in (outs[2], outs[4], outs[5]) if t is not None]
        if outs[3] is not None and mask_mode == 1:
            nondiff.append(outs[3])            # train mode: entropy is detached (ref :278)
        ctx.mark_non_differentiable(*nondiff)
        return outs

    @staticmethod
    def backward(ctx, dy, d_attn_w, _d_masked, d_entropy, _d_rate, _d_target=None):
        lib = _lib.load()
        xc, qc, w_in_c, b_in_c, w_out_c, probs, saved_o, attn_w, saved_v, saved_prep, saved_o_lo = ctx.saved_tensors
        desc = ctx.desc
        B, M, E = xc.shape
        dev = xc.device
        dt = xc.dtype
        dy_c = torch.zeros(B, E, dtype=dt, device=dev) if dy is None else dy.to(dt).contiguous()
        daw = None if d_attn_w is None else d_attn_w.to(torch.float32).contiguous()
        dent = None
        if d_entropy is not None and desc.mask_mode == 2:
            dent = d_entropy.to(torch.float32).contiguous()
        dx = torch.empty_like(xc)
        # parameter gradients: in the parameters' own dtype when that is the activation dtype (the library rounds
        # its float32 batch sums once), float32 otherwise (e.g. float32 master weights under bf16 activations)
        qd, wid, bid, wod, bod = ctx.param_dtypes
        gdt = dt if all(p is None or p == dt for p in ctx.param_dtypes) else torch.float32
        # one allocation for the five of them: autograd keeps these tensors as p.grad without copying, so a data-parallel
        # caller can all-reduce the whole run in place with a single collective (aecf_amd/dp.py: all_reduce_grads)
        opts = ctx.opts
        dp_ = opts.dp
        keep32 = dp_ is not None and dp_.keep_f32 and gdt != torch.float32
        gscale = dp_.grad_scale if dp_ is not None else 1.0          # (1 / world: the average is then ONE sum all-reduce)
        out_dt = torch.float32 if keep32 else gdt                 # what the library writes
        flat = torch.empty(4 * E * E + 5 * E, dtype=out_dt, device=dev)
        dquery, dw_in, db_in, dw_out, db_out = flat.split([E, 3 * E * E, 3 * E, E * E, E])
        dw_in, dw_out = dw_in.view(3 * E, E), dw_out.view(E, E)
        ws_bytes = ctx.bwd_ws_bytes
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        hook, early = (None if dp_ is None else dp_.hook), None
        if hook is not None and any(p is not None and p != gdt for p in ctx.param_dtypes):
            hook = None        # the gradients handed to autograd would be cast COPIES of `flat`: nothing to reduce in place
        if hook is not None:
            early = torch.cuda.Event()
            early.record()                         # (creates the underlying hipEvent; the library records it again later)
        args = _lib.PoolBwdArgs(
            _ptr(xc), _ptr(qc), _ptr(w_in_c), _ptr(b_in_c), _ptr(w_out_c), _ptr(dy_c), _ptr(daw), _ptr(dent),
            _ptr(attn_w), _ptr(probs), _ptr(saved_o), _ptr(saved_v), _ptr(dx), _ptr(dquery), _ptr(dw_in), _ptr(db_in),
            _ptr(dw_out), _ptr(db_out), _ptr(ws), ws_bytes,
            None if opts.stage_events is None else ctypes.addressof(opts.stage_events[1]),
            _DTYPES[out_dt], 0 if saved_o_lo is None else _lib.AECF_HILO_GRADS, _ptr(saved_prep),
            None if early is None else early.cuda_event, _ptr(saved_o_lo), float(gscale))
        _lib.check(lib.aecf_pool_backward(ctypes.byref(desc), ctypes.byref(args), _stream()), "aecf_pool_backward")
        flat32 = None
        if keep32:
            # the float32 run stays behind for the collective, which rounds the MEAN once into the allocation autograd holds.
            # What autograd gets meanwhile: this rank's sums rounded to the parameters' dtype -- unless the collective is certain
            # to overwrite them before anyone may look (GradOverlap's hook is about to mutate flat32 on its side stream, so a cast
            # on this stream would race it; DpState.defer_rounding: the caller promised to call all_reduce_grads): then the
            # tensors stay uninitialised and the backward has no cast launch at all
            flat32 = flat
            defer = hook is not None or dp_.defer_rounding
            if defer and hook is None and dp_.runs:
                # an earlier run of this step is still unreduced and autograd is about to ADD this call's gradients to it (two
                # pool applications, micro-batches): both get their real values now
                for f_, w_, v_ in dp_.runs:
                    if w_ is not None and f_._version == v_:
                        f_.copy_(w_)
                defer = False
            flat = torch.empty_like(flat32, dtype=gdt) if defer else flat32.to(gdt)
            dquery, dw_in, db_in, dw_out, db_out = flat.split([E, 3 * E * E, 3 * E, E * E, E])
            dw_in, dw_out = dw_in.view(3 * E, E), dw_out.view(E, E)
        if dp_ is not None:
            dp_.record(flat, flat32)
        if hook is not None:
            hook(flat, early, flat32, dp_)         # [dquery | dw_in | db_in | dw_out | db_out]: final once `early` has fired
        needs = ctx.needs_input_grad
        if gscale != 1.0 and not ctx.q_leaf and needs[1]:
            dquery = dquery * (1.0 / gscale)       # a COMPUTED query: its gradient flows on into whoever made it, unscaled like dx
        return (dx if needs[0] else None,
                dquery.to(qd).reshape(ctx.q_shape) if needs[1] else None,
                dw_in.to(wid) if needs[2] else None,
                db_in.to(bid) if (ctx.has_bias[0] and needs[3]) else None,
                dw_out.to(wod) if needs[4] else None,
                db_out.to(bod) if (ctx.has_bias[1] and needs[5]) else None,
                None, None, None, None, None, None, None, None, None, None, None, None, None, None)


def precise_forward_backward(x: torch.Tensor, query: torch.Tensor, w_in: torch.Tensor, b_in: Optional[torch.Tensor],
                             w_out: torch.Tensor, b_out: Optional[torch.Tensor], num_heads: int, dy: torch.Tensor,
                             d_attn_w: Optional[torch.Tensor] = None, key_padding_mask: Optional[torch.Tensor] = None,
                             ) -> Dict[str, torch.Tensor]:
    """The float32-STORE form of the bf16 path (``AECF_PRECISE``, include/aecf_hip.h): bf16 ``x`` / ``query`` / weights /
    ``dy``, bf16 MFMA for every product of exact bf16 operands, no intermediate rounded to bf16, float32 outputs.  This is
    what BASELINE.json's "within 1e-3 relative, bf16" is asserted on: a bf16-STORED output cannot
    meet 1e-3 (one output rounding alone is 2^-9 of the element).  Not an autograd node (autograd wants bf16 gradients for
    bf16 tensors): forward and backward are called back to back and everything comes back in one dictionary."""
    lib = _lib.load()
    bf = torch.bfloat16
    B, M, E = x.shape
    dev = x.device
    _require_device(x, "x")
    desc = _lib.PoolDesc(B, M, E, num_heads, _lib.AECF_BF16, 0, 1, 0.15, 0.7, 1e-8)
    _lib.check(lib.aecf_pool_check(ctypes.byref(desc)), "aecf_pool_check")
    c = lambda t: None if t is None else t.detach().to(device=dev, dtype=bf).contiguous()
    xc, qc, w_in_c, b_in_c, w_out_c, b_out_c = c(x), c(query.reshape(E)), c(w_in), c(b_in), c(w_out), c(b_out)
    dyc = c(dy.reshape(B, E))
    kpm = None if key_padding_mask is None else key_padding_mask.to(device=dev, dtype=torch.uint8).contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    y, attn_w, probs, o = (torch.empty(B, E, **f32), torch.empty(B, M, **f32), torch.empty(B, num_heads, M, **f32),
                           torch.empty(B, E, **f32))
    ws_bytes = lib.aecf_pool_precise_workspace_bytes(ctypes.byref(desc), 0)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    fa = _lib.PoolFwdArgs(_ptr(xc), _ptr(qc), _ptr(w_in_c), _ptr(b_in_c), _ptr(w_out_c), _ptr(b_out_c), _ptr(kpm), None,
                          _ptr(y), _ptr(attn_w), None, None, None, _ptr(probs), _ptr(o), None, _ptr(ws), ws_bytes, None,
                          None, None, None, None, None, None, 0.0, _lib.AECF_PRECISE)
    _lib.check(lib.aecf_pool_forward(ctypes.byref(desc), ctypes.byref(fa), _stream()), "aecf_pool_forward (precise)")
    daw = None if d_attn_w is None else d_attn_w.detach().to(device=dev, dtype=torch.float32).reshape(B, M).contiguous()
    dx = torch.empty(B, M, E, **f32)
    flat = torch.empty(4 * E * E + 5 * E, **f32)
    dquery, dw_in, db_in, dw_out, db_out = flat.split([E, 3 * E * E, 3 * E, E * E, E])
    bws_bytes = lib.aecf_pool_precise_workspace_bytes(ctypes.byref(desc), 1)
    bws = torch.empty(bws_bytes, dtype=torch.uint8, device=dev)
    ba = _lib.PoolBwdArgs(_ptr(xc), _ptr(qc), _ptr(w_in_c), _ptr(b_in_c), _ptr(w_out_c), _ptr(dyc), _ptr(daw), None,
                          _ptr(attn_w), _ptr(probs), _ptr(o), None, _ptr(dx), _ptr(dquery), _ptr(dw_in), _ptr(db_in),
                          _ptr(dw_out), _ptr(db_out), _ptr(bws), bws_bytes, None, _lib.AECF_F32, _lib.AECF_PRECISE, None)
    _lib.check(lib.aecf_pool_backward(ctypes.byref(desc), ctypes.byref(ba), _stream()), "aecf_pool_backward (precise)")
    return dict(y=y.view(B, 1, E), wbar=attn_w.view(B, 1, M), probs=probs, dx=dx, dquery=dquery.view(1, 1, E),
                dw_in=dw_in.view(3 * E, E), db_in=db_in, dw_out=dw_out.view(E, E), db_out=db_out)


class _MaskFunction(torch.autograd.Function):
    """aecf_curriculum_mask_forward / _backward on free-standing weight rows."""

    @staticmethod
    def forward(ctx, weights, uniforms, mode, min_active, base_mask_prob, entropy_target, eps):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        L = weights.shape[-1]
        rows = weights.numel() // L
        w32 = weights.detach().to(torch.float32).contiguous()
        dev = weights.device
        masked = torch.empty_like(w32)
        entropy = torch.empty(weights.shape[:-1], dtype=torch.float32, device=dev)
        mask_rate = torch.empty(weights.shape[:-1], dtype=torch.float32, device=dev)
        bits = torch.empty(weights.shape, dtype=torch.uint8, device=dev)
        _lib.check(lib.aecf_curriculum_mask_forward(
            rows, L, mode, min_active, base_mask_prob, entropy_target, eps, _ptr(w32), _ptr(uniforms), _ptr(masked),
            _ptr(entropy), _ptr(mask_rate), _ptr(bits), _stream()), "aecf_curriculum_mask_forward")
        ctx.save_for_backward(w32, bits)
        ctx.cfg = (rows, L, mode, eps, weights.dtype)
        if mode == 1:
            ctx.mark_non_differentiable(entropy, mask_rate)
        else:
            ctx.mark_non_differentiable(mask_rate)
        return masked, entropy, mask_rate

    @staticmethod
    def backward(ctx, d_masked, d_entropy, _d_rate):
        lib = _lib.load()
        w32, bits = ctx.saved_tensors
        rows, L, mode, eps, wdt = ctx.cfg
        dm = None if d_masked is None else d_masked.to(torch.float32).contiguous()
        de = None if (d_entropy is None or mode != 2) else d_entropy.to(torch.float32).contiguous()
        dw = torch.empty_like(w32)
        _lib.check(lib.aecf_curriculum_mask_backward(rows, L, mode, eps, _ptr(w32), _ptr(bits), _ptr(dm), _ptr(de),
                                                     _ptr(dw), _stream()), "aecf_curriculum_mask_backward")
        return dw.to(wdt), None, None, None, None, None, None


class _EntropyLossFunction(torch.autograd.Function):
    """aecf_entropy_loss_fwd_bwd: the loss and its gradient come out of one launch pair."""

    @staticmethod
    def forward(ctx, entropy, last_seq_len, entropy_target):
        lib = _lib.load()
        ec = entropy.detach()
        if ec.dtype not in _DTYPES:
            ec = ec.to(torch.float32)
        ec = ec.contiguous()
        n = ec.numel()
        dev = entropy.device
        loss = torch.empty(1, dtype=ec.dtype, device=dev)
        dent = torch.empty(ec.shape, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.aecf_entropy_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(lib.aecf_entropy_loss_fwd_bwd(n, _DTYPES[ec.dtype], last_seq_len, entropy_target, _ptr(ec), 1.0,
                                                 _ptr(loss), _ptr(dent), _ptr(ws), _stream()),
                   "aecf_entropy_loss_fwd_bwd")
        ctx.save_for_backward(dent)
        ctx.edtype = entropy.dtype
        return loss.reshape(()).to(entropy.dtype)

    @staticmethod
    def backward(ctx, dloss):
        (dent,) = ctx.saved_tensors
        return (dent * dloss.to(torch.float32)).to(ctx.edtype), None, None


class _MhaFunction(torch.autograd.Function):
    """aecf_mha_forward / aecf_mha_backward: the general nn.MultiheadAttention case (per-sample queries, tgt_len > 1,
    key != value, attn_mask, key_padding_mask, dropout).  Batch-major [B,T,E] / [B,S,E].  Outputs y [B,T,E] and the
    head-averaged (post-dropout) weights [B,T,S] float32."""

    @staticmethod
    def forward(ctx, query, key, value, w_in, b_in, w_out, b_out, attn_mask, mask_stride, kpm, drop_u, drop_p,
                num_heads):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        B, T, E = query.shape
        S = key.shape[1]
        dt = query.dtype
        dev = query.device
        desc = _lib.MhaDesc(B, T, S, E, num_heads, _DTYPES[dt], float(drop_p))
        _lib.check(lib.aecf_mha_check(ctypes.byref(desc)), "aecf_mha_check")
        qc, kc, vc = query.detach().contiguous(), key.detach().to(dt).contiguous(), value.detach().to(dt).contiguous()
        w_in_c, w_out_c = w_in.detach().to(dt).contiguous(), w_out.detach().to(dt).contiguous()
        b_in_c = None if b_in is None else b_in.detach().to(dt).contiguous()
        b_out_c = None if b_out is None else b_out.detach().to(dt).contiguous()
        y = torch.empty(B, T, E, dtype=dt, device=dev)
        attn_w = torch.empty(B, T, S, dtype=torch.float32, device=dev)
        sq = torch.empty(B * T, E, dtype=dt, device=dev)
        sk = torch.empty(B * S, E, dtype=dt, device=dev)
        sv = torch.empty(B * S, E, dtype=dt, device=dev)
        so = torch.empty(B * T, E, dtype=dt, device=dev)
        probs = torch.empty(B, num_heads, T, S, dtype=torch.float32, device=dev)
        args = _lib.MhaFwdArgs(_ptr(qc), _ptr(kc), _ptr(vc), _ptr(w_in_c), _ptr(b_in_c), _ptr(w_out_c), _ptr(b_out_c),
                               _ptr(attn_mask), int(mask_stride), _ptr(kpm), _ptr(drop_u), _ptr(y), _ptr(attn_w),
                               _ptr(sq), _ptr(sk), _ptr(sv), _ptr(so), _ptr(probs))
        _lib.check(lib.aecf_mha_forward(ctypes.byref(desc), ctypes.byref(args), _stream()), "aecf_mha_forward")
        ctx.save_for_backward(qc, kc, vc, w_in_c, w_out_c, drop_u, sq, sk, sv, so, probs)
        ctx.desc = desc
        ctx.in_dtypes = (key.dtype, value.dtype, w_in.dtype, None if b_in is None else b_in.dtype, w_out.dtype,
                         None if b_out is None else b_out.dtype)
        return y, attn_w

    @staticmethod
    def backward(ctx, dy, d_attn_w):
        lib = _lib.load()
        qc, kc, vc, w_in_c, w_out_c, drop_u, sq, sk, sv, so, probs = ctx.saved_tensors
        desc = ctx.desc
        B, T, E = qc.shape
        S = kc.shape[1]
        dt, dev = qc.dtype, qc.device
        dy_c = torch.zeros(B, T, E, dtype=dt, device=dev) if dy is None else dy.to(dt).contiguous()
        daw = None if d_attn_w is None else d_attn_w.to(torch.float32).contiguous()
        dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
        f32 = dict(dtype=torch.float32, device=dev)
        dw_in, db_in = torch.empty(3 * E, E, **f32), torch.empty(3 * E, **f32)
        dw_out, db_out = torch.empty(E, E, **f32), torch.empty(E, **f32)
        ws_bytes = lib.aecf_mha_bwd_workspace_bytes(ctypes.byref(desc))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        args = _lib.MhaBwdArgs(_ptr(qc), _ptr(kc), _ptr(vc), _ptr(w_in_c), _ptr(w_out_c), _ptr(drop_u), _ptr(dy_c),
                               _ptr(daw), _ptr(sq), _ptr(sk), _ptr(sv), _ptr(so), _ptr(probs), _ptr(dq), _ptr(dk),
                               _ptr(dv), _ptr(dw_in), _ptr(db_in), _ptr(dw_out), _ptr(db_out), _ptr(ws), ws_bytes)
        _lib.check(lib.aecf_mha_backward(ctypes.byref(desc), ctypes.byref(args), _stream()), "aecf_mha_backward")
        kd, vd, wid, bid, wod, bod = ctx.in_dtypes
        needs = ctx.needs_input_grad
        return (dq if needs[0] else None, dk.to(kd) if needs[1] else None, dv.to(vd) if needs[2] else None,
                dw_in.to(wid) if needs[3] else None, db_in.to(bid) if (bid is not None and needs[4]) else None,
                dw_out.to(wod) if needs[5] else None, db_out.to(bod) if (bod is not None and needs[6]) else None,
                None, None, None, None, None, None)


class _SdpaFunction(torch.autograd.Function):
    """aecf_sdpa_forward / _backward (projection-free single-head attention, ref :556-581)."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        lib = _lib.load()
        B, S, E = q.shape
        T = k.shape[1]
        dt = q.dtype
        qc, kc, vc = q.contiguous(), k.to(dt).contiguous(), v.to(dt).contiguous()
        out = torch.empty(B, S, E, dtype=dt, device=q.device)
        probs = torch.empty(B, S, T, dtype=torch.float32, device=q.device)
        _lib.check(lib.aecf_sdpa_forward(B, S, T, E, _DTYPES[dt], scale, _ptr(qc), _ptr(kc), _ptr(vc), _ptr(out),
                                         _ptr(probs), _stream()), "aecf_sdpa_forward")
        ctx.save_for_backward(qc, kc, vc, probs)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        qc, kc, vc, probs = ctx.saved_tensors
        B, S, E = qc.shape
        T = kc.shape[1]
        dt = qc.dtype
        do = dout.to(dt).contiguous()
        dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
        _lib.check(lib.aecf_sdpa_backward(B, S, T, E, _DTYPES[dt], ctx.scale, _ptr(qc), _ptr(kc), _ptr(vc), _ptr(probs),
                                          _ptr(do), _ptr(dq), _ptr(dk), _ptr(dv), _stream()), "aecf_sdpa_backward")
        return dq, dk, dv, None


# ----------------------------------------------------------------------------------------------
# ref: aecf/AECFLayer.py:33-319
# ----------------------------------------------------------------------------------------------
class CurriculumMasking(nn.Module):
    r"""Entropy-driven curriculum masking for attention weights (ref: aecf/AECFLayer.py:33-319).

    Same constructor, ``forward`` contract and ``entropy_loss`` as the reference.  The Bernoulli
    draw of ref :204 is made with one float32 uniform per weight element from torch's default
    generator of the weights' device (``mask = U < keep_prob``), so the generator advances by
    exactly ``weights.numel()`` draws as ``torch.bernoulli`` would.
    """

    def __init__(self, base_mask_prob: float = 0.15, entropy_target: float = 0.7, min_active: int = 1):
        super().__init__()
        if not 0.0 < base_mask_prob <= 1.0:                                   # ref :84-89
            raise ValueError(f"base_mask_prob must be in (0, 1], got {base_mask_prob}")
        if not 0.0 < entropy_target <= 1.0:
            raise ValueError(f"entropy_target must be in (0, 1], got {entropy_target}")
        if min_active < 1:
            raise ValueError(f"min_active must be >= 1, got {min_active}")
        self.base_mask_prob = base_mask_prob
        self.entropy_target = entropy_target
        self.min_active = min_active
        self.register_buffer('_eps', torch.tensor(1e-8))                      # ref :96
        self._last_seq_len = 2                                                # ref :99

    def _mode(self) -> int:
        return 1 if self.training else 2

    def compute_entropy(self, weights: torch.Tensor) -> torch.Tensor:
        """ref :101-128 -- Shannon entropy of the rows, clamped to [0, log L]."""
        return self.compute_entropy_fused(weights)

    def compute_entropy_fused(self, weights: torch.Tensor) -> torch.Tensor:
        _require_device(weights, "weights")
        _, entropy, _ = _MaskFunction.apply(weights, None, 2, self.min_active, float(self.base_mask_prob),
                                            float(self.entropy_target), 1e-8)
        return entropy.to(weights.dtype)

    def forward(self, weights: torch.Tensor, *, uniforms: Optional[torch.Tensor] = None,
                generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """ref :130-283.  ``uniforms`` / ``generator`` (keyword-only, not in the reference): where the Bernoulli
        draw of ref :204 takes its float32 uniforms from -- see ``_draw_uniforms``."""
        _require_device(weights, "weights")
        dt = weights.dtype
        seq_len = weights.size(-1)
        if not self.training:                                                 # ref :150-156
            _, entropy, mask_rate = _MaskFunction.apply(weights, None, 2, self.min_active,
                                                        float(self.base_mask_prob), float(self.entropy_target), 1e-8)
            return weights, {'entropy': entropy.to(dt), 'mask_rate': mask_rate.to(dt)}
        if seq_len <= 1:                                                      # ref :160-167
            z = torch.zeros(weights.shape[:-1], device=weights.device, dtype=dt)
            return weights, {'entropy': z, 'mask_rate': z.clone(), 'target_entropy': z.clone()}
        uniforms = _draw_uniforms(tuple(weights.shape), weights.device, uniforms, generator)   # ref :204
        masked, entropy, mask_rate = _MaskFunction.apply(weights, uniforms, 1, self.min_active,
                                                         float(self.base_mask_prob), float(self.entropy_target), 1e-8)
        self._last_seq_len = seq_len                                          # ref :187
        entropy = entropy.to(dt)
        info = {
            'entropy': entropy,
            'mask_rate': mask_rate,                                           # always float32 (ref :275)
            'target_entropy': torch.full_like(entropy, math.log(float(seq_len)) * self.entropy_target),
        }
        return masked.to(dt), info

    def entropy_loss(self, entropy: torch.Tensor) -> torch.Tensor:
        """ref :285-314 -- MSE between entropy and log(L_last) * entropy_target."""
        _require_device(entropy, "entropy")
        seq_len = self._last_seq_len if hasattr(self, '_last_seq_len') else 2
        tag = getattr(entropy, "_aecf_entropy_partials", None)
        if tag is not None and not entropy.requires_grad and entropy.dtype in _DTYPES:
            # info['entropy'] of a fused pool forward, untouched: the kernel that wrote it left the per-block sums of
            # (nan_to_num(H) - target)^2 behind (aecf_pool_fwd_args.ent_loss_partial) -- one small launch adds them up
            (partial, n, target, ready), version = tag
            want = math.log(float(seq_len)) * self.entropy_target if seq_len > 1 else 0.0
            # ... and UNTOUCHED is checked, not assumed: an in-place edit of the tensor (info['entropy'].clamp_(...), .mul_())
            # or of any view of it moves the version counter they share, and the partial sums no longer describe the values:
            # the full kernel below reads the tensor as it is now
            if entropy._version == version and entropy.numel() == n and abs(want - target) <= 1e-6 * max(1.0, abs(want)):
                if ready is not None and ready.dtype == entropy.dtype:
                    return ready.reshape(())       # the forward's out-projection launch already added the partials up
                loss = torch.empty(1, dtype=entropy.dtype, device=entropy.device)
                _lib.check(_lib.load().aecf_entropy_loss_from_partials(n, _DTYPES[entropy.dtype], _ptr(partial), _ptr(loss),
                                                                       _stream()), "aecf_entropy_loss_from_partials")
                return loss.reshape(())
        return _EntropyLossFunction.apply(entropy, int(seq_len), float(self.entropy_target))

    def extra_repr(self) -> str:                                              # ref :316-319
        return (f'base_mask_prob={self.base_mask_prob}, '
                f'entropy_target={self.entropy_target}, '
                f'min_active={self.min_active}')


# ----------------------------------------------------------------------------------------------
# ref: aecf/AECFLayer.py:322-552
# ----------------------------------------------------------------------------------------------
def _shared_query_base(query: torch.Tensor) -> Optional[torch.Tensor]:
    """If ``query`` is ``p.expand(B, -1, -1)`` of a [1,1,E] tensor (ref :694-695), return p so the gradient
    reaches it directly (no [B,1,E] gradient is ever materialised)."""
    B, T, E = query.shape
    if T != 1:
        return None
    if B > 1 and query.stride(0) != 0:
        return None
    base = query._base if query._is_view() else None
    if base is not None and base.dim() == 3 and tuple(base.shape) == (1, 1, E) and base.stride(-1) == query.stride(-1):
        return base
    return query[:1]


def _padded_embed(E: int, H: int, dtype: torch.dtype) -> Optional[Tuple[int, int]]:
    """(E', head_dim') of the smallest per-head padding the general kernels tile: E' = H * head_dim' a multiple of 64 (bf16) /
    32 (float32).  None when E already is one."""
    unit = 64 if dtype == torch.bfloat16 else 32
    if E % H != 0 or E % unit == 0:
        return None
    hd = E // H
    g = unit // math.gcd(unit, H)
    hd2 = (hd + g - 1) // g * g
    return H * hd2, hd2


def _pad_heads(w_in, b_in, w_out, b_out, E: int, H: int, pad_to: Tuple[int, int]):
    """The parameters of an E-wide attention as those of an E'-wide one that computes the same function on zero-padded
    inputs: every head keeps its head_dim rows of W_q / W_k / W_v (and columns of W_o) at the start of a head_dim'-wide
    slot, the rest is zero; W_q and b_q carry sqrt(head_dim' / head_dim), because the kernels scale the scores by
    1 / sqrt(head_dim').  Differentiable torch ops: the gradients reach the original parameters through them."""
    E2, hd2 = pad_to
    hd = E // H
    pad = torch.nn.functional.pad
    s = math.sqrt(hd2 / hd)

    def rows(w, scale):                                   # [E, E] -> [E', E']
        return pad(w.reshape(H, hd, E), (0, E2 - E, 0, hd2 - hd)).reshape(E2, E2) * scale

    w_in2 = torch.cat([rows(w_in[:E], s), rows(w_in[E:2 * E], 1.0), rows(w_in[2 * E:], 1.0)])
    b_in2 = None
    if b_in is not None:
        vec = lambda b, scale: pad(b.reshape(H, hd), (0, hd2 - hd)).reshape(E2) * scale
        b_in2 = torch.cat([vec(b_in[:E], s), vec(b_in[E:2 * E], 1.0), vec(b_in[2 * E:], 1.0)])
    w_out2 = pad(pad(w_out.reshape(E, H, hd), (0, hd2 - hd)).reshape(E, E2), (0, 0, 0, E2 - E))
    b_out2 = None if b_out is None else pad(b_out, (0, E2 - E))
    return w_in2, b_in2, w_out2, b_out2


class MultimodalAttentionPool(nn.Module):
    r"""Multimodal attention pooling with optional curriculum masking (ref: aecf/AECFLayer.py:322-552).

    Parameters live in a real ``torch.nn.MultiheadAttention`` under ``self.attention`` (same init RNG
    order and state_dict keys as the reference); its ``forward`` is never called -- the arithmetic runs
    in ``libaecf_hip.so``.

    Hot path (the fused kernels): one query shared by the batch (``fusion_query.expand(B, -1, -1)``, tgt_len 1),
    ``value is key``, optional boolean ``key_padding_mask``, dropout 0, bf16 or fp32.  Every other argument combination
    ``nn.MultiheadAttention`` takes here (per-sample queries, tgt_len > 1, key != value, attn_mask, float masks, attention
    dropout, embedding sizes the fused kernels do not tile) runs on the general kernels (``_forward_general``); only dtypes
    other than bfloat16 / float32 raise NotImplementedError.  Nothing is routed to PyTorch.

    ``self.options`` (``PoolOptions``): per-module switches of the fused path; ``aecf_amd.dp.attach(pool)`` makes the module
    data-parallel (gradients pre-scaled by 1 / world, float32 sums kept for the collective).
    """

    def __init__(self, embed_dim: int, num_heads: int = 1, dropout: float = 0.0, bias: bool = True,
                 curriculum_masking: Optional[CurriculumMasking] = None, batch_first: bool = True,
                 device: Optional[torch.device] = None, dtype: Optional[torch.dtype] = None):
        super().__init__()
        if embed_dim <= 0:                                                    # ref :384-391
            raise ValueError(f"embed_dim must be positive, got {embed_dim}")
        if num_heads <= 0:
            raise ValueError(f"num_heads must be positive, got {num_heads}")
        if embed_dim % num_heads != 0:
            raise ValueError(f"embed_dim ({embed_dim}) must be divisible by num_heads ({num_heads})")
        if not 0.0 <= dropout <= 1.0:
            raise ValueError(f"dropout must be in [0, 1], got {dropout}")
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.curriculum_masking = curriculum_masking
        self.attention = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads, dropout=dropout,    # ref :399-407
                                               bias=bias, batch_first=batch_first, device=device, dtype=dtype)
        self._cast_cache: Dict[str, Any] = {}
        self._prep_cache: Optional[Tuple[Tuple, torch.Tensor]] = None       # (key, preparation buffer) while the parameters stand still
        self.options = PoolOptions()

    def _options(self) -> PoolOptions:
        o = self.__dict__.get("options")
        if o is None:                              # (a module unpickled from a build that had no per-module options)
            o = self.options = PoolOptions()
        return o

    def invalidate_cast_cache(self) -> None:
        """Forget the activation-dtype copies of the parameters (see _activation_dtype_params).  Call it after changing a
        parameter through ``p.data`` (``p.data.copy_(...)``, EMA / weight swapping, optimizers that step on ``.data``) while
        the module is in eval mode under ``torch.no_grad()``: such writes do not move the parameter's version counter."""
        self._cast_cache.clear()
        self._prep_cache = None

    def _activation_dtype_params(self, dt: torch.dtype, query: Optional[torch.Tensor] = None):
        """Master weights kept in another dtype than the activations (float32 parameters, bf16 data): the kernels want
        them in the activation dtype.  Inference (eval mode, or no gradient recording) reuses the copies while the
        parameters' version counters and storage stand still; whenever the module trains they are remade on every forward:
        an optimizer that steps through ``p.data`` (apex / DeepSpeed style, EMA, clipping on ``.data``) leaves the version
        counter alone, and stale weights would go unnoticed.  ``invalidate_cast_cache()`` covers ``.data`` writes in eval."""
        a = self.attention
        reuse = not (self.training and torch.is_grad_enabled())
        named = (("w_in", a.in_proj_weight), ("b_in", a.in_proj_bias), ("w_out", a.out_proj.weight), ("b_out", a.out_proj.bias))
        if not reuse:
            # training: ONE cast launch for the four of them + the query (aecf_cast_f32_to_bf16; one torch launch each was ~22 us
            # of a 0.7 ms step, and torch's multi-tensor copy takes as long: it hands a block 65536 elements) into fresh
            # allocations (the backward keeps them: a later forward must not write over what an earlier one saved)
            srcs = [p.detach() for _, p in named if p is not None]
            ride = query is not None and query.dtype != dt and query.is_contiguous()      # the shared query travels along
            if ride:
                srcs.append(query.detach())
            dsts = [torch.empty(p.shape, dtype=dt, device=p.device) for p in srcs]
            if (dt == torch.bfloat16 and all(p.dtype == torch.float32 and p.is_cuda and p.is_contiguous() for p in srcs)):
                n = len(srcs)
                vp = ctypes.c_void_p
                _lib.check(_lib.load().aecf_cast_f32_to_bf16(
                    n, (vp * n)(*[p.data_ptr() for p in srcs]), (vp * n)(*[d.data_ptr() for d in dsts]),
                    (ctypes.c_int64 * n)(*[p.numel() for p in srcs]), _stream()), "aecf_cast_f32_to_bf16")
            else:
                torch._foreach_copy_(dsts, srcs)
            it = iter(dsts)
            return tuple(None if p is None else next(it) for _, p in named) + ((dsts[-1],) if ride else (None,))
        out = []
        for name, p in named:
            if p is None:
                out.append(None)
                continue
            key = (p._version, p.data_ptr(), p.device, dt)
            hit = self._cast_cache.get(name)
            if hit is None or hit[0] != key:
                hit = (key, p.detach().to(dt).contiguous())
                self._cast_cache[name] = hit
            out.append(hit[1])
        return tuple(out) + (None,)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: Optional[torch.Tensor] = None,
                key_padding_mask: Optional[torch.Tensor] = None, attn_mask: Optional[torch.Tensor] = None,
                return_info: bool = False, use_checkpoint: bool = False, *,
                uniforms: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                batch_shard: Optional[Tuple[int, int]] = None,
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, Dict[str, Any]]]:
        # ``uniforms`` / ``generator`` (keyword-only, not in the reference): source of the curriculum mask's float32
        # uniforms, [B, tgt_len, src_len] (see _draw_uniforms).  Ignored when no training-mode curriculum masking runs.
        # ``batch_shard = (first_row, global_batch)`` (keyword-only, data parallel): this call's rows are rows
        # [first_row, first_row + B) of a global batch whose mask uniforms are ONE draw of [global_batch, tgt_len, src_len]
        # from ``generator`` (default: the device's default generator) -- every rank seeds that generator alike and names its
        # shard; the statistics kernel evaluates the rank's elements of the global draw (no tensor, no launch) and the
        # generator advances as the global call would, so N-rank masks are the one-rank masks bit for bit.
        if batch_shard is not None:
            row0, global_batch = int(batch_shard[0]), int(batch_shard[1])
            nrows = query.shape[0] if self.batch_first else query.shape[1]
            if row0 < 0 or row0 + nrows > global_batch:
                raise ValueError(f"batch_shard {batch_shard} does not hold this call's {nrows} rows")
        # type / shape validation: ref :450-498 (messages identical)
        if not isinstance(query, torch.Tensor):
            raise TypeError(f"Expected query to be torch.Tensor, got {type(query)}")
        if not isinstance(key, torch.Tensor):
            raise TypeError(f"Expected key to be torch.Tensor, got {type(key)}")
        if value is not None and not isinstance(value, torch.Tensor):
            raise TypeError(f"Expected value to be torch.Tensor or None, got {type(value)}")
        if value is None:
            value = key
        if self.batch_first:
            if query.dim() != 3:
                raise ValueError(f"Expected 3D query tensor with batch_first=True, got {query.dim()}D")
            if key.dim() != 3:
                raise ValueError(f"Expected 3D key tensor with batch_first=True, got {key.dim()}D")
            if value.dim() != 3:
                raise ValueError(f"Expected 3D value tensor with batch_first=True, got {value.dim()}D")
            batch_size, tgt_len, embed_dim = query.shape
            src_len = key.shape[1]
            if src_len == 0:
                raise ValueError("Key sequence length cannot be zero")
            if key.shape[0] != batch_size or key.shape[2] != embed_dim:
                raise RuntimeError(f"Key shape {key.shape} incompatible with query shape {query.shape}")
            if value.shape[0] != batch_size or value.shape[1] != key.shape[1] or value.shape[2] != embed_dim:
                raise RuntimeError(f"Value shape {value.shape} incompatible with key shape {key.shape}")
        else:
            if query.dim() != 3:
                raise ValueError(f"Expected 3D query tensor with batch_first=False, got {query.dim()}D")
            if key.dim() != 3:
                raise ValueError(f"Expected 3D key tensor with batch_first=False, got {key.dim()}D")
            if value.dim() != 3:
                raise ValueError(f"Expected 3D value tensor with batch_first=False, got {value.dim()}D")
            tgt_len, batch_size, embed_dim = query.shape
            src_len = key.shape[0]
            if src_len == 0:
                raise ValueError("Key sequence length cannot be zero")
            if key.shape[1] != batch_size or key.shape[2] != embed_dim:
                raise RuntimeError(f"Shape mismatch: query {query.shape}, key {key.shape}")
            if value.shape[0] != src_len or value.shape[1] != batch_size or value.shape[2] != embed_dim:
                raise RuntimeError(f"Value shape {value.shape} incompatible with key shape {key.shape}")
        if embed_dim != self.embed_dim:
            raise AssertionError(f"was expecting embedding dimension of {self.embed_dim}, but got {embed_dim}")

        _require_device(key, "key")
        _require_device(query, "query")
        if batch_size == 0:
            return self._forward_empty(query, key, tgt_len, src_len, return_info)
        same_kv = (value is key) or (value.data_ptr() == key.data_ptr() and value.shape == key.shape
                                     and value.stride() == key.stride())
        if key.dtype not in _DTYPES:
            raise NotImplementedError(f"aecf_amd: dtype {key.dtype} is not supported (bfloat16 / float32 only)")

        # to batch-first [B, M, E] / [B, 1, E] (ref: torch activation.py:1453-1463 does the inverse)
        if self.batch_first:
            q_bf, x = query, key
        else:
            q_bf, x = query.transpose(0, 1), key.transpose(0, 1)
        q_base = _shared_query_base(q_bf)
        dropping = self.attention.dropout > 0.0 and self.training
        # shapes the shared-query kernels do not take (more than 8 modalities, more than 16 heads, head sizes that are
        # not MFMA K-step multiples) are served by the general kernels as long as THEY take them
        fkey = (batch_size, src_len, embed_dim, self.num_heads, key.dtype, 0)
        facts = _shape_facts.get(fkey)
        if facts is None:
            facts = _pool_facts(_lib.load(), _lib.PoolDesc(batch_size, src_len, embed_dim, self.num_heads, _DTYPES[key.dtype],
                                                           0, 1, 0.15, 0.7, 1e-8), fkey)
        fast_ok = facts[0] == 0
        general_ok = fast_ok or _lib.load().aecf_mha_check(ctypes.byref(_lib.MhaDesc(
            batch_size, tgt_len, src_len, embed_dim, self.num_heads, _DTYPES[key.dtype], 0.0))) == 0
        float_kpm = key_padding_mask is not None and key_padding_mask.is_floating_point()   # additive in torch
        pad_to = None
        if not general_ok:
            # embedding sizes no kernel tiles (the reference takes any E % H == 0, ref :384-391): every head is padded with
            # zero rows / columns to the next size the general kernels take -- the same function of the inputs
            pad_to = _padded_embed(embed_dim, self.num_heads, key.dtype)
            if pad_to is not None and _lib.load().aecf_mha_check(ctypes.byref(_lib.MhaDesc(
                    batch_size, tgt_len, src_len, pad_to[0], self.num_heads, _DTYPES[key.dtype], 0.0))) != 0:
                pad_to = None
        if (q_base is None or not same_kv or attn_mask is not None or dropping or float_kpm
                or (not fast_ok and general_ok) or pad_to is not None):
            # everything outside the shared-query hot path: the general attention kernels
            if (batch_shard is not None and uniforms is None and self.curriculum_masking is not None
                    and self.curriculum_masking.training and src_len > 1):
                uniforms = _draw_uniforms((global_batch, tgt_len, src_len), key.device, None, generator)[row0:row0 + batch_size]
            return self._forward_general(q_bf, x, value if self.batch_first else value.transpose(0, 1),
                                         key_padding_mask, attn_mask, return_info, batch_size, tgt_len, src_len,
                                         uniforms, generator, pad_to)
        kpm = None
        if key_padding_mask is not None:
            if key_padding_mask.shape != (batch_size, src_len):
                raise RuntimeError(f"key_padding_mask shape {tuple(key_padding_mask.shape)} != {(batch_size, src_len)}")
            kpm = key_padding_mask.to(device=x.device, dtype=torch.uint8).contiguous()

        cm = self.curriculum_masking
        opts = self._options()
        mask_mode = 0
        mask_u = philox = None
        if cm is not None:
            mask_mode = 1 if cm.training else 2
            if mask_mode == 1 and src_len <= 1:
                mask_mode = 0          # ref :160-167 early-out: handled on the host below
            if mask_mode == 1:
                # one float32 uniform per weight element, row-major, default generator (ref :204): drawn by the statistics
                # kernel itself from the generator's (seed, offset) where that can be read on the host, else as a tensor
                draw_rows, row0 = (batch_size, 0) if (batch_shard is None or uniforms is not None) else (global_batch, row0)
                if uniforms is None and opts.draw_in_kernel:
                    philox = _philox_draw(draw_rows * tgt_len * src_len, x.device, generator, row0 * tgt_len * src_len)
                if philox is None:
                    mask_u = _draw_uniforms((draw_rows, tgt_len, src_len), x.device, uniforms, generator)
                    if draw_rows != batch_size:
                        mask_u = mask_u[row0:row0 + batch_size].contiguous()
        a = self.attention
        tgt_value = math.log(float(src_len)) * cm.entropy_target if mask_mode == 1 else None        # ref :273
        side: Dict[str, Any] = {}
        prep_key = None
        if not (self.training and torch.is_grad_enabled()):
            # inference / no gradient recording: what the kernels derive from the parameters alone (scaled query projection, folded
            # key matrix, transposes, MFMA-fragment copies: one launch, ~10 us) is kept until a parameter or the query moves --
            # same policy as the cast cache (version counters + storage; invalidate_cast_cache() after writes through .data)
            prep_key = tuple((t._version, t.data_ptr()) for t in (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, q_base)
                             if t is not None) + (x.dtype, x.device, embed_dim, self.num_heads)
            hit = self.__dict__.get("_prep_cache")
            side["prep_cache"] = (hit[1], True) if (hit is not None and hit[0] == prep_key) else (None, False)
        y, attn_w, masked_w, entropy, mask_rate, tgt_entropy = _PoolFunction.apply(
            x, q_base, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, kpm, mask_u,
            self.num_heads, mask_mode, 1 if cm is None else int(cm.min_active),
            0.15 if cm is None else float(cm.base_mask_prob), 0.7 if cm is None else float(cm.entropy_target), 1e-8,
            False, tgt_value, self._activation_dtype_params(x.dtype, q_base) if a.in_proj_weight.dtype != x.dtype else None, side,
            philox, opts)
        if prep_key is not None and "prep_cache" in side and side["prep_cache"][0] is not None:
            self._prep_cache = (prep_key, side["prep_cache"][0])

        dt = x.dtype
        attn_output = y.unsqueeze(1) if self.batch_first else y.unsqueeze(0)          # [B,1,E] / [1,B,E]
        attn_weights = attn_w.to(dt).unsqueeze(1)                                    # [B,1,M] (always batch-major)

        info: Dict[str, Any] = {}
        if cm is not None:                                                           # ref :526-541
            if cm.training and src_len <= 1:
                z = torch.zeros(batch_size, tgt_len, device=x.device, dtype=dt)
                mask_info = {'entropy': z, 'mask_rate': z.clone(), 'target_entropy': z.clone()}
                masked_weights = attn_weights
            elif cm.training:
                cm._last_seq_len = src_len                                           # ref :187
                ent = entropy.to(dt).unsqueeze(1)
                if "ent_partial" in side and ent.dtype == entropy.dtype:
                    # this very tensor object carries the regulariser's partial sums (entropy_loss looks for them)
                    ent._aecf_entropy_partials = (side["ent_partial"], ent._version)
                mask_info = {
                    'entropy': ent,
                    'mask_rate': mask_rate.unsqueeze(1),                             # float32 (ref :275)
                    'target_entropy': tgt_entropy.unsqueeze(1),
                }
                masked_weights = masked_w.to(dt).unsqueeze(1)
            else:
                mask_info = {'entropy': entropy.to(dt).unsqueeze(1), 'mask_rate': mask_rate.to(dt).unsqueeze(1)}
                masked_weights = attn_weights
            info.update(mask_info)
            info['attention_weights'] = attn_weights
            if return_info:
                info['masked_attention_weights'] = masked_weights.detach()
        elif return_info:
            info['attention_weights'] = attn_weights
        if return_info:
            return attn_output, info
        return attn_output

    def _forward_empty(self, query, key, tgt_len, src_len, return_info):
        """Empty batch (PROBED on the reference: empty outputs with the usual keys, float dtype of the input): no kernel
        is launched; the outputs stay attached to the parameters so that a backward yields zero gradients."""
        dt, dev, E = key.dtype, key.device, self.embed_dim
        a = self.attention
        zero = (a.in_proj_weight.sum() + a.out_proj.weight.sum() + query.sum() + key.sum()).to(dt) * 0
        if a.in_proj_bias is not None:
            zero = zero + (a.in_proj_bias.sum() + a.out_proj.bias.sum()).to(dt) * 0
        shape = (0, tgt_len, E) if self.batch_first else (tgt_len, 0, E)
        attn_output = torch.zeros(shape, dtype=dt, device=dev) + zero
        weights = torch.zeros(0, tgt_len, src_len, dtype=dt, device=dev) + zero
        info: Dict[str, Any] = {}
        cm = self.curriculum_masking
        if cm is not None:
            z = torch.zeros(0, tgt_len, dtype=dt, device=dev)
            info.update({'entropy': z, 'mask_rate': z.to(torch.float32) if cm.training else z.clone()})
            if cm.training:
                info['target_entropy'] = z.clone()
            info['attention_weights'] = weights
            if return_info:
                info['masked_attention_weights'] = weights.detach()
        elif return_info:
            info['attention_weights'] = weights
        return (attn_output, info) if return_info else attn_output

    def _forward_general(self, q_bf, k_bf, v_bf, key_padding_mask, attn_mask, return_info, batch_size, tgt_len,
                         src_len, uniforms=None, generator=None, pad_to=None):
        """nn.MultiheadAttention semantics for per-sample queries / tgt_len > 1 / key != value / attn_mask / dropout
        (ref :503-521 -> torch functional.py:5836-5852, 6504-6612), then the curriculum hook exactly as the
        reference applies it to the pooled weights (ref :526-541)."""
        dev, dt, H = k_bf.device, k_bf.dtype, self.num_heads
        add_mask, stride = None, 0
        if attn_mask is not None:                                              # torch functional.py:6237-6264
            if attn_mask.dim() == 2:
                if tuple(attn_mask.shape) != (tgt_len, src_len):
                    raise RuntimeError(f"The shape of the 2D attn_mask is {attn_mask.shape}, but should be "
                                       f"{(tgt_len, src_len)}.")
            elif attn_mask.dim() == 3:
                if tuple(attn_mask.shape) != (batch_size * H, tgt_len, src_len):
                    raise RuntimeError(f"The shape of the 3D attn_mask is {attn_mask.shape}, but should be "
                                       f"{(batch_size * H, tgt_len, src_len)}.")
                stride = tgt_len * src_len
            else:
                raise RuntimeError(f"attn_mask's dimension {attn_mask.dim()} is not supported")
            am = attn_mask.to(dev)
            if am.dtype == torch.bool:
                add_mask = torch.zeros(am.shape, dtype=torch.float32, device=dev).masked_fill_(am, float("-inf"))
            else:
                add_mask = am.to(torch.float32)
            add_mask = add_mask.contiguous()
        kpm = None
        if key_padding_mask is not None:
            if tuple(key_padding_mask.shape) != (batch_size, src_len):
                raise RuntimeError(f"key_padding_mask shape {tuple(key_padding_mask.shape)} != {(batch_size, src_len)}")
            if key_padding_mask.dtype == torch.bool or not key_padding_mask.is_floating_point():
                kpm = key_padding_mask.to(device=dev, dtype=torch.uint8).contiguous()
            else:                                                              # float masks are additive in torch
                extra = key_padding_mask.to(dev, torch.float32).view(batch_size, 1, 1, src_len)
                base = 0.0 if add_mask is None else (add_mask.view(batch_size, H, tgt_len, src_len) if stride
                                                     else add_mask.view(1, 1, tgt_len, src_len))
                add_mask = (base + extra).expand(batch_size, H, tgt_len, src_len).reshape(
                    batch_size * H, tgt_len, src_len).contiguous()
                stride = tgt_len * src_len
        drop_p = float(self.attention.dropout) if (self.attention.dropout > 0.0 and self.training) else 0.0
        drop_u = torch.rand(batch_size * H, tgt_len, src_len, device=dev) if drop_p > 0.0 else None
        a = self.attention
        if pad_to is None:
            y, attn_w = _MhaFunction.apply(q_bf.to(dt), k_bf, v_bf, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight,
                                           a.out_proj.bias, add_mask, stride, kpm, drop_u, drop_p, H)
        else:
            E = self.embed_dim
            w_in2, b_in2, w_out2, b_out2 = _pad_heads(a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                                      E, H, pad_to)
            wide = lambda t_: torch.nn.functional.pad(t_.to(dt), (0, pad_to[0] - E))
            y, attn_w = _MhaFunction.apply(wide(q_bf), wide(k_bf), wide(v_bf), w_in2, b_in2, w_out2, b_out2, add_mask, stride,
                                           kpm, drop_u, drop_p, H)
            y = y[..., :E]
        attn_output = y if self.batch_first else y.transpose(0, 1)
        attn_weights = attn_w.to(dt)                                           # [B,T,S], always batch-major

        info: Dict[str, Any] = {}
        cm = self.curriculum_masking
        if cm is not None:                                                     # ref :526-541
            masked_weights, mask_info = cm(attn_weights, uniforms=uniforms, generator=generator)
            info.update(mask_info)
            info['attention_weights'] = attn_weights
            if return_info:
                info['masked_attention_weights'] = masked_weights.detach()
        elif return_info:
            info['attention_weights'] = attn_weights
        if return_info:
            return attn_output, info
        return attn_output

    def extra_repr(self) -> str:                                              # ref :549-552
        return (f'embed_dim={self.embed_dim}, num_heads={self.num_heads}, '
                f'batch_first={self.batch_first}, '
                f'curriculum_masking={self.curriculum_masking is not None}')


# ----------------------------------------------------------------------------------------------
# ref: aecf/AECFLayer.py:556-652
# ----------------------------------------------------------------------------------------------
def _scaled_dot_product_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                                  scale: Optional[float] = None) -> torch.Tensor:
    """softmax(Q K^T * scale) V without projections (ref :556-581), HIP kernel."""
    _require_device(query, "query")
    if query.dtype not in _DTYPES:
        raise NotImplementedError(f"aecf_amd: dtype {query.dtype} is not supported (bfloat16 / float32 only)")
    if scale is None:
        scale = query.size(-1) ** -0.5
    return _SdpaFunction.apply(query, key, value, float(scale))


def multimodal_attention_pool(query: torch.Tensor, key: torch.Tensor, value: Optional[torch.Tensor] = None,
                              embed_dim: Optional[int] = None, num_heads: int = 1, dropout: float = 0.0,
                              curriculum_masking: Optional[CurriculumMasking] = None,
                              training: bool = False) -> torch.Tensor:
    """Functional interface (ref :584-652): projection-free fast path, else a fresh randomly
    initialised MultimodalAttentionPool per call (initialised on the CPU generator exactly like the
    reference, then moved to the query's device)."""
    if embed_dim is None:
        embed_dim = query.size(-1)
    if value is None:
        value = key
    if (not training and curriculum_masking is None and dropout == 0.0 and num_heads == 1):   # ref :638-640
        return _scaled_dot_product_attention(query, key, value)
    pool = MultimodalAttentionPool(embed_dim=embed_dim, num_heads=num_heads, dropout=dropout,
                                   curriculum_masking=curriculum_masking, batch_first=True)
    pool = pool.to(device=query.device)
    pool.train(training)
    return pool(query, key, value)


# ----------------------------------------------------------------------------------------------
# ref: aecf/AECFLayer.py:655-728
# ----------------------------------------------------------------------------------------------
def create_fusion_pool(embed_dim: int, num_modalities: int, mask_prob: float = 0.15,
                       **kwargs) -> Tuple[nn.Parameter, MultimodalAttentionPool]:
    """Factory (ref :655-728): ``(fusion_query ~ N(0, 2/E) of shape [1,1,E], pool with curriculum masking)``."""
    if not isinstance(embed_dim, int) or embed_dim <= 0:                      # ref :706-711
        raise ValueError(f"embed_dim must be a positive integer, got {embed_dim}")
    if not isinstance(num_modalities, int) or num_modalities <= 0:
        raise ValueError(f"num_modalities must be a positive integer, got {num_modalities}")
    if not isinstance(mask_prob, (int, float)) or not (0.0 < mask_prob <= 1.0):
        raise ValueError(f"mask_prob must be in (0, 1], got {mask_prob}")
    fusion_query = nn.Parameter(torch.empty(1, 1, embed_dim))                 # ref :714-716
    nn.init.normal_(fusion_query, 0.0, (2.0 / embed_dim) ** 0.5)
    curriculum_masking = CurriculumMasking(base_mask_prob=mask_prob)          # ref :719
    attention_pool = MultimodalAttentionPool(embed_dim=embed_dim, curriculum_masking=curriculum_masking, **kwargs)
    return fusion_query, attention_pool
