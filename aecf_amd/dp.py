"""Data-parallel plumbing for the fusion path: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-process (no collective anywhere in it), so this layer has no reference
counterpart; it is what BASELINE.json's north_star asks for around the fused kernels:

* samples are independent in the forward/backward kernels -> each rank runs a contiguous batch shard,
  parameters and the fusion query are replicated;
* the only exchange steps are (1) ONE all-reduce of all parameter gradients per step, flat-bucketed so
  the 4E^2+5E values travel as a single collective (a few MB: latency-bound on xGMI, so one big
  message, not one per tensor), (2) an all-gather of fused embeddings for cross-batch contrastive
  negatives whose backward is a reduce-scatter of the gradient, (3) scalar means of the logged statistics.
* mask RNG: every rank seeds its device generator alike and names its rows of the global batch
  (``pool(..., batch_shard=(first_row, global_batch))``): the statistics kernel evaluates the rank's elements of ONE global
  draw, so N-rank masks equal 1-rank masks bit for bit with no uniforms tensor and no launch (a recorded draw still goes
  through the public ``uniforms=`` argument: ``global_uniforms`` + ``shard_batch``).
* replicas are made identical by ``broadcast_parameters``; gradient averaging is unweighted, uneven shards scale their
  local loss by ``shard_loss_scale``.
* ``attach(pool)`` makes a pool module data-parallel: its backward folds 1 / world into the parameter gradients as it stores
  them (the average is then ONE sum all-reduce, no divide launch) and keeps float32 sums for bf16 parameters.  Per module,
  explicit -- nothing here flips a process-wide switch.
"""
from __future__ import annotations

import weakref
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_bounds(global_batch: int, rank: int, world: int):
    """Contiguous, balanced shard [lo, hi) of the batch for this rank (first ranks take the remainder)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def global_uniforms(global_batch: int, tgt_len: int, modalities: int, seed: int, device) -> torch.Tensor:
    """The [B_global, tgt, M] float32 uniforms every rank agrees on (same seed -> same tensor); a rank
    consumes ``shard_batch(u, rank, world)``.  B_global*M floats: negligible next to the activations."""
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.rand(global_batch, tgt_len, modalities, dtype=torch.float32, device=device, generator=g)


class FlatGradBucket:
    """All parameter gradients of the replicated model in ONE flat buffer.

    ``p.grad`` of every parameter is a view into ``self.flat`` (per dtype), so backward accumulates
    straight into the bucket and ``all_reduce()`` is a single in-place collective -- no per-tensor copies.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBucket needs at least one parameter that requires grad")
        self.buffers = {}
        by_dtype = {}
        for p in self.params:
            by_dtype.setdefault((p.dtype, p.device), []).append(p)
        for (dtype, device), ps in by_dtype.items():
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=dtype, device=device)
            off = 0
            for p in ps:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self.buffers[(dtype, device)] = flat

    @property
    def numel(self) -> int:
        return sum(b.numel() for b in self.buffers.values())

    def zero(self) -> None:
        for b in self.buffers.values():
            b.zero_()

    def all_reduce(self, group=None, average: bool = True, async_op: bool = False):
        """Sum (and optionally average) the bucket across ranks, in place."""
        _, world = world_info(group)
        if world == 1:
            return []
        works = []
        for b in self.buffers.values():
            if average:
                b.div_(world)           # pre-divide: the sum of bf16 buckets then stays in range
            works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=async_op))
        return works


# ReduceOp.AVG support of the backend, decided ONCE per (backend, dtype) by probe_avg_support -- a tiny all-reduce whose
# outcome all ranks agree on -- never by catching an exception inside the step (RCCL errors surface asynchronously; ranks
# that disagreed would issue different collectives and hang).  Unprobed = sum + divide, which every backend takes.
_avg_support = {}


def probe_avg_support(dtype: torch.dtype, device, group=None) -> bool:
    """Collective: every rank calls it with the same arguments (at start-up).  Tries ReduceOp.AVG on 1 element, then
    agrees on the outcome with a MIN all-reduce, so either all ranks use AVG from now on or none does."""
    _, world = world_info(group)
    key = (dist.get_backend(group) if world > 1 else "none", dtype)
    if key in _avg_support:
        return _avg_support[key]
    ok = 0
    if world > 1 and key[0] == "nccl":
        try:
            t = torch.ones(1, dtype=dtype, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            ok = 1 if abs(float(t.item()) - 1.0) < 1e-3 else 0
        except (RuntimeError, ValueError):
            ok = 0
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = int(flag.item())
    _avg_support[key] = bool(ok)
    return bool(ok)


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0, group=None) -> None:
    """Replicas start identical: every parameter (and buffer) of rank ``src`` overwrites the other ranks' (one flat
    broadcast per dtype).  Without it, replicas built under different RNG states would silently diverge."""
    _, world = world_info(group)
    if world == 1:
        return
    by_dtype = {}
    for p in params:
        by_dtype.setdefault(p.dtype, []).append(p)
    with torch.no_grad():
        for ps in by_dtype.values():
            flat = torch.cat([p.detach().reshape(-1) for p in ps])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for p in ps:
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()


# pool modules made data-parallel by attach(): parameter id -> weak reference to the module (all_reduce_grads / GradOverlap find
# the module's DpState through the parameters they are handed; a module that is gone drops out by itself)
_attached = {}


def attach(pool, group=None, average: bool = True, keep_f32: bool = True, defer_rounding: bool = False, world: Optional[int] = None):
    """Make ONE ``MultimodalAttentionPool`` data-parallel (explicit, per module).  Its fused backward then
    (a) multiplies the five parameter gradients by 1 / world as it stores them (``average``; aecf_pool_bwd_args.grad_scale),
    so ``all_reduce_grads`` / ``GradOverlap`` issue ONE sum collective with no divide launch around it, and
    (b) with bf16 parameters (``keep_f32``) writes its float32 batch sums: the collective moves and adds THOSE and the bf16
    gradient is rounded ONCE from the mean -- an N-rank gradient then differs from the one-rank gradient by float32 summation
    order only.  ``defer_rounding``: the bf16 ``p.grad`` tensors stay UNINITIALISED between the backward and the collective
    (no cast launch; the caller promises to call ``all_reduce_grads`` / ``GradOverlap.finish`` before reading them).
    Until the collective has run, ``p.grad`` holds this rank's gradient divided by world.  ``world`` overrides the group's
    size (rehearsals).  ``detach(pool)`` undoes it.  Returns the module's ``layer.DpState``."""
    from . import layer
    if world is None:
        _, world = world_info(group)
    st = layer.DpState(world, (1.0 / world) if (average and world > 1) else 1.0, keep_f32, defer_rounding)
    pool._options().dp = st
    ref = weakref.ref(pool)
    for p in pool.parameters():
        _attached[id(p)] = ref
        st.add_scaled(p)
    return st


def detach(pool) -> None:
    pool._options().dp = None
    for p in pool.parameters():
        _attached.pop(id(p), None)


def _states_of(params) -> list:
    """DpStates of the attached pool modules that own any of ``params`` (each once)."""
    out, seen = [], set()
    for p in params:
        ref = _attached.get(id(p))
        pool = None if ref is None else ref()
        if pool is None:
            continue
        st = pool._options().dp
        if st is not None and id(st) not in seen:
            seen.add(id(st))
            out.append(st)
    return out


def _all_states() -> list:
    out, seen = [], set()
    for key, ref in list(_attached.items()):
        pool = ref()
        if pool is None:
            _attached.pop(key, None)
            continue
        st = pool._options().dp
        if st is not None and id(st) not in seen:
            seen.add(id(st))
            out.append(st)
    return out


def shard_loss_scale(b_local: int, b_global: int, world: int) -> float:
    """Factor for a rank's mean-over-its-shard loss so that the AVERAGE of the ranks' gradients equals the gradient of
    the mean loss over the global batch: ``world * b_local / b_global`` (1.0 for even shards).  shard_bounds hands the
    first ranks one extra row when world does not divide the batch; all_reduce_grads / FlatGradBucket average without
    weights, so uneven shards need this factor on the local loss (or an upstream gradient scaled by it)."""
    return float(world) * float(b_local) / float(b_global)


def flat_grad_alias(params: Iterable[torch.nn.Parameter]) -> Optional[torch.Tensor]:
    """The fusion layer's backward writes its five parameter gradients into ONE allocation (aecf_amd/layer.py:
    _PoolFunction.backward) and autograd keeps those tensors as ``p.grad`` without copying.  If the gradients of
    ``params`` tile one contiguous run of a single storage, return that run as a flat tensor (so the whole set
    travels as one in-place collective with no zero / accumulate / copy kernels around it); otherwise None."""
    grads = [p.grad for p in params if p.requires_grad]
    if not grads or any(g is None for g in grads):
        return None
    g0 = grads[0]
    st = g0.untyped_storage()
    for g in grads:
        if g.untyped_storage().data_ptr() != st.data_ptr() or g.dtype != g0.dtype or not g.is_contiguous():
            return None
    spans = sorted((g.storage_offset(), g.numel()) for g in grads)
    off = spans[0][0]
    for o, n in spans:
        if o != off:                                  # a gap or an overlap: not one run
            return None
        off += n
    return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(st, spans[0][0], (off - spans[0][0],))


def all_reduce_grads(params: Iterable[torch.nn.Parameter], group=None, average: bool = True, fp32: Optional[bool] = None,
                     rehearse: bool = False):
    """One collective for the gradients of ``params``: in place over their shared allocation when they alias one
    (see flat_grad_alias), else through a temporary flat copy.  Averages inside the collective (ReduceOp.AVG) when
    probe_avg_support found the backend takes it, else divides and sums.  The average is unweighted: with uneven
    shards scale the local loss by shard_loss_scale.  ``fp32`` (default None = True whenever a gradient is bf16 / fp16):
    reduced-precision gradients travel and are summed as float32 and are rounded ONCE after the collective -- a ring sum in
    bf16 rounds at every hop, eight ranks' worth of it is the difference between 2e-2 and the single rounding the one-rank
    step has (each rank's own rounding of its float32 batch sums to its bf16 ``p.grad`` has already happened and stays; at
    4 E^2 + 5 E elements the float32 transport is 4 MB at d = 512, a latency-bound collective either way).  ``fp32=False``
    keeps the gradients' own dtype on the wire.  ``rehearse``: issue the collective even on a ONE-rank group (bench.py
    --force-dp: the RCCL call, and its capture into a HIP graph, on a one-GPU box)."""
    params = [p for p in params if p.requires_grad and p.grad is not None]
    _, world = world_info(group)
    if (world == 1 and not (rehearse and dist.is_initialized())) or not params:
        return
    # parameters whose gradients an attached pool's backward already multiplied by 1 / world (the module's own and the leaf
    # fusion queries it has seen: layer.DpState.scaled) go through their own collective, without the divide
    states = [st for st in _all_states() if st.grad_scale != 1.0]
    pre = [p for p in params if any(st.is_scaled(p) for st in states)]
    if pre and len(pre) < len(params):
        taken = {id(p) for p in pre}
        all_reduce_grads(pre, group, average, fp32, rehearse)
        all_reduce_grads([p for p in params if id(p) not in taken], group, average, fp32, rehearse)
        return
    prescaled = bool(pre)
    flat = flat_grad_alias(params)
    reduced = any(p.grad.dtype in (torch.bfloat16, torch.float16) for p in params)
    if fp32 is None:
        fp32 = reduced
    # the float32 sums the backward kept behind this very run (bf16 parameters, dp.attach(keep_f32=True))
    wide = None
    for st in _all_states():
        w_ = st.take(flat)
        if w_ is not None:
            wide = w_
    if prescaled and not average:
        raise RuntimeError("all_reduce_grads(average=False) on gradients an attached pool already divided by world")
    divide = average and not prescaled
    if fp32 and reduced:
        if wide is None:
            wide = flat.float() if flat is not None else torch.cat([p.grad.reshape(-1).float() for p in params])
        if divide:
            wide.div_(world)                          # (before the sum, as GradOverlap does: the two paths stay bit-equal)
        dist.all_reduce(wide, op=dist.ReduceOp.SUM, group=group)
        if flat is not None:
            flat.copy_(wide)                          # one rounding, in place over the allocation autograd holds
            return
        off = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(wide[off:off + n].view_as(p.grad))
            off += n
        return
    copied = flat is None
    if copied:
        flat = torch.cat([p.grad.reshape(-1).to(params[0].grad.dtype) for p in params])
    if divide and _avg_support.get((dist.get_backend(group), flat.dtype), False):
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    else:
        if divide:
            flat.div_(world)                          # pre-divide: the sum of bf16 values then stays in range
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if copied:
        off = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


class GradOverlap:
    """EXPERIMENTAL (never run over RCCL on real multi-GPU hardware by this build: only the gloo rehearsal on one GPU).
    Runs the all-reduce of the fusion layer's parameter gradients BEHIND the input-gradient kernel of its backward.

    With this object installed (on the ``DpState`` of the attached pool modules: ``dp.attach``) the backward computes dx last and the library announces
    the moment the five parameter gradients are final with a HIP event (aecf_pool_bwd_args.param_grads_event); a side
    stream waits for that event and issues ONE collective there, in place on the gradient allocation, while the dx
    kernel runs on the caller's stream.  ``finish`` makes the current stream wait for it and reduces, the plain way, any
    parameters it did not cover.  A second fused backward inside one region (two applications of the pool feeding one loss)
    is reduced too, but on the main stream and after the first collective has completed: autograd adds its gradients into
    the first one's allocation, which nothing may be mutating meanwhile.  Use:

        overlap = dp.GradOverlap(params=params)
        with overlap:                 # installs / removes the hook
            loss.backward()
            overlap.finish(params)    # instead of dp.all_reduce_grads(params)

    The in-place collective is only sound when autograd KEEPS the backward's tensors as ``p.grad`` (it then never reads them
    on the main stream while the side stream mutates them).  Two cases break that and are refused: gradients that arrive as
    cast copies (parameter dtype != gradient dtype: the layer does not call the hook) and parameters that already hold a
    gradient (accumulation adds into it on the main stream): pass ``params`` and the hook is not installed for a backward
    that starts with any ``p.grad`` set -- ``finish`` then falls back to the plain all-reduce.

    The collective is ~2 MB and latency-bound on xGMI; dx is ~13 % of the step, which is what it can hide behind."""

    def __init__(self, group=None, average: bool = True, params: Optional[Iterable[torch.nn.Parameter]] = None):
        self.group, self.average = group, average
        self.params = None if params is None else list(params)
        self.stream = None
        self.pending = []                                   # (flat, work) of every collective issued since the last finish
        self._hooked = []                                   # DpStates this region installed itself on
        self._fired = []                                    # ... and those whose backward has called the hook since the last finish

    def __enter__(self):
        _, world = world_info(self.group)
        accumulating = self.params is not None and any(p.grad is not None for p in self.params)
        self._hooked = []
        if world > 1 and not accumulating:
            # the hook lives on the attached pool modules (dp.attach) whose parameters this region covers -- all attached
            # modules when no parameter list was given; a pool that was never attached is reduced by finish() the plain way
            for st in (_states_of(self.params) if self.params is not None else _all_states()):
                st.hook = self
                self._hooked.append(st)
        return self

    def __exit__(self, *exc):
        for st in self._hooked:
            st.hook = None
        self._hooked = []
        return False

    def _reduce(self, flat, async_op, divide=True):
        _, world = world_info(self.group)
        divide = divide and self.average
        if divide and _avg_support.get((dist.get_backend(self.group), flat.dtype), False):
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        if divide:
            flat.div_(world)
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def __call__(self, flat: torch.Tensor, event, flat32: Optional[torch.Tensor] = None, state=None) -> None:
        # reduced-precision gradients whose float32 sums the backward kept: the collective runs on THOSE and finish() rounds the
        # mean into `flat` (the allocation autograd holds, left uninitialised by the backward) once
        prescaled = state is not None and state.grad_scale != 1.0
        if state is not None:
            if state.runs:
                state.runs.pop()                             # this call's record: consumed here, not by all_reduce_grads
            if not any(st is state for st in self._fired):
                self._fired.append(state)
        low = None
        if flat32 is not None:
            low, flat = flat, flat32
        if self.pending:
            # A SECOND fused backward inside one region (two pool applications, or two backward() calls): autograd is about
            # to ADD this call's gradients into the p.grad the first call left -- on the main stream, reading both allocations.
            # Nothing may mutate them behind its back: the first collective is waited for here and this one runs on the main
            # stream (no overlap for it); the sum autograd forms is then the sum of two reduced gradients.
            for low_, work, wide_ in self.pending:
                if work is not None:
                    work.wait()
                if wide_ is not None:                        # the first collective's float32 mean -> the allocation autograd adds into
                    torch.cuda.current_stream().wait_stream(self.stream)
                    low_.copy_(wide_)
            self.pending = [(a_, None, None) for a_, _, _ in self.pending]
            torch.cuda.current_stream().wait_stream(self.stream)
            self._reduce(flat, async_op=False, divide=not prescaled)
            if low is not None:
                low.copy_(flat)
            self.pending.append((flat if low is None else low, None, None))
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=flat.device)
        self.stream.wait_event(event)                       # the gradients are final once the library's event has fired
        with torch.cuda.stream(self.stream):
            work = self._reduce(flat, async_op=True, divide=not prescaled)
        flat.record_stream(self.stream)
        if low is not None:
            low.record_stream(self.stream)
        self.pending.append((flat if low is None else low, work, None if low is None else flat))

    def finish(self, params: Iterable[torch.nn.Parameter]) -> None:
        """Wait for the collectives issued behind dx; all-reduce whatever they did not cover."""
        params = [p for p in params if p.requires_grad and p.grad is not None]
        _, world = world_info(self.group)
        pending, self.pending = self.pending, []
        if world == 1 or not params:
            self._fired = []
            return
        if not pending:                                     # the hook never fired (no fused backward ran): plain path
            all_reduce_grads(params, self.group, self.average)
            return
        fired, self._fired = self._fired, []
        for flat, work, wide in pending:
            if work is not None:
                work.wait()
        torch.cuda.current_stream().wait_stream(self.stream)
        for flat, _, wide in pending:
            if wide is not None:
                flat.copy_(wide)                             # ONE rounding of the float32 mean, in place
        # covered = the gradients of the modules whose backward called the hook (their parameters and the leaf queries they saw),
        # WHEREVER autograd has put them since: two pool applications in one backward are summed into fresh allocations, and a
        # sum of reduced gradients must not be reduced again
        rest = [p for p in params if not any(st.is_scaled(p) for st in fired)]
        if rest:
            all_reduce_grads(rest, self.group, self.average)


class _AllGatherRows(torch.autograd.Function):
    """z_local [b, d] -> z_all [sum b, d] (rank order); backward: each rank keeps the gradient rows of its own
    shard summed over ranks (reduce-scatter; all-reduce + slice where the backend has no reduce_scatter)."""

    @staticmethod
    def forward(ctx, z, group, sizes=None):
        rank, world = world_info(group)
        ctx.group, ctx.rank, ctx.world = group, rank, world
        if world == 1:
            ctx.sizes = [z.shape[0]]
            return z.clone()
        if sizes is not None:                               # the caller knows every rank's row count: no exchange, no host sync
            ctx.sizes = [int(v) for v in sizes]
            if len(ctx.sizes) != world or ctx.sizes[rank] != z.shape[0]:
                raise ValueError(f"all_gather_rows: sizes {ctx.sizes} do not describe {world} ranks with {z.shape[0]} rows here")
        else:
            n = torch.tensor([z.shape[0]], device=z.device, dtype=torch.int64)
            got = [torch.zeros_like(n) for _ in range(world)]
            dist.all_gather(got, n, group=group)
            ctx.sizes = [int(v.item()) for v in got]
        zc = z.contiguous()
        if len(set(ctx.sizes)) == 1:
            out = torch.empty(world * zc.shape[0], *zc.shape[1:], dtype=z.dtype, device=z.device)
            dist.all_gather_into_tensor(out, zc, group=group)
            return out
        parts = [torch.empty(s, *zc.shape[1:], dtype=z.dtype, device=z.device) for s in ctx.sizes]
        dist.all_gather(parts, zc, group=group)
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, dz_all):
        if ctx.world == 1:
            return dz_all, None, None
        lo = sum(ctx.sizes[:ctx.rank])
        hi = lo + ctx.sizes[ctx.rank]
        dz_all = dz_all.contiguous()
        even = len(set(ctx.sizes)) == 1
        if even and dist.get_backend(ctx.group) == "nccl":
            out = torch.empty(ctx.sizes[ctx.rank], *dz_all.shape[1:], dtype=dz_all.dtype, device=dz_all.device)
            dist.reduce_scatter_tensor(out, dz_all, op=dist.ReduceOp.SUM, group=ctx.group)
            return out, None, None
        dist.all_reduce(dz_all, op=dist.ReduceOp.SUM, group=ctx.group)
        return dz_all[lo:hi].clone(), None, None


def all_gather_rows(z: torch.Tensor, group=None, sizes=None) -> torch.Tensor:
    """Autograd-aware all-gather of fused embeddings (cross-batch contrastive negatives).  ``sizes`` (every rank's row count,
    rank order) skips the small exchange + host read that otherwise finds them out on every call."""
    return _AllGatherRows.apply(z, group, sizes)


def all_reduce_mean_scalar(v: torch.Tensor, weight: float = 1.0, group=None) -> torch.Tensor:
    """Weighted mean of a scalar statistic over ranks (weight = local sample count)."""
    _, world = world_info(group)
    if world == 1:
        return v.detach().clone()
    buf = torch.stack([v.detach().float() * weight, torch.tensor(float(weight), device=v.device)])
    dist.all_reduce(buf, group=group)
    return buf[0] / buf[1]
