"""AdamW of the example trainer as one launch (``aecf_adamw_step``; ref xrays/train_xrays_example.py:322-323, 376 uses
``torch.optim.AdamW(lr=1e-4, weight_decay=0.01)``).  Same update rule and the same state layout as torch's (per parameter:
``step`` -- a float32 scalar on the device --, ``exp_avg``, ``exp_avg_sq``), so state dicts move between the two; the step
counters advance on the device, which makes ``step()`` capturable into a HIP graph without further flags.  One limit under
capture: the hyper-parameters (``lr`` included) are kernel ARGUMENTS, so a captured step replays with the values it was captured
with -- a learning-rate schedule needs a re-capture (or the eager step) when the rate changes."""
from __future__ import annotations

import ctypes
from typing import Iterable

import torch

from . import _lib
from .layer import _stream


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or not 0 < betas[0] < 1 or not 0 < betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tickets = {}

    def _ticket(self, device, group_index, n):
        """Ticket words of one parameter group's launch, kept for the optimizer's lifetime: a buffer that has been handed to a
        launch is NEVER freed or replaced (a captured graph replays with its address), a group that grows gets a new, larger one
        next to it."""
        need = 65 * ((n + 23) // 24)                              # AECF_ADAMW_TICKET_WORDS per launch group
        held = self._tickets.setdefault((device, group_index), [])
        for t in held:
            if t.numel() >= need:
                return t
        t = torch.zeros(max(need, 65 * 8), dtype=torch.int32, device=device)
        held.append(t)
        return t

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        vp = ctypes.c_void_p
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                # (a state dict loaded from torch.optim.AdamW can carry these; the kernel implements neither)
                raise RuntimeError("FusedAdamW: amsgrad / maximize are not implemented by aecf_adamw_step")
            ps = [p for p in group["params"] if p.grad is not None and p.numel() > 0]    # (torch skips empty tensors too)
            if not ps:
                continue
            dev = ps[0].device
            for p in ps:
                if p.device.type != "cuda" or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("FusedAdamW: float32 parameters with dense float32 gradients on a ROCm device only "
                                       "(no CPU fallback is provided)")
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdamW: parameters must be contiguous")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                else:
                    # a state loaded from torch.optim.AdamW keeps `step` on the host (or as a Python number) unless that optimizer
                    # was capturable: the kernel reads and advances it on the device
                    step = st["step"]
                    if not torch.is_tensor(step) or step.device != p.device or step.dtype != torch.float32:
                        st["step"] = torch.as_tensor(float(step), dtype=torch.float32, device=p.device)
                    for key in ("exp_avg", "exp_avg_sq"):
                        if st[key].device != p.device or st[key].dtype != torch.float32 or not st[key].is_contiguous():
                            st[key] = st[key].to(device=p.device, dtype=torch.float32).contiguous()
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            arr = lambda ts: (vp * len(ts))(*[t.data_ptr() for t in ts])
            b1, b2 = group["betas"]
            _lib.check(lib.aecf_adamw_step(
                len(ps), arr(ps), arr(grads), arr([self.state[p]["exp_avg"] for p in ps]),
                arr([self.state[p]["exp_avg_sq"] for p in ps]), arr([self.state[p]["step"] for p in ps]),
                (ctypes.c_int64 * len(ps))(*[p.numel() for p in ps]), self._ticket(dev, gi, len(ps)).data_ptr(),
                float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), _stream()),
                "aecf_adamw_step")
        return loss
