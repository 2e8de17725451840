"""The C ABI's memory contract (include/aecf_hip.h: "the CALLER owns every buffer ... the library never allocates"): every
output and both workspaces of aecf_pool_forward / aecf_pool_backward are handed over with guard bands in front of and behind
them, sized EXACTLY as the header and the *_workspace_bytes entry points say; after a forward + backward the payloads are
written and not one guard byte has changed.  Shapes: the headline kernels (d = 512, the hi + lo gradient products included),
the flat-row form (d = 1024, M = 4), d = 768, a float32 shape, batches that end in ragged tiles and ragged batch splits."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 4096
PATTERN = 0xA7


class Guarded:
    """Device buffers carved out of one allocation each, with GUARD bytes of PATTERN on both sides."""

    def __init__(self, dev):
        self.dev = dev
        self.bufs = []

    def new(self, nbytes, fill=None):
        nbytes = int(nbytes)
        pad = (-nbytes) % 256
        raw = torch.full((GUARD + nbytes + pad + GUARD,), PATTERN, dtype=torch.uint8, device=self.dev)
        body = raw[GUARD:GUARD + nbytes]
        if fill is not None:
            body.fill_(fill)
        self.bufs.append((raw, nbytes, pad))
        return body

    def tensor(self, shape, dtype, fill=None):
        n = 1
        for s in shape:
            n *= int(s)
        body = self.new(n * torch.empty((), dtype=dtype).element_size(), fill)
        return body.view(dtype).view(*shape)

    def check(self):
        for i, (raw, nbytes, pad) in enumerate(self.bufs):
            head = raw[:GUARD]
            tail = raw[GUARD + nbytes + pad:]
            assert bool((head == PATTERN).all()), f"buffer {i} ({nbytes} bytes): bytes in FRONT of it were written"
            assert bool((tail == PATTERN).all()), f"buffer {i} ({nbytes} bytes): bytes BEHIND it were written"
            if pad:                                              # (the alignment slack is not the library's either)
                assert bool((raw[GUARD + nbytes:GUARD + nbytes + pad] == PATTERN).all()), f"buffer {i}: slack written"


CASES = [
    # B, M, E, H, dtype, mask_mode, hilo
    (1100, 3, 512, 8, torch.bfloat16, 1, False),
    (1100, 3, 512, 8, torch.bfloat16, 1, True),
    (4133, 3, 512, 8, torch.bfloat16, 2, True),      # several ragged batch splits
    (257, 2, 256, 8, torch.bfloat16, 1, True),       # four head slots per 128 rows
    (333, 1, 256, 4, torch.bfloat16, 0, True),
    (515, 4, 1024, 8, torch.bfloat16, 1, False),
    (700, 2, 768, 8, torch.bfloat16, 1, False),
    (130, 3, 192, 2, torch.float32, 1, False),
    (70, 8, 128, 2, torch.float32, 2, False),
]


@pytest.mark.parametrize("B,M,E,H,dtype,mask_mode,hilo", CASES,
                         ids=[f"B{c[0]}_M{c[1]}_E{c[2]}_H{c[3]}_{'bf16' if c[4] == torch.bfloat16 else 'f32'}_m{c[5]}{'_hilo' if c[6] else ''}"
                              for c in CASES])
def test_pool_calls_stay_inside_the_callers_buffers(B, M, E, H, dtype, mask_mode, hilo):
    from aecf_amd import _lib
    from aecf_amd.layer import _stream
    lib = _lib.load()
    dev = torch.device("cuda:0")
    gd = Guarded(dev)
    dt_code = _lib.AECF_BF16 if dtype == torch.bfloat16 else _lib.AECF_F32
    desc = _lib.PoolDesc(B, M, E, H, dt_code, mask_mode, 1, 0.3, 0.7, 1e-8)
    assert lib.aecf_pool_check(ctypes.byref(desc)) == 0
    hilo_bytes = lib.aecf_pool_hilo_bwd_workspace_bytes(ctypes.byref(desc))
    if hilo and hilo_bytes == 0:
        pytest.skip("AECF_HILO_GRADS is not built for this shape")
    g = torch.Generator(device=dev).manual_seed(B + E)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    x = rnd(B, M, E).to(dtype)
    q = (rnd(E) * 0.3).to(dtype)
    w_in = (rnd(3 * E, E) / E ** 0.5).to(dtype)
    b_in = (rnd(3 * E) * 0.05).to(dtype)
    w_out = (rnd(E, E) / E ** 0.5).to(dtype)
    b_out = (rnd(E) * 0.05).to(dtype)
    dy = rnd(B, E).to(dtype)
    u = torch.rand(B, M, device=dev, generator=g)
    f32 = torch.float32
    nan = 0xFF                                   # payloads start as NaN patterns: "written" is checkable
    y = gd.tensor((B, E), dtype, nan)
    attn_w = gd.tensor((B, M), f32, nan)
    probs = gd.tensor((B, H, M), f32, nan)
    saved_o = gd.tensor((B, E), dtype, nan)
    masked_w = entropy = mask_rate = None
    if mask_mode != 0:
        masked_w, entropy, mask_rate = gd.tensor((B, M), f32, nan), gd.tensor((B,), f32, nan), gd.tensor((B,), f32, nan)
    wants_v = bool(lib.aecf_pool_wants_saved_v(ctypes.byref(desc)))
    saved_v = gd.tensor((B, M, E), dtype, nan) if wants_v else None
    saved_prep = gd.new(lib.aecf_pool_prep_bytes(ctypes.byref(desc)))
    saved_o_lo = gd.tensor((B, E), dtype, nan) if hilo else None
    ent_partial = gd.tensor(((B + 255) // 256,), f32, nan) if (mask_mode == 1 and dtype == torch.bfloat16) else None
    fwd_ws_bytes = lib.aecf_pool_fwd_workspace_bytes(ctypes.byref(desc))
    fwd_ws = gd.new(fwd_ws_bytes)
    p = lambda t_: None if t_ is None else t_.data_ptr()
    flags = _lib.AECF_HILO_GRADS if hilo else 0
    fa = _lib.PoolFwdArgs(p(x), p(q), p(w_in), p(b_in), p(w_out), p(b_out), None, p(u) if mask_mode == 1 else None, p(y), p(attn_w),
                          p(masked_w), p(entropy), p(mask_rate), p(probs), p(saved_o), p(saved_v), p(fwd_ws), fwd_ws_bytes, None,
                          None, None, None, None, p(saved_prep), None, 0.7 * float(torch.log(torch.tensor(float(M)))), flags,
                          p(ent_partial), 0, 0, 0, None, p(saved_o_lo), 0)
    _lib.check(lib.aecf_pool_forward(ctypes.byref(desc), ctypes.byref(fa), _stream()), "aecf_pool_forward")
    torch.cuda.synchronize()
    gd.check()
    assert torch.isfinite(y.float()).all() and torch.isfinite(attn_w).all() and torch.isfinite(probs).all()
    assert torch.isfinite(saved_o.float()).all()
    if hilo:
        assert torch.isfinite(saved_o_lo.float()).all()

    dx = gd.tensor((B, M, E), dtype, nan)
    gdt = f32
    dquery, dw_in, db_in = gd.tensor((E,), gdt, nan), gd.tensor((3 * E, E), gdt, nan), gd.tensor((3 * E,), gdt, nan)
    dw_out, db_out = gd.tensor((E, E), gdt, nan), gd.tensor((E,), gdt, nan)
    bwd_ws_bytes = hilo_bytes if hilo else lib.aecf_pool_bwd_workspace_bytes(ctypes.byref(desc))
    bwd_ws = gd.new(bwd_ws_bytes)
    ba = _lib.PoolBwdArgs(p(x), p(q), p(w_in), p(b_in), p(w_out), p(dy), None, None, p(attn_w), p(probs), p(saved_o), p(saved_v),
                          p(dx), p(dquery), p(dw_in), p(db_in), p(dw_out), p(db_out), p(bwd_ws), bwd_ws_bytes, None, _lib.AECF_F32,
                          flags, p(saved_prep), None, p(saved_o_lo), 0.5)
    _lib.check(lib.aecf_pool_backward(ctypes.byref(desc), ctypes.byref(ba), _stream()), "aecf_pool_backward")
    torch.cuda.synchronize()
    gd.check()
    for name, t_ in (("dx", dx), ("dquery", dquery), ("dw_in", dw_in), ("db_in", db_in), ("dw_out", dw_out), ("db_out", db_out)):
        assert torch.isfinite(t_.float()).all(), name

    # a workspace one byte short is refused before anything is launched
    ba.workspace_bytes = bwd_ws_bytes - 1
    assert lib.aecf_pool_backward(ctypes.byref(desc), ctypes.byref(ba), _stream()) == -4
    fa.workspace_bytes = fwd_ws_bytes - 1
    assert lib.aecf_pool_forward(ctypes.byref(desc), ctypes.byref(fa), _stream()) == -4
