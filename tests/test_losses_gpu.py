"""GPU tests of the loss side (SURVEY.md 8a rows A6 + A9).  The contrastive term is build-defined (not in the
reference): parity is UNPINNED; the HIP path is checked against the oracle's closed form (itself checked against
autograd in tests/test_oracle_golden.py)."""
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("n,d", [(128, 64), (320, 192)])
def test_info_nce_matches_oracle(dtype, tol, n, d):
    from aecf_amd import losses
    from oracle import aecf_oracle as O
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n + d)
    za = torch.randn(n, d, generator=g).to(dtype)
    zb = (za.float() * 0.7 + 0.5 * torch.randn(n, d, generator=g)).to(dtype)
    a = za.to(dev).requires_grad_(True)
    b = zb.to(dev).requires_grad_(True)
    loss = losses.info_nce(a, b, temperature=0.1)
    loss.backward()
    want = O.info_nce(za.double(), zb.double(), 0.1)
    dza, dzb = O.info_nce_backward(za.double(), zb.double(), 0.1)
    assert abs(float(loss) - float(want)) < tol * max(1.0, abs(float(want)))
    assert rel_err(a.grad.float().cpu(), dza) < tol * 5
    assert rel_err(b.grad.float().cpu(), dzb) < tol * 5


def test_nce_direction_row_offsets_emulate_two_ranks():
    """The sharded call (local rows + offset into the gathered keys) sums to the unsharded one: what two ranks compute."""
    from aecf_amd.losses import _NceDirection, l2_normalize
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    q = l2_normalize(torch.randn(256, 128, generator=g).to(dev))
    k = l2_normalize(torch.randn(256, 128, generator=g).to(dev))
    full = _NceDirection.apply(q, k, 0, 0.07, 0.5 / 256)
    lo = _NceDirection.apply(q[:96], k, 0, 0.07, 0.5 / 256)
    hi = _NceDirection.apply(q[96:], k, 96, 0.07, 0.5 / 256)
    assert abs(float(full) - float(lo + hi)) < 1e-5 * abs(float(full))


def test_l2_normalize_forward_backward():
    from aecf_amd.losses import l2_normalize
    dev = torch.device("cuda:0")
    z = torch.randn(70, 96, generator=torch.Generator().manual_seed(1))
    z[3] = 0.0                                            # zero row: stays zero, finite gradient
    zd = z.to(dev).requires_grad_(True)
    zn = l2_normalize(zd)
    w = torch.randn(70, 96, generator=torch.Generator().manual_seed(2))
    (zn * w.to(dev)).sum().backward()
    zc = z.clone().requires_grad_(True)
    ref = zc / zc.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    (ref * w).sum().backward()
    assert rel_err(zn.detach().cpu(), ref.detach()) < 1e-6
    ok = torch.ones(70, dtype=torch.bool)
    ok[3] = False
    assert rel_err(zd.grad.cpu()[ok], zc.grad[ok]) < 1e-5
    assert torch.isfinite(zd.grad).all()


def test_fusion_objective_composes():
    import aecf_amd
    from aecf_amd import losses
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    query, pool = aecf_amd.create_fusion_pool(128, 3, num_heads=4)
    pool = pool.to(dev).eval()           # eval: the entropy keeps its graph, so the regulariser has a gradient
    query = query.detach().to(dev).requires_grad_(True)
    xa = torch.randn(128, 3, 128, device=dev)
    xb = xa + 0.1 * torch.randn_like(xa)
    za, ia = pool(query.expand(128, -1, -1), xa, return_info=True)
    zb, _ = pool(query.expand(128, -1, -1), xb, return_info=True)
    task = za.float().pow(2).mean()
    total = losses.fusion_objective(task, pool.curriculum_masking, ia["entropy"], za.squeeze(1), zb.squeeze(1),
                                    temperature=0.1)
    total.backward()
    assert torch.isfinite(total) and query.grad is not None and torch.isfinite(query.grad).all()
    assert float(query.grad.abs().sum()) > 0


def _nce_reference(q, k, off, T, coef, rows_sel=None, cols_sel=None):
    """float32 torch reference of one direction on the device, in row chunks (the logits of a chunk only)."""
    qf, kf = q.float(), k.float()
    rows, cols = qf.shape[0], kf.shape[0]
    lse = torch.empty(rows, device=q.device)
    loss = torch.empty(rows, device=q.device)
    dq = torch.zeros_like(qf)
    dk = torch.zeros_like(kf)
    for r0 in range(0, rows, 1024):
        r1 = min(r0 + 1024, rows)
        s = qf[r0:r1] @ kf.T / T
        lse[r0:r1] = torch.logsumexp(s, dim=1)
        idx = torch.arange(r0, r1, device=q.device)
        loss[r0:r1] = lse[r0:r1] - s[idx - r0, off + idx]
        p = torch.exp(s - lse[r0:r1, None])
        p[idx - r0, off + idx] -= 1.0
        p *= coef / T
        dq[r0:r1] = p @ kf
        dk += p.T @ qf[r0:r1]
    return loss, dq, dk


@pytest.mark.parametrize("low_memory", [False, True], ids=["gemm", "stream"])
@pytest.mark.parametrize("rows,cols,off,d", [(128, 128, 0, 128), (100, 333, 57, 256), (640, 2048, 1000, 512),
                                             (97, 1500, 3, 768), (64, 700, 600, 1024), (200, 200, 0, 384)])
def test_flash_nce_matches_float32_reference(rows, cols, off, d, low_memory):
    """One InfoNCE direction -- tile-GEMM form (aecf_nce_gemm.hip, the default) and streaming form (aecf_nce_flash.hip, chosen
    by handing over the O(rows d) workspace) -- against float32 torch math on the same bf16 inputs: ragged row / column
    counts (no % 64 restriction), key splits, every supported width, positives at an offset (a data-parallel shard)."""
    from aecf_amd.losses import _NceDirection, l2_normalize
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + cols + d)
    q = l2_normalize(torch.randn(rows, d, generator=g).to(dev, torch.bfloat16)).detach().requires_grad_(True)
    k = l2_normalize(torch.randn(cols, d, generator=g).to(dev, torch.bfloat16)).detach()
    with torch.no_grad():
        k[off:off + rows] = (0.8 * q.detach().float() + 0.6 * k[off:off + rows].float()).to(torch.bfloat16)   # real positives
    k.requires_grad_(True)
    T, coef = 0.07, 0.5 / cols
    out = _NceDirection.apply(q, k, off, T, coef, low_memory)
    out.backward()
    loss, dq, dk = _nce_reference(q.detach(), k.detach(), off, T, coef)
    assert abs(float(out) - float(loss.sum() * coef)) < 2e-3 * abs(float(loss.sum() * coef))
    assert rel_err(q.grad.float().cpu(), dq.cpu()) < 1.5e-2        # P is rounded to bf16 for its MFMA (2^-9 per weight)
    assert rel_err(k.grad.float().cpu(), dk.cpu()) < 1.5e-2


def test_flash_nce_config3_size():
    """BASELINE configs[2]: 8192 local rows against 65536 gathered keys, d = 768, bf16 -- the STREAMING form: 2.1 GB of
    float32 logits that are never materialised (workspace O(rows d): ~50 MB).  Checked against float32 torch math (row-chunked) and, on a
    row subset, against the CPU oracle's closed form."""
    from aecf_amd import _lib
    from aecf_amd.losses import _NceDirection, l2_normalize
    from oracle import aecf_oracle as O
    dev = torch.device("cuda:0")
    rows, cols, d, off, T = 8192, 65536, 768, 3 * 8192, 0.07
    g = torch.Generator(device=dev).manual_seed(5)
    q = l2_normalize(torch.randn(rows, d, device=dev, generator=g).to(torch.bfloat16)).detach()
    k = l2_normalize(torch.randn(cols, d, device=dev, generator=g).to(torch.bfloat16)).detach()
    k[off:off + rows] = (0.9 * q.float() + 0.45 * k[off:off + rows].float()).to(torch.bfloat16)
    ws = _lib.load().aecf_nce_stream_workspace_bytes(rows, cols, d, _lib.AECF_BF16)
    assert ws < 64 << 20, ws                                       # O(rows d), not O(rows cols)
    q.requires_grad_(True)
    k.requires_grad_(True)
    coef = 0.5 / cols
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out = _NceDirection.apply(q, k, off, T, coef, True)
    out.backward()
    torch.cuda.synchronize()
    assert torch.cuda.max_memory_allocated() - base < 1200 << 20   # outputs + gradients + workspace; no logits
    loss, dq, dk = _nce_reference(q.detach(), k.detach(), off, T, coef)
    assert abs(float(out) - float(loss.sum() * coef)) < 2e-3 * abs(float(loss.sum() * coef))
    assert rel_err(q.grad.float(), dq) < 1.5e-2 and rel_err(k.grad.float(), dk) < 1.5e-2
    # oracle closed form on a row subset (CPU, float64)
    sel = torch.arange(0, rows, 257)
    qs, kc = q.detach()[sel].double().cpu(), k.detach().double().cpu()
    s = qs @ kc.T / T
    want = torch.logsumexp(s, 1) - s[torch.arange(len(sel)), off + sel]
    got_rows = (dq[sel].cpu(), None)
    p = torch.softmax(s, 1)
    p[torch.arange(len(sel)), off + sel] -= 1.0
    assert rel_err(q.grad.float()[sel].cpu(), (coef / T) * (p @ kc)) < 1.5e-2
    assert rel_err(loss[sel].cpu(), want) < 1e-3


def test_contrastive_plus_entropy_loss_in_one_call():
    """north_star "second fused kernel": InfoNCE + entropy_loss with their gradients from one C-ABI call equal the two
    separate operators."""
    import aecf_amd
    from aecf_amd import losses
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    za = torch.randn(256, 256, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    zb = (za.detach().float() * 0.8 + 0.5 * torch.randn(256, 256, generator=g).to(dev)).to(torch.bfloat16).requires_grad_(True)
    ent = (torch.rand(256, 1, generator=g) * 1.0).to(dev).requires_grad_(True)
    cm = aecf_amd.CurriculumMasking().to(dev)
    cm._last_seq_len = 3
    fused = losses.contrastive_entropy_loss(za, zb, cm, ent, temperature=0.1, entropy_weight=0.5)
    fused.backward()
    ga, gb, ge = za.grad.clone(), zb.grad.clone(), ent.grad.clone()
    za.grad = zb.grad = ent.grad = None
    sep = losses.info_nce(za, zb, temperature=0.1) + 0.5 * cm.entropy_loss(ent)
    sep.backward()
    assert abs(float(fused) - float(sep)) < 1e-3 * abs(float(sep))
    assert rel_err(ga.float(), za.grad.float()) < 1e-5 and rel_err(gb.float(), zb.grad.float()) < 1e-5
    assert rel_err(ge, ent.grad) < 1e-5


@pytest.mark.parametrize("n,d", [(1024, 192), (1024, 512), (320, 320)])
def test_info_nce_low_temperature_takes_a_form_sized_for_itself(n, d):
    """CLIP-style T = 0.02 is below what the tile GEMMs' constant-shift softmax takes (T >= 0.025): the call falls through to
    the streaming kernels (d in their set) or to the logits-in-memory form (d = 192, 320).  ADVICE r3 (high): that form
    used to check its workspace against the tile-GEMM size and write past the buffer.  Guard bytes behind the workspace the
    C ABI is handed must survive, and the loss must match float32 math."""
    import ctypes
    from aecf_amd import _lib, losses
    from aecf_amd.layer import _ptr, _stream
    dev = torch.device("cuda:0")
    T = 0.02
    g = torch.Generator().manual_seed(n + d)
    za = torch.randn(n, d, generator=g).to(torch.bfloat16)
    # weakly correlated views: at T = 0.02 a strong positive makes the loss ~0 and its gradient rounding noise on both sides
    zb = (za.float() * 0.12 + torch.randn(n, d, generator=g)).to(torch.bfloat16)
    # (1) the public operator against float32 math on the same bf16 inputs
    a = za.to(dev).requires_grad_(True)
    b = zb.to(dev).requires_grad_(True)
    loss = losses.info_nce(a, b, temperature=T)
    loss.backward()
    af = za.float().to(dev).requires_grad_(True)
    bf = zb.float().to(dev).requires_grad_(True)
    na, nb = torch.nn.functional.normalize(af, dim=-1), torch.nn.functional.normalize(bf, dim=-1)
    lg = na @ nb.t() / T
    tgt = torch.arange(n, device=dev)
    want = 0.5 * (torch.nn.functional.cross_entropy(lg, tgt) + torch.nn.functional.cross_entropy(lg.t(), tgt))
    want.backward()
    assert abs(float(loss) - float(want)) < 3e-2 * max(1.0, abs(float(want)))
    assert rel_err(a.grad.float().cpu(), af.grad.cpu()) < 0.1
    # (2) the C ABI: a buffer of exactly aecf_nce_workspace_bytes followed by guard bytes
    lib = _lib.load()
    q = losses.l2_normalize(za.to(dev)).detach().contiguous()
    k = losses.l2_normalize(zb.to(dev)).detach().contiguous()
    need = lib.aecf_nce_workspace_bytes(n, n, d, _lib.AECF_BF16)
    guard = 1 << 20
    buf = torch.full((need + guard,), 0x5A, dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    lr, dq, dk = torch.empty(n, **f32), torch.empty(n, d, **f32), torch.empty(n, d, **f32)
    _lib.check(lib.aecf_nce_fwd_bwd(n, n, 0, d, _lib.AECF_BF16, T, 0.5 / n, _ptr(q), _ptr(k), _ptr(lr), _ptr(dq), _ptr(dk),
                                    _ptr(buf), need, _stream()), "aecf_nce_fwd_bwd")
    torch.cuda.synchronize()
    assert bool((buf[need:] == 0x5A).all()), "the call wrote past the workspace it was given"
    assert torch.isfinite(lr).all() and torch.isfinite(dq).all() and torch.isfinite(dk).all()
    # one byte less than the form's own need is refused, not overrun
    if need > 4096:
        st = lib.aecf_nce_fwd_bwd(n, n, 0, d, _lib.AECF_BF16, T, 0.5 / n, _ptr(q), _ptr(k), _ptr(lr), _ptr(dq), _ptr(dk),
                                  _ptr(buf), 4096, _stream())
        assert st != 0


def test_sym_memory_guard_falls_back_to_the_streaming_form(monkeypatch):
    """when the symmetric form is refused for memory, the fallbacks must not allocate rows x cols either."""
    from aecf_amd import losses
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    n, d = 512, 256
    za = torch.randn(n, d, generator=g).to(torch.bfloat16).to(dev)
    zb = (za.float() * 0.7 + 0.5 * torch.randn(n, d, generator=g).to(dev)).to(torch.bfloat16)
    want = float(losses.info_nce(za, zb, temperature=0.1))
    monkeypatch.setattr(losses, "_sym_supported", lambda *a, **k: False)
    seen = []
    real = losses._NceDirection.apply

    def spy(*args):
        seen.append(bool(args[5]) if len(args) > 5 else False)
        return real(*args)
    monkeypatch.setattr(losses._NceDirection, "apply", staticmethod(spy))
    got = float(losses.info_nce(za, zb, temperature=0.1))
    assert seen == [True, True], f"the fallback must select the streaming kernels (low_memory), saw {seen}"
    assert abs(got - want) < 2e-2 * max(1.0, abs(want))
