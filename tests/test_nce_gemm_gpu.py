"""GPU tests of the tile-GEMM InfoNCE (aecf_nce_gemm.hip; SURVEY.md 8a row A9, build-defined: parity unpinned, checked
against float32 torch math on the same bf16 inputs and, through tests/test_losses_gpu.py, against the CPU oracle)."""
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _inputs(n_all, d, seed, dev):
    from aecf_amd.losses import l2_normalize
    g = torch.Generator(device=dev).manual_seed(seed)
    a = l2_normalize(torch.randn(n_all, d, device=dev, generator=g).to(torch.bfloat16)).detach()
    b = l2_normalize((0.8 * a.float() + 0.6 * l2_normalize(torch.randn(n_all, d, device=dev, generator=g)).float())
                     .to(torch.bfloat16)).detach()
    return a, b


def _symmetric_reference(a, b, T, chunk=2048):
    """float32 math of the full symmetric objective 0.5/n sum_i [CE(a_i.b/T, i) + CE(b_i.a/T, i)] and its gradients,
    in row chunks (two sweeps: sums, then gradients)."""
    af, bf = a.float(), b.float()
    n = af.shape[0]
    coef = 0.5 / n
    row_lse = torch.empty(n, device=a.device)
    col_sum = torch.zeros(n, device=a.device, dtype=torch.float64)
    diag = (af * bf).sum(1) / T
    for r0 in range(0, n, chunk):
        s = af[r0:r0 + chunk] @ bf.T / T
        row_lse[r0:r0 + chunk] = torch.logsumexp(s, 1)
        col_sum += torch.exp(s.double() - 1.0 / T).sum(0)
    col_lse = (torch.log(col_sum) + 1.0 / T).float()
    loss_rows = (row_lse - diag) + (col_lse - diag)
    da, db = torch.zeros_like(af), torch.zeros_like(bf)
    for r0 in range(0, n, chunk):
        s = af[r0:r0 + chunk] @ bf.T / T
        w = torch.exp(s - row_lse[r0:r0 + chunk, None]) + torch.exp(s - col_lse[None, :])
        idx = torch.arange(r0, min(r0 + chunk, n), device=a.device)
        w[idx - r0, idx] -= 2.0
        w *= coef / T
        da[r0:r0 + chunk] = w @ bf
        db += w.T @ af[r0:r0 + chunk]
    return loss_rows, da, db


def _run_sharded(a, b, T, bounds):
    """What the ranks of a data-parallel group compute: pass 1 per shard, the column sums added up (the all-reduce), pass 2
    per shard, the shares of db added up (the reduce-scatter)."""
    from aecf_amd import _lib
    from aecf_amd.layer import _ptr, _stream
    lib = _lib.load()
    n, d = a.shape
    dev = a.device
    f32 = dict(dtype=torch.float32, device=dev)
    coef = 0.5 / n
    shards = []
    col_total = torch.zeros(n, **f32)
    for lo, hi in bounds:
        rows = hi - lo
        ws_bytes = lib.aecf_nce_sym_workspace_bytes(rows, n, d)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        cs = torch.empty(n, **f32)
        al = a[lo:hi].contiguous()
        _lib.check(lib.aecf_nce_sym_pass1(rows, n, d, T, _ptr(al), _ptr(b), _ptr(ws), ws_bytes, _ptr(cs), _stream()), "pass1")
        col_total += cs
        shards.append((lo, hi, al, ws, ws_bytes))
    loss_rows, da, db = torch.empty(n, **f32), torch.empty(n, d, **f32), torch.zeros(n, d, **f32)
    for lo, hi, al, ws, ws_bytes in shards:
        rows = hi - lo
        lr, da_s, db_s = torch.empty(rows, **f32), torch.empty(rows, d, **f32), torch.empty(n, d, **f32)
        _lib.check(lib.aecf_nce_sym_loss(rows, n, lo, d, T, _ptr(al), _ptr(b), _ptr(col_total), _ptr(ws), ws_bytes, _ptr(lr), 0, 2,
                                         0.0, None, 1.0, None, None, _stream()), "loss")
        _lib.check(lib.aecf_nce_sym_grads(rows, n, lo, d, T, coef, _ptr(al), _ptr(b), _ptr(ws), ws_bytes, None, _lib.AECF_F32,
                                          _ptr(da_s), _ptr(db_s), _stream()), "grads")
        loss_rows[lo:hi], da[lo:hi] = lr, da_s
        db += db_s
    return loss_rows, da, db


@pytest.mark.parametrize("n,d,bounds", [(256, 128, [(0, 256)]), (333, 256, [(0, 333)]), (1000, 512, [(0, 400), (400, 1000)]),
                                        (1500, 768, [(0, 97), (97, 1100), (1100, 1500)]), (700, 1024, [(0, 700)]),
                                        (520, 192, [(0, 260), (260, 520)])])
def test_symmetric_nce_matches_float32_reference(n, d, bounds):
    """Both directions from one logits block, one rank and emulated shards (ragged sizes, positives at an offset)."""
    dev = torch.device("cuda:0")
    a, b = _inputs(n, d, n + d, dev)
    T = 0.07
    loss_rows, da, db = _run_sharded(a, b, T, bounds)
    want_rows, want_da, want_db = _symmetric_reference(a, b, T)
    # the loss of a well-separated positive is a small difference of O(1/T) terms: absolute floor of float32 logits
    assert torch.allclose(loss_rows.cpu(), want_rows.cpu(), rtol=1e-3, atol=2e-4)
    assert rel_err(da.cpu(), want_da.cpu()) < 1.5e-2            # the softmax weights are rounded to bf16 for their MFMAs
    assert rel_err(db.cpu(), want_db.cpu()) < 1.5e-2


def test_symmetric_nce_is_deterministic_and_shard_invariant():
    dev = torch.device("cuda:0")
    a, b = _inputs(1024, 256, 7, dev)
    one = _run_sharded(a, b, 0.07, [(0, 1024)])
    again = _run_sharded(a, b, 0.07, [(0, 1024)])
    for x, y in zip(one, again):
        assert torch.equal(x, y)
    two = _run_sharded(a, b, 0.07, [(0, 512), (512, 1024)])
    assert rel_err(two[0].cpu(), one[0].cpu()) < 1e-5
    assert rel_err(two[1].cpu(), one[1].cpu()) < 2e-3           # column sums arrive in another order: bf16 weights move by an ulp
    assert rel_err(two[2].cpu(), one[2].cpu()) < 2e-3


def test_info_nce_uses_the_symmetric_form_and_matches_the_oracle():
    from aecf_amd import losses
    from oracle import aecf_oracle as O
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    za = torch.randn(320, 192, generator=g).to(torch.bfloat16)
    zb = (za.float() * 0.7 + 0.5 * torch.randn(320, 192, generator=g)).to(torch.bfloat16)
    a = za.to(dev).requires_grad_(True)
    b = zb.to(dev).requires_grad_(True)
    assert losses._sym_supported(a, 0.1)
    loss = losses.info_nce(a, b, temperature=0.1)
    loss.backward()
    want = O.info_nce(za.double(), zb.double(), 0.1)
    dza, dzb = O.info_nce_backward(za.double(), zb.double(), 0.1)
    assert abs(float(loss) - float(want)) < 2e-2 * max(1.0, abs(float(want)))
    assert rel_err(a.grad.float().cpu(), dza) < 0.1 and rel_err(b.grad.float().cpu(), dzb) < 0.1


def test_symmetric_nce_config3_size():
    """BASELINE configs[2]: one rank's block of the global problem -- 8192 local rows against 65536 gathered keys, d = 768.
    The column direction needs every rank's rows, so the check runs the 8 shards of a 16384-row problem at d = 768 and, at the
    full 8192 x 65536 block, the row direction plus the column sums against float32 math."""
    from aecf_amd import _lib
    from aecf_amd.layer import _ptr, _stream
    dev = torch.device("cuda:0")
    lib = _lib.load()
    n, d, T = 16384, 768, 0.07
    a, b = _inputs(n, d, 3, dev)
    bounds = [(i * 2048, (i + 1) * 2048) for i in range(8)]
    loss_rows, da, db = _run_sharded(a, b, T, bounds)
    want_rows, want_da, want_db = _symmetric_reference(a, b, T)
    assert rel_err(loss_rows, want_rows) < 1e-3
    assert rel_err(da, want_da) < 1.5e-2 and rel_err(db, want_db) < 1.5e-2
    del loss_rows, da, db, want_rows, want_da, want_db
    # full block: rows 3 x 8192 .. 4 x 8192 of a 65536-row problem
    rows, cols, off = 8192, 65536, 3 * 8192
    a, b = _inputs(cols, d, 5, dev)
    al = a[off:off + rows].contiguous()
    ws_bytes = lib.aecf_nce_sym_workspace_bytes(rows, cols, d)
    assert ws_bytes < (rows * cols * 2) * 1.25 + (64 << 20)      # E + the float32 split slabs of da + O(rows + cols)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    cs = torch.empty(cols, **f32)
    _lib.check(lib.aecf_nce_sym_pass1(rows, cols, d, T, _ptr(al), _ptr(b), _ptr(ws), ws_bytes, _ptr(cs), _stream()), "pass1")
    want_cs = torch.zeros(cols, device=dev, dtype=torch.float64)
    row_lse = torch.empty(rows, device=dev)
    for r0 in range(0, rows, 1024):
        s = al[r0:r0 + 1024].float() @ b.float().T / T
        want_cs += torch.exp(s.double() - 1.0 / T).sum(0)
        row_lse[r0:r0 + 1024] = torch.logsumexp(s, 1)
    assert rel_err(cs.double(), want_cs) < 1e-3
    # pass 2 with this block's own column sums standing in for the all-reduced ones: weights E (1/l_i + 1/c_j)
    coef = 0.5 / cols
    lr, da, db = torch.empty(rows, **f32), torch.empty(rows, d, **f32), torch.empty(cols, d, **f32)
    _lib.check(lib.aecf_nce_sym_loss(rows, cols, off, d, T, _ptr(al), _ptr(b), _ptr(cs), _ptr(ws), ws_bytes, _ptr(lr), 0, 2, 0.0,
                                     None, 1.0, None, None, _stream()), "loss")
    _lib.check(lib.aecf_nce_sym_grads(rows, cols, off, d, T, coef, _ptr(al), _ptr(b), _ptr(ws), ws_bytes, None, _lib.AECF_F32,
                                      _ptr(da), _ptr(db), _stream()), "grads")
    col_lse = (torch.log(want_cs) + 1.0 / T).float()
    want_da, want_db = torch.zeros(rows, d, **f32), torch.zeros(cols, d, **f32)
    bf = b.float()
    for r0 in range(0, rows, 1024):
        s = al[r0:r0 + 1024].float() @ bf.T / T
        w = torch.exp(s - row_lse[r0:r0 + 1024, None]) + torch.exp(s - col_lse[None, :])
        idx = torch.arange(r0, r0 + 1024, device=dev)
        w[idx - r0, off + idx] -= 2.0
        w *= coef / T
        want_da[r0:r0 + 1024] = w @ bf
        want_db += w.T @ al[r0:r0 + 1024].float()
    assert rel_err(da, want_da) < 1.5e-2 and rel_err(db, want_db) < 1.5e-2
    diag = (al.float() * bf[off:off + rows]).sum(1) / T
    assert rel_err(lr, (row_lse - diag) + (col_lse[off:off + rows] - diag)) < 1e-3


def test_symmetric_nce_gradients_in_bf16_scaled_on_the_device():
    """aecf_nce_sym_grads with an upstream scalar in device memory and bf16 outputs (what the autograd backward asks for)
    equals the float32 gradients times that scalar, rounded once."""
    from aecf_amd import _lib
    from aecf_amd.layer import _ptr, _stream
    dev = torch.device("cuda:0")
    lib = _lib.load()
    n, d, T = 777, 256, 0.07
    a, b = _inputs(n, d, 21, dev)
    coef = 0.5 / n
    f32 = dict(dtype=torch.float32, device=dev)
    outs = []
    for gdt, up in ((torch.float32, None), (torch.bfloat16, torch.tensor([0.375], **f32))):
        ws_bytes = lib.aecf_nce_sym_workspace_bytes(n, n, d)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        cs, lr = torch.empty(n, **f32), torch.empty(n, **f32)
        da, db = torch.empty(n, d, dtype=gdt, device=dev), torch.empty(n, d, dtype=gdt, device=dev)
        _lib.check(lib.aecf_nce_sym_pass1(n, n, d, T, _ptr(a), _ptr(b), _ptr(ws), ws_bytes, _ptr(cs), _stream()), "pass1")
        _lib.check(lib.aecf_nce_sym_loss(n, n, 0, d, T, _ptr(a), _ptr(b), _ptr(cs), _ptr(ws), ws_bytes, _ptr(lr), 0, 2, 0.0, None,
                                         1.0, None, None, _stream()), "loss")
        _lib.check(lib.aecf_nce_sym_grads(n, n, 0, d, T, coef, _ptr(a), _ptr(b), _ptr(ws), ws_bytes, None if up is None else _ptr(up),
                                          _lib.AECF_BF16 if gdt == torch.bfloat16 else _lib.AECF_F32, _ptr(da), _ptr(db), _stream()),
                   "grads")
        outs.append((da.float(), db.float()))
    # the weights are rounded to bf16 AFTER the scaling (another draw of the same rounding: each run sits within 1.5e-2 of
    # float32 math, see test_symmetric_nce_matches_float32_reference) and the outputs carry one bf16 rounding
    e_a, e_b = rel_err(outs[1][0], 0.375 * outs[0][0]), rel_err(outs[1][1], 0.375 * outs[0][1])
    assert e_a < 2e-2 and e_b < 2e-2, (e_a, e_b)
