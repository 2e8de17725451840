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
