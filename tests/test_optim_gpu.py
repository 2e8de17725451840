"""FusedAdamW (aecf_adamw_step) against torch.optim.AdamW, the optimiser of the reference's example trainer
(xrays/train_xrays_example.py:322-323, 376): same update, same state layout, capturable as it stands."""
import copy

import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _params(dev, seed, n_tensors=5):
    g = torch.Generator().manual_seed(seed)
    shapes = [(256, 512), (256,), (3, 7, 5), (1,), (1031,), (64, 64)] + [(17 + i, 3) for i in range(max(0, n_tensors - 6))]
    return [torch.randn(*s, generator=g).to(dev).requires_grad_() for s in shapes[:n_tensors]]


@pytest.mark.parametrize("n_tensors", [5, 31])        # 31: two launches (24 tensors per launch)
def test_fused_adamw_matches_torch_adamw(n_tensors):
    from aecf_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    pa = _params(dev, 1, n_tensors)
    pb = [p.detach().clone().requires_grad_() for p in pa]
    oa = torch.optim.AdamW(pa, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.05)
    ob = FusedAdamW(pb, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.05)
    g = torch.Generator().manual_seed(2)
    for step in range(7):
        for a, b in zip(pa, pb):
            grad = torch.randn(a.shape, generator=g).to(dev) * (10.0 ** (step % 3 - 1))
            a.grad, b.grad = grad.clone(), grad.clone()
        if step == 3:
            pa[1].grad = None                         # a parameter without a gradient is skipped, its counter stays
            pb[1].grad = None
        oa.step()
        ob.step()
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert rel_err(b.detach().cpu(), a.detach().cpu()) < 2e-6, i
        assert rel_err(ob.state[b]["exp_avg"].cpu(), oa.state[a]["exp_avg"].cpu()) < 2e-6
        assert rel_err(ob.state[b]["exp_avg_sq"].cpu(), oa.state[a]["exp_avg_sq"].cpu()) < 2e-6
        assert float(ob.state[b]["step"]) == float(oa.state[a]["step"])
    # the state dict of one loads into the other
    oc = torch.optim.AdamW([p.detach().clone().requires_grad_() for p in pb], lr=3e-3, betas=(0.9, 0.98), weight_decay=0.05)
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))


def test_fused_adamw_replays_in_a_graph_with_a_fresh_step_count():
    from aecf_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    pa = _params(dev, 3)
    pb = [p.detach().clone().requires_grad_() for p in pa]
    oa = torch.optim.AdamW(pa, lr=1e-2, weight_decay=0.01)
    ob = FusedAdamW(pb, lr=1e-2, weight_decay=0.01)
    grads = [torch.zeros_like(p) for p in pb]
    for p, gbuf in zip(pb, grads):
        p.grad = gbuf
    ob.step()                                           # builds the state outside the capture (zero gradient: weight decay only)
    for a in pa:
        a.grad = torch.zeros_like(a)
    oa.step()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ob.step()
    # (the capture itself does not run the kernel)
    g = torch.Generator().manual_seed(4)
    for _ in range(5):
        for a, gbuf in zip(pa, grads):
            grad = torch.randn(a.shape, generator=g).to(dev)
            a.grad = grad.clone()
            gbuf.copy_(grad)
        oa.step()
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        assert rel_err(b.detach().cpu(), a.detach().cpu()) < 2e-6
        assert float(ob.state[b]["step"]) == float(oa.state[a]["step"]) == 6.0


def test_fused_adamw_continues_from_a_torch_adamw_state():
    """A state dict written by torch.optim.AdamW (step counters on the host) loads into FusedAdamW and the next steps agree."""
    from aecf_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    pa = _params(dev, 7)
    oa = torch.optim.AdamW(pa, lr=2e-3, weight_decay=0.02)
    g = torch.Generator().manual_seed(8)
    for _ in range(3):
        for a in pa:
            a.grad = torch.randn(a.shape, generator=g).to(dev)
        oa.step()
    pb = [p.detach().clone().requires_grad_() for p in pa]
    ob = FusedAdamW(pb, lr=2e-3, weight_decay=0.02)
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))
    for _ in range(4):
        for a, b in zip(pa, pb):
            grad = torch.randn(a.shape, generator=g).to(dev)
            a.grad, b.grad = grad.clone(), grad.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert rel_err(b.detach().cpu(), a.detach().cpu()) < 2e-6
        assert float(ob.state[b]["step"]) == float(oa.state[a]["step"]) == 7.0
        assert ob.state[b]["step"].device == b.device


def test_fused_adamw_refuses_cpu_parameters():
    from aecf_amd.optim import FusedAdamW
    p = torch.zeros(4, requires_grad=True)
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        FusedAdamW([p]).step()
