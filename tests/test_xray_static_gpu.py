"""Static routing and the captured training step of the example model (aecf_amd/xray.py): every branch
processes every row and a per-row select keeps the owner's result -- the same logits, loss and parameter gradients as the
compact routing, with no class sizes read back, so the whole optimisation step replays as one HIP graph."""
import copy

import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _batch(n, dev, seed):
    g = torch.Generator().manual_seed(seed)
    image, text = torch.randn(n, 512, generator=g), torch.randn(n, 512, generator=g)
    image[::5] = 0.0                                   # rows without an image / a text / either
    text[1::7] = 0.0
    image[3::35] = 0.0
    text[3::35] = 0.0
    labels = (torch.rand(n, 15, generator=g) < 0.2).float()
    return image.to(dev), text.to(dev), labels.to(dev)


def test_static_routing_equals_compact_routing():
    from aecf_amd.xray import AECFModel
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    compact = AECFModel(512, 512, 15).to(dev).train()
    for m in compact.modules():                       # (dropout off: the two forwards must see the same function)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    compact.toggle_curriculum(True)
    static = copy.deepcopy(compact)
    static.toggle_curriculum(True)
    static.static_routing = True
    image, text, labels = _batch(211, dev, 1)
    u = torch.rand(211, 1, 2, generator=torch.Generator().manual_seed(2)).to(dev)     # per batch row: both routings compact it the same way
    crit = torch.nn.BCEWithLogitsLoss()
    la, ia = compact(image, text, return_info=True, mask_uniforms=u)
    lb, ib = static(image, text, return_info=True, mask_uniforms=u)
    assert rel_err(lb.detach().cpu(), la.detach().cpu()) < 1e-5
    both = ib["both"]
    assert int(both.sum()) == ia["entropy"].shape[0]
    assert rel_err(ib["entropy"][both].float().cpu(), ia["entropy"].float().cpu()) < 1e-5
    crit(la, labels).backward()
    crit(lb, labels).backward()
    for (n, p), q in zip(compact.named_parameters(), static.parameters()):
        assert rel_err(q.grad.cpu(), p.grad.cpu()) < 2e-5, n


def test_graphed_train_step_equals_eager_steps():
    from aecf_amd.xray import AECFModel, GraphedTrainStep, train_step
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    eager = AECFModel(512, 512, 15).to(dev).train()
    for m in eager.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    graphed_model = copy.deepcopy(eager)              # (no curriculum, no missing-modality draws: a deterministic step)
    crit = torch.nn.BCEWithLogitsLoss()
    opt_e = torch.optim.AdamW(eager.parameters(), lr=1e-3, weight_decay=0.01)
    opt_g = torch.optim.AdamW(graphed_model.parameters(), lr=1e-3, weight_decay=0.01, capturable=True)
    state = copy.deepcopy(graphed_model.state_dict())
    step = GraphedTrainStep(graphed_model, opt_g, crit, 64, 512, 512, 15, dev, warmup=3)
    graphed_model.load_state_dict(state)              # undo the warm-up / capture steps: same start as the eager model
    for st in opt_g.state.values():
        for k, v in st.items():
            if torch.is_tensor(v):
                v.zero_()
    losses_e, losses_g = [], []
    for k in range(6):
        image, text, labels = _batch(64, dev, 10 + k)
        losses_e.append(float(train_step(eager, opt_e, crit, image, text, labels)[0]))
        losses_g.append(float(step(image, text, labels)))
    for a, b in zip(losses_e, losses_g):
        assert abs(a - b) < 1e-4 * max(1.0, abs(a)), (losses_e, losses_g)
    for p, q in zip(eager.parameters(), graphed_model.parameters()):
        assert rel_err(q.detach().cpu(), p.detach().cpu()) < 1e-4


@pytest.mark.timeout(300)
def test_bench_pool_step_replays_as_one_graph():
    """`bench.py --graph`: the pool forward + entropy loss + backward captured once by torch.cuda.graphs (the library's launches
    join the caller's capture) and replayed -- what makes the host-bound shards device-bound."""
    import json
    import os
    import subprocess
    import sys
    from tests.helpers import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny", "--graph", "--steps", "20", "--warmup", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["graph_replay"] is True and line["value"] > 0
