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
    # the constructor's warm-up and capture steps run on noise; it restores parameters and optimizer state itself
    step = GraphedTrainStep(graphed_model, opt_g, crit, 64, 512, 512, 15, dev, warmup=3)
    for p, q in zip(eager.parameters(), graphed_model.parameters()):
        assert torch.equal(p.detach(), q.detach())
    for st in opt_g.state.values():
        for k, v in st.items():
            if torch.is_tensor(v):
                assert float(v.abs().max()) == 0.0, k
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


@pytest.mark.timeout(300)
@pytest.mark.parametrize("extra", [["--graph"], ["--graph", "--no-tune-gemm"], []])
def test_bench_example_model_step(extra):
    """`bench.py --config c4`: the example model's optimisation step, captured (static routing, FusedAdamW inside the graph, the
    nn.Linear GEMMs picked by TunableOp during the warm-up or left on torch's default) and eager."""
    import json
    import os
    import subprocess
    import sys
    from tests.helpers import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c4", "--batch", "64", "--steps", "10", "--warmup", "3"]
                       + extra, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["value"] > 0 and line["dtype"] == "f32" and line["config"]["global_batch"] == 64
    assert ("captured" in line["config"]["workload"]) == ("--graph" in extra)
    assert ("TunableOp" in line["gemm_selection"]) == (extra == ["--graph"])


def test_graphed_step_keeps_a_trained_optimizers_state():
    """A GraphedTrainStep built around an optimizer that has already stepped gives its moments and step counters back."""
    from aecf_amd.xray import AECFModel, GraphedTrainStep, train_step
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    model = AECFModel(512, 512, 15).to(dev).train()
    crit = torch.nn.BCEWithLogitsLoss()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01, capturable=True)
    for k in range(2):
        image, text, labels = _batch(64, dev, 40 + k)
        train_step(model, opt, crit, image, text, labels)
    before_p = [p.detach().clone() for p in model.parameters()]
    before_s = copy.deepcopy(opt.state_dict()["state"])
    GraphedTrainStep(model, opt, crit, 64, 512, 512, 15, dev, warmup=2)
    for p, q in zip(before_p, model.parameters()):
        assert torch.equal(p, q.detach())
    after_s = opt.state_dict()["state"]
    for idx, st in before_s.items():
        for name, val in st.items():
            if torch.is_tensor(val):
                assert torch.equal(val, after_s[idx][name]), (idx, name)


def test_graphed_step_with_curriculum_masking_and_modality_draws():
    """Capture with the random parts ON (curriculum masking in the pool, missing-modality training draws): the replayed steps
    must draw fresh randomness per replay (the device generator's offset advances) and train -- losses finite, not
    constant, parameters moving; and a replay sequence is reproducible from the same seed."""
    from aecf_amd.xray import AECFModel, GraphedTrainStep
    dev = torch.device("cuda:0")

    def run(seed):
        torch.manual_seed(seed)
        model = AECFModel(512, 512, 15).to(dev).train()
        model.toggle_curriculum(True)
        crit = torch.nn.BCEWithLogitsLoss()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01, capturable=True)
        start = [p.detach().clone() for p in model.parameters()]
        step = GraphedTrainStep(model, opt, crit, 64, 512, 512, 15, dev, warmup=2)
        torch.cuda.manual_seed(seed + 100)            # (the constructor's draws advanced the generator: pin the replays' stream)
        out = []
        for k in range(5):
            image, text, labels = _batch(64, dev, 20 + k)
            out.append(float(step(image, text, labels)))
        moved = max(float((p.detach() - q).abs().max()) for p, q in zip(model.parameters(), start))
        return out, moved

    a, moved = run(11)
    b, _ = run(11)
    assert all(torch.isfinite(torch.tensor(a))) and moved > 0
    assert len(set(round(v, 6) for v in a)) > 1
    assert a == b, (a, b)


def test_static_routing_ignores_non_finite_features_of_absent_modalities():
    """a row whose ABSENT modality holds NaN / Inf is excluded by compact routing (presence = norm > 1e-6 is
    False for it); static routing must give the same logits and finite, equal gradients -- select, not multiply by 0."""
    from aecf_amd.xray import AECFModel
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    compact = AECFModel(512, 512, 15).to(dev).train()
    for m in compact.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    static = copy.deepcopy(compact)
    static.static_routing = True
    image, text, labels = _batch(96, dev, 7)
    image[5] = float("nan")                            # absent image (a NaN norm is not > 1e-6), text present
    text[8] = float("nan")                             # absent text
    crit = torch.nn.BCEWithLogitsLoss()
    la = compact(image, text)
    lb = static(image, text)
    assert torch.isfinite(lb).all()
    assert rel_err(lb.detach().cpu(), la.detach().cpu()) < 1e-5
    crit(la, labels).backward()
    crit(lb, labels).backward()
    for (n, p), q in zip(compact.named_parameters(), static.parameters()):
        assert torch.isfinite(q.grad).all(), n         # static routing feeds absent rows to the encoders as zeros
        if "encoder" not in n:                         # (compact routing, like the reference, runs the encoders on every row:
            assert rel_err(q.grad.cpu(), p.grad.cpu()) < 2e-5, n      #  its encoder weight gradients are NaN here)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_front_pair_equals_the_separate_front_ends(dtype):
    """aecf_front_pair against draw_missing + modality_frontend x 2 + the presence classes, on the same uniforms; rows with a NaN,
    an all-zero and a dropped modality included."""
    from aecf_amd.xray import AECFModel, Route, front_pair, modality_frontend
    dev = torch.device("cuda:0")
    n = 333
    g = torch.Generator().manual_seed(11)
    image = torch.randn(n, 512, generator=g).to(dev, dtype)
    text = torch.randn(n, 320, generator=g).to(dev, dtype)
    image[4::9] = 0.0
    text[5::11] = 0.0
    image[7, 100] = float("nan")
    text[8, 3] = float("inf")
    model = AECFModel(512, 320, 15)
    gen = torch.Generator(device=dev).manual_seed(21)
    drop_a, drop_b = model.draw_missing(n, dev, generator=gen)
    gen.manual_seed(21)
    u = torch.rand(3, n, device=dev, generator=gen)
    ref_a, has_a = modality_frontend(image, drop_a)
    ref_b, has_b = modality_frontend(text, drop_b)
    route = Route(has_a, has_b)
    for kwargs in (dict(uniforms=u), dict(drop=(drop_a, drop_b))):
        xa, xb, pa, pb, cls = front_pair(image, text, missing_prob=0.3, **kwargs)
        assert torch.equal(pa, has_a) and torch.equal(pb, has_b)
        assert torch.equal(cls, route.cls)
        za, zb = torch.zeros_like(ref_a), torch.zeros_like(ref_b)
        assert torch.equal(xa, torch.where(has_a.bool().unsqueeze(1), ref_a, za))
        assert torch.equal(xb, torch.where(has_b.bool().unsqueeze(1), ref_b, zb))
    assert 0 < int(drop_a.sum()) < n and int((drop_a & drop_b).sum()) == 0
    # no decisions at all: presence alone
    xa, xb, pa, pb, cls = front_pair(image, text)
    _, qa = modality_frontend(image)
    _, qb = modality_frontend(text)
    assert torch.equal(pa, qa) and torch.equal(pb, qb)
    assert not bool(torch.isnan(xa).any())            # (the Inf row of text counts as present: norm = inf > 1e-6)


def test_static_select_and_its_backward():
    from aecf_amd.xray import _StaticSelect
    dev = torch.device("cuda:0")
    n, w = 157, 512
    g = torch.Generator().manual_seed(4)
    parts = [torch.randn(n, w, generator=g).to(dev).requires_grad_() for _ in range(3)]
    cls = torch.randint(0, 4, (n,), generator=g).to(dev, torch.int32)
    out = _StaticSelect.apply(*parts, cls)
    ref = sum(torch.where((cls == c).unsqueeze(1), parts[c], torch.zeros((), device=dev)) for c in range(3))
    assert torch.equal(out, ref)
    d = torch.randn(n, w, generator=g).to(dev)
    out.backward(d)
    for c in range(3):
        assert torch.equal(parts[c].grad, torch.where((cls == c).unsqueeze(1), d, torch.zeros((), device=dev)))
