"""World-size-2 CPU (gloo) tests of the data-parallel layer (aecf_amd/dp.py): the collectives the N>1 path
uses are correct by construction -- flat-bucket gradient all-reduce equals the single-process full-batch
gradient, the embedding all-gather has the right backward, the global mask uniforms shard consistently.

The fused kernels themselves need a GPU; here a small torch model stands in for "the replicated model"
because only the plumbing around it is under test."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aecf_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        B = 22                                      # uneven on purpose: shards of 11/11, later 7/6/...
        x = torch.randn(B, 16)
        tgt = torch.randn(B, 8)
        model = _model()
        bucket = dp.FlatGradBucket(model.parameters())
        lo, hi = dp.shard_bounds(B, rank, world)
        bucket.zero()
        # sum-reduction loss divided by the GLOBAL batch: the averaged-over-ranks gradient times world equals
        # the full-batch gradient
        loss = ((model(x[lo:hi]) - tgt[lo:hi]) ** 2).sum() / B
        loss.backward()
        bucket.all_reduce(average=False)
        grads = [p.grad.clone() for p in model.parameters()]
        # embedding all-gather with autograd
        z = (x[lo:hi] @ torch.eye(16)[:, :4]).requires_grad_(True)
        z_all = dp.all_gather_rows(z)
        w = torch.arange(B * 4, dtype=torch.float32).reshape(B, 4)
        (z_all * w).sum().backward()
        # ... and with the row counts given by the caller (no size exchange): same rows, same gradient
        z2 = z.detach().clone().requires_grad_(True)
        sizes = [dp.shard_bounds(B, r, world)[1] - dp.shard_bounds(B, r, world)[0] for r in range(world)]
        z_all2 = dp.all_gather_rows(z2, sizes=sizes)
        (z_all2 * w).sum().backward()
        assert torch.equal(z_all2, z_all) and torch.equal(z2.grad, z.grad)
        u = dp.global_uniforms(B, 1, 3, seed=123, device="cpu")
        mean_stat = dp.all_reduce_mean_scalar(x[lo:hi].mean(), weight=hi - lo)
        # replicas built under different RNG states are made identical by broadcast_parameters
        torch.manual_seed(100 + rank)
        other = torch.nn.Linear(5, 3)
        dp.broadcast_parameters(list(other.parameters()))
        # uneven shards + per-shard MEAN loss: shard_loss_scale makes the unweighted average the global-mean gradient
        B2 = 23
        x2, t2 = torch.randn(B2, 16, generator=torch.Generator().manual_seed(7)), torch.randn(
            B2, 8, generator=torch.Generator().manual_seed(8))
        lo2, hi2 = dp.shard_bounds(B2, rank, world)
        m2 = _model()
        loss2 = ((m2(x2[lo2:hi2]) - t2[lo2:hi2]) ** 2).mean() * dp.shard_loss_scale(hi2 - lo2, B2, world)
        loss2.backward()
        avg_ok = dp.probe_avg_support(torch.float32, torch.device("cpu"))
        dp.all_reduce_grads(list(m2.parameters()))
        q.put((rank, [g.numpy() for g in grads], z_all.detach().numpy(), z.grad.numpy(),
               dp.shard_batch(u, rank, world).numpy(), float(mean_stat), (lo, hi),
               [p.detach().numpy() for p in other.parameters()], [p.grad.numpy() for p in m2.parameters()], avg_ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_dp_world2_matches_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0

    torch.manual_seed(0)
    B = 22
    x = torch.randn(B, 16)
    tgt = torch.randn(B, 8)
    model = _model()
    (((model(x) - tgt) ** 2).sum() / B).backward()
    ref = [p.grad for p in model.parameters()]
    u = dp.global_uniforms(B, 1, 3, seed=123, device="cpu")
    w = torch.arange(B * 4, dtype=torch.float32).reshape(B, 4)
    B2 = 23
    x2, t2 = torch.randn(B2, 16, generator=torch.Generator().manual_seed(7)), torch.randn(
        B2, 8, generator=torch.Generator().manual_seed(8))
    m2 = _model()
    ((m2(x2) - t2) ** 2).mean().backward()
    ref2 = [p.grad for p in m2.parameters()]
    for rank, grads, z_all, dz, u_shard, mean_stat, (lo, hi), bcast, grads2, avg_ok in res:
        assert not avg_ok                                                              # gloo: sum + divide, agreed by all
        for a, b in zip(bcast, res[0][7]):
            assert (a == b).all()                                                      # replicas identical to rank 0's
        for g, r in zip(grads2, ref2):
            assert torch.allclose(torch.from_numpy(g), r, rtol=1e-5, atol=1e-6)        # uneven shards, mean loss
        for g, r in zip(grads, ref):
            assert torch.allclose(torch.from_numpy(g), r, rtol=1e-5, atol=1e-6)       # all-reduced == full batch
        assert torch.allclose(torch.from_numpy(z_all), x[:, :4])                       # gathered in rank order
        assert torch.allclose(torch.from_numpy(dz), world * w[lo:hi])                  # every rank's loss sees z
        assert torch.equal(torch.from_numpy(u_shard), u[lo:hi])                        # masks independent of N
        assert abs(mean_stat - float(x.mean())) < 1e-6


def test_shard_bounds_cover_batch():
    for B in (1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_flat_bucket_views_and_single_process():
    model = _model()
    bucket = dp.FlatGradBucket(model.parameters())
    assert bucket.numel == sum(p.numel() for p in model.parameters())
    model(torch.randn(3, 16)).sum().backward()
    flat = next(iter(bucket.buffers.values()))
    off = 0
    for p in model.parameters():
        assert p.grad.data_ptr() == flat[off:].data_ptr()       # gradients live inside the bucket
        off += p.numel()
    assert bucket.all_reduce() == []                            # world 1: nothing to do
    bucket.zero()
    assert float(flat.abs().sum()) == 0.0


class _OneAllocationGrads(torch.autograd.Function):
    """Stands in for the fusion layer's backward: the parameter gradients are slices of ONE allocation
    (aecf_amd/layer.py:_PoolFunction.backward), which autograd keeps as p.grad without copying."""

    @staticmethod
    def forward(ctx, x, a, b):
        ctx.save_for_backward(x, a, b)
        return x @ a + b

    @staticmethod
    def backward(ctx, g):
        x, a, b = ctx.saved_tensors
        flat = torch.empty(a.numel() + b.numel())
        da, db = flat.split([a.numel(), b.numel()])
        da, db = da.view_as(a), db.view_as(b)
        torch.matmul(x.t(), g, out=da)
        torch.sum(g, 0, out=db)
        return None, da, db


def _alias_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        B = 10
        x = torch.randn(B, 6)
        a = torch.nn.Parameter(torch.randn(6, 4))
        b = torch.nn.Parameter(torch.randn(4))
        extra = torch.nn.Parameter(torch.randn(4))             # a parameter whose gradient lives elsewhere
        lo, hi = dp.shard_bounds(B, rank, world)
        (_OneAllocationGrads.apply(x[lo:hi], a, b).sum() / B * world).backward()
        aliased = dp.flat_grad_alias([a, b]) is not None
        ptr = a.grad.data_ptr()
        dp.all_reduce_grads([a, b])                              # in place over the shared allocation
        in_place = a.grad.data_ptr() == ptr
        # fallback: gradients that do not tile one allocation travel through a flat copy
        a2 = torch.nn.Parameter(a.detach().clone())
        ((x[lo:hi] @ a2 + extra).sum() / B * world).backward()
        not_aliased = dp.flat_grad_alias([a2, extra]) is None
        dp.all_reduce_grads([a2, extra])
        q.put((rank, aliased, in_place, not_aliased, a.grad.numpy(), b.grad.numpy(), a2.grad.numpy(), extra.grad.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_all_reduce_grads_in_place_over_one_allocation():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_alias_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    torch.manual_seed(0)
    x = torch.randn(10, 6)
    da = x.sum(0, keepdim=True).t().expand(6, 4) / 10           # d/da of mean over the batch of sum(x a + b)
    for rank, aliased, in_place, not_aliased, ga, gb, ga2, gextra in res:
        assert aliased and in_place and not_aliased
        assert torch.allclose(torch.from_numpy(ga), da, atol=1e-6)                    # averaged over ranks == full batch
        assert torch.allclose(torch.from_numpy(gb), torch.ones(4), atol=1e-6)
        assert torch.allclose(torch.from_numpy(ga2), da, atol=1e-6)
        assert torch.allclose(torch.from_numpy(gextra), torch.ones(4), atol=1e-6)


def _bf16_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        flat = (torch.randn(4096, generator=g) * 3.0).to(torch.bfloat16)          # one allocation, as the fused backward makes it
        a, b = torch.nn.Parameter(torch.zeros(60, 64, dtype=torch.bfloat16)), torch.nn.Parameter(torch.zeros(256, dtype=torch.bfloat16))
        ga, gb = flat.split([3840, 256])
        a.grad, b.grad = ga.view(60, 64), gb
        ptr = a.grad.data_ptr()
        mine = flat.float().clone()
        dp.all_reduce_grads([a, b])                                # default: float32 on the wire, ONE rounding afterwards
        q.put((rank, mine.numpy(), a.grad.float().numpy().reshape(-1), b.grad.float().numpy(), a.grad.data_ptr() == ptr,
               a.grad.dtype == torch.bfloat16))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bf16_gradients_travel_as_float32_by_default():
    """bf16 parameter gradients averaged in bf16 round at every hop of a ring; the default transport is
    float32 with one rounding of the mean, in place over the allocation the backward wrote."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bf16_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    mean = (torch.from_numpy(res[0][1]) + torch.from_numpy(res[1][1])) / 2        # exact in float32 for two bf16 addends
    want = mean.to(torch.bfloat16).float()
    for rank, _, ga, gb, in_place, is_bf16 in res:
        assert in_place and is_bf16
        got = torch.cat([torch.from_numpy(ga), torch.from_numpy(gb)])
        assert torch.equal(got, want), rank


class _StandInPool(torch.nn.Module):
    """What dp.attach needs of a pool module: parameters and ``_options()`` (aecf_amd/layer.py: PoolOptions)."""

    def __init__(self):
        super().__init__()
        from aecf_amd.layer import PoolOptions
        self.w = torch.nn.Parameter(torch.zeros(60, 64, dtype=torch.bfloat16))
        self.b = torch.nn.Parameter(torch.zeros(256, dtype=torch.bfloat16))
        self.options = PoolOptions()

    def _options(self):
        return self.options


def _attached_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pool = _StandInPool()
        st = dp.attach(pool, defer_rounding=True)
        g = torch.Generator().manual_seed(100 + rank)
        sums = torch.randn(4096, generator=g) * 3.0                  # this rank's float32 batch sums
        wide = sums * st.grad_scale                                  # ... as the backward stores them (aecf_pool_bwd_args.grad_scale)
        flat = torch.full((4096,), float("nan"), dtype=torch.bfloat16)     # deferred rounding: autograd's tensors are not written
        pool.w.grad, pool.b.grad = flat[:3840].view(60, 64), flat[3840:]
        st.record(flat, wide)
        ptr = pool.w.grad.data_ptr()
        dp.all_reduce_grads([pool.w, pool.b])
        consumed = not st.runs
        first = flat.float().clone()
        # gradients that autograd summed into fresh allocations (two pool applications in one backward): no float32 sums on
        # record, but still this module's parameters -- still no divide
        g2 = torch.Generator().manual_seed(200 + rank)
        va = (torch.randn(3840, generator=g2) * st.grad_scale).to(torch.bfloat16)
        vb = (torch.randn(256, generator=g2) * st.grad_scale).to(torch.bfloat16)
        pool.w.grad, pool.b.grad = va.clone().view(60, 64), vb.clone()
        dp.all_reduce_grads([pool.w, pool.b])
        second = torch.cat([pool.w.grad.reshape(-1).float(), pool.b.grad.float()])
        q.put((rank, st.grad_scale, sums.numpy(), first.numpy(), second.numpy(), torch.cat([va, vb]).float().numpy(),
               pool.w.grad.data_ptr() != ptr, consumed))
        dp.detach(pool)
        assert pool.options.dp is None and not dp._states_of([pool.w])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_attached_module_gradients_are_prescaled_and_rounded_once():
    """dp.attach (round 5): the backward stores sums / world and keeps them in float32; all_reduce_grads then issues a SUM with
    no divide, rounds the mean ONCE into the (until then unwritten) bf16 allocation, and consumes the record."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_attached_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    mean = (torch.from_numpy(res[0][2]) * 0.5 + torch.from_numpy(res[1][2]) * 0.5)
    want = mean.to(torch.bfloat16).float()
    want2 = (torch.from_numpy(res[0][5]) + torch.from_numpy(res[1][5])).to(torch.bfloat16).float()    # a plain SUM of pre-scaled values
    for rank, scale, _, first, second, _, fresh, consumed in res:
        assert scale == 0.5 and fresh and consumed
        assert torch.equal(torch.from_numpy(first), want), rank
        assert torch.equal(torch.from_numpy(second), want2), rank
