"""Data-parallel fusion path on the GPU: two ranks run the FUSED kernels on their batch shards
and must reproduce the single-rank full-batch result -- gradients (after the one all-reduce) equal, masks bit-equal.

Two ranks on two GPUs use RCCL ("nccl"); on a one-GPU box both ranks share device 0 and rendezvous over gloo (the
collective payload is then staged through the host by torch, the kernels and the plumbing under test are the same).
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu

B, M, E, H = 1000, 3, 128, 4           # uneven-free global batch of the two-rank run; d_attn_w path exercised too


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _inputs(dtype):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(B, M, E, generator=g) * torch.tensor([1.0, 1.6, 2.2]).view(1, 3, 1)).to(dtype)
    dy = torch.randn(B, 1, E, generator=g).to(dtype)
    return x, dy


def _build(dev, dtype, seed):
    import aecf_amd
    torch.manual_seed(seed)
    query, pool = aecf_amd.create_fusion_pool(E, M, mask_prob=0.3, num_heads=H)
    with torch.no_grad():
        pool.attention.in_proj_bias.normal_(0, 0.05)
    pool = pool.to(dev, dtype).train()
    query = torch.nn.Parameter(query.detach().to(dev, dtype))
    return query, pool


def _run_shard(query, pool, x, dy, u, dev, scale, shard=None):
    xs = x.to(dev).requires_grad_(True)
    out, info = pool(query.expand(xs.shape[0], -1, -1), xs, return_info=True, uniforms=u, batch_shard=shard)
    ent = pool.curriculum_masking.entropy_loss(info["entropy"])
    loss = ((out.float() * dy.to(dev).float()).sum() / B + 1e-2 * info["attention_weights"].float().pow(2).sum() / B
            + 0.0 * ent) * scale
    loss.backward()
    return out.detach(), info["masked_attention_weights"].detach(), xs.grad / scale    # (a power of two: exact)


def _worker(rank, world, port, backend, dtype, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from aecf_amd import dp
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, dy = _inputs(dtype)
        query, pool = _build(dev, dtype, seed=50 + rank)           # replicas differ until the broadcast
        params = [query] + list(pool.parameters())
        dp.broadcast_parameters(params)
        dp.probe_avg_support(params[0].dtype, dev)
        st = dp.attach(pool)          # 1 / world folded into the backward's gradient stores; float32 sums kept for bf16 parameters
        assert st.world == world and st.grad_scale == 1.0 / world
        lo, hi = dp.shard_bounds(B, rank, world)
        u = dp.shard_batch(dp.global_uniforms(B, 1, M, seed=77, device=dev), rank, world)
        # the loss is a SUM over the shard divided by the global B, so the average over ranks needs the factor `world`
        out, masked, dx = _run_shard(query, pool, x[lo:hi], dy[lo:hi], u, dev, float(world))
        dp.all_reduce_grads(params)
        torch.cuda.synchronize()
        plain = [p.grad.clone() for p in params]
        # the same step with NO uniforms tensor: every rank seeds its generator alike and names its rows of the global batch;
        # the statistics kernel evaluates the rank's elements of the one global draw (ABI v9: philox_element0)
        for p in params:
            p.grad = None
        torch.cuda.manual_seed(4242)
        out_k, masked_k, dx_k = _run_shard(query, pool, x[lo:hi], dy[lo:hi], None, dev, float(world), shard=(lo, B))
        dp.all_reduce_grads(params)
        torch.cuda.synchronize()
        drawn = (out_k.float().cpu().numpy(), masked_k.float().cpu().numpy(), dx_k.float().cpu().numpy(),
                 [p.grad.float().cpu().numpy() for p in params])
        if dtype == torch.bfloat16:
            # defer_rounding: the bf16 tensors autograd holds stay uninitialised until the collective rounds the float32 mean
            # into them -- no cast launch in the backward, the same bits afterwards
            for p in params:
                p.grad = None
            st.defer_rounding = True
            _run_shard(query, pool, x[lo:hi], dy[lo:hi], u, dev, float(world))
            dp.all_reduce_grads(params)
            torch.cuda.synchronize()
            st.defer_rounding = False
            for a_, p in zip(plain, params):
                assert torch.equal(a_, p.grad), "deferred rounding differs from the cast-then-overwrite path" 
        # the same step with the out-projection gradients' all-reduce started behind the rest of the backward
        for p in params:
            p.grad = None
        overlap = dp.GradOverlap(params=params)
        with overlap:
            _run_shard(query, pool, x[lo:hi], dy[lo:hi], u, dev, float(world))
            overlap.finish(params)
        torch.cuda.synchronize()
        for a_, b_ in zip(plain, [p.grad for p in params]):
            assert torch.equal(a_, b_), "overlapped all-reduce differs from the single collective"
        # two fused backward calls inside one overlapped region (two pool applications feeding one loss): two collectives
        # are in flight, finish() waits for both; the gradients are the sum of the two backwards' reduced gradients
        for p in params:
            p.grad = None
        mid = (lo + hi) // 2
        overlap = dp.GradOverlap()
        with overlap:
            xa, xb = x[lo:mid].to(dev).requires_grad_(True), x[mid:hi].to(dev).requires_grad_(True)
            oa, _ = pool(query.expand(mid - lo, -1, -1), xa, return_info=True, uniforms=u[:mid - lo])
            ob, _ = pool(query.expand(hi - mid, -1, -1), xb, return_info=True, uniforms=u[mid - lo:])
            loss2 = ((oa.float() * dy[lo:mid].to(dev).float()).sum() + (ob.float() * dy[mid:hi].to(dev).float()).sum()) / B * world
            loss2.backward()
            assert len(overlap.pending) == 2
            overlap.finish(params)
        torch.cuda.synchronize()
        two = [p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        loss2 = None
        xa, xb = x[lo:mid].to(dev).requires_grad_(True), x[mid:hi].to(dev).requires_grad_(True)
        oa, _ = pool(query.expand(mid - lo, -1, -1), xa, return_info=True, uniforms=u[:mid - lo])
        ob, _ = pool(query.expand(hi - mid, -1, -1), xb, return_info=True, uniforms=u[mid - lo:])
        (((oa.float() * dy[lo:mid].to(dev).float()).sum() + (ob.float() * dy[mid:hi].to(dev).float()).sum()) / B * world).backward()
        dp.all_reduce_grads(params)
        torch.cuda.synchronize()
        tol2 = 1e-5 if dtype == torch.float32 else 1e-2       # (sum of two reduced gradients vs reduced sum: one more rounding)
        for a_, p in zip(two, params):
            err = (a_.float() - p.grad.float()).abs().max().item() / max(p.grad.float().abs().max().item(), 1e-12)
            assert err < tol2, err
        # micro-batches: two backward() calls accumulate into p.grad before ONE collective (with bf16 parameters and deferred
        # rounding the second call has to give the first one's tensors their values before autograd adds to them)
        for p in params:
            p.grad = None
        st.defer_rounding = dtype == torch.bfloat16
        _run_shard(query, pool, x[lo:mid], dy[lo:mid], u[:mid - lo], dev, float(world))
        _run_shard(query, pool, x[mid:hi], dy[mid:hi], u[mid - lo:], dev, float(world))
        dp.all_reduce_grads(params)
        torch.cuda.synchronize()
        st.defer_rounding = False
        for a_, p in zip(plain, params):
            err = (a_.float() - p.grad.float()).abs().max().item() / max(a_.float().abs().max().item(), 1e-12)
            assert err < tol2, ("micro-batches", err)
        # a backward that starts with gradients already set (accumulation): the hook is not installed, finish() reduces plainly
        overlap = dp.GradOverlap(params=params)
        with overlap:
            assert st.hook is None
        # bf16 transport (fp32=False) against the default float32 transport (`plain`)
        if dtype == torch.bfloat16:
            for p in params:
                p.grad = None
            _run_shard(query, pool, x[lo:hi], dy[lo:hi], u, dev, float(world))
            dp.all_reduce_grads(params, fp32=False)        # the gradients' own dtype on the wire: within a rounding or two of the default
            torch.cuda.synchronize()
            for a_, p in zip(plain, params):
                err = (a_.float() - p.grad.float()).abs().max().item() / max(a_.float().abs().max().item(), 1e-12)
                assert err < 1e-2, err
        for p, g in zip(params, plain):
            p.grad = g
        q.put((rank, lo, hi, out.float().cpu().numpy(), masked.float().cpu().numpy(), dx.float().cpu().numpy(),
               [p.grad.float().cpu().numpy() for p in params], [p.detach().float().cpu().numpy() for p in params], drawn))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_ranks_equal_one_rank(dtype):
    from aecf_amd import dp
    world = 2
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    # single rank, full batch, rank 0's parameters, the same global uniforms
    dev = torch.device("cuda:0")
    x, dy = _inputs(dtype)
    query, pool = _build(dev, dtype, seed=50)
    params = [query] + list(pool.parameters())
    for p, v in zip(params, res[0][7]):
        assert torch.equal(p.detach().float().cpu(), torch.from_numpy(v))            # broadcast: rank 0's values
    for a, b in zip(res[0][7], res[1][7]):
        assert (a == b).all()
    u = dp.global_uniforms(B, 1, M, seed=77, device=dev)
    out, masked, dx = _run_shard(query, pool, x, dy, u, dev, 1.0)
    torch.cuda.synchronize()
    # bf16: each rank rounds its float32 batch sums to bf16 once, the collective moves and sums them as float32 (the default
    # of all_reduce_grads for reduced-precision gradients) and rounds the mean once: at most three bf16 roundings against the
    # one-rank gradient (which carries one itself) -- 5e-3 of the largest element, not the 2e-2 a bf16 ring sum needed
    tol = 2e-5 if dtype == torch.float32 else 5e-3
    for rank, lo, hi, o_r, m_r, dx_r, grads, _, _ in res:
        assert torch.equal(torch.from_numpy(o_r), out[lo:hi].float().cpu())           # per-sample outputs bit-equal
        assert torch.equal(torch.from_numpy(m_r), masked[lo:hi].float().cpu())        # masks independent of N
        assert torch.equal(torch.from_numpy(dx_r), dx[lo:hi].float().cpu())
        for g, p in zip(grads, params):
            ref = p.grad.float().cpu()
            err = (torch.from_numpy(g) - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
            assert err < tol, (rank, tuple(ref.shape), err)
    # the in-kernel draw: one rank drawing the whole batch from the same seed sees, row for row, what the two ranks drew
    for p in params:
        p.grad = None
    torch.cuda.manual_seed(4242)
    out, masked, dx = _run_shard(query, pool, x, dy, None, dev, 1.0)
    torch.cuda.synchronize()
    assert 0.02 < float((masked == 0).float().mean()) < 0.9                           # (it did mask something)
    for rank, lo, hi, _, _, _, _, _, (o_r, m_r, dx_r, grads) in res:
        assert torch.equal(torch.from_numpy(o_r), out[lo:hi].float().cpu())
        assert torch.equal(torch.from_numpy(m_r), masked[lo:hi].float().cpu()), "N-rank masks differ from the one-rank masks"
        assert torch.equal(torch.from_numpy(dx_r), dx[lo:hi].float().cpu())
        for g, p in zip(grads, params):
            ref = p.grad.float().cpu()
            err = (torch.from_numpy(g) - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
            assert err < tol, (rank, tuple(ref.shape), err)


@pytest.mark.timeout(600)
def test_bench_gpus2_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher starts two ranks itself and reports n_gpus = 2 (one GPU: both ranks
    share it over gloo; two or more: RCCL)."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if torch.cuda.device_count() < 2:
        env["AECF_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "5",
                        "--warmup", "2", "--scaling", "strong"], env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["scaling"] == "strong"
    assert line["config"]["global_batch"] == 4096 and line["value"] > 0


@pytest.mark.timeout(600)
def test_bench_weak_scaling_line_carries_the_strong_point():
    """The driver runs ONE `bench.py --gpus N` per N (weak scaling): the same line also carries the strong-scaling point of
    that N and names the collective library, so one lease of a multi-GPU node yields both curves."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if torch.cuda.device_count() < 2:
        env["AECF_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "5",
                        "--warmup", "2"], env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 2 * 4096
    st = line["strong_scaling"]
    assert st["scaling"] == "strong" and st["global_batch"] == 4096 and st["per_gpu_batch"] == 2048 and st["value"] > 0
    assert line["collective_library"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("graph", [False, True])
def test_bench_force_dp_runs_the_rccl_step_on_one_gpu(graph):
    """`bench.py --force-dp`: the N > 1 step (dp.attach: pre-scaled gradients, float32 sums, the sharded in-kernel mask draw,
    ONE collective) on a one-rank RCCL group -- the RCCL call itself and, with --graph, its capture with the step into one HIP
    graph, which is what the strong-scaling point replays on a multi-GPU node."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "AECF_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny", "--steps", "5", "--warmup", "2", "--force-dp",
           "--no-cpu-baseline", "--settle-seconds", "0"] + (["--graph"] if graph else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["graph_replay"] == graph and line["value"] > 0
    assert line["collective_ms"] is not None and line["collective_ms"] > 0
    ex = line["extra_launches_per_step"]
    assert ex["gradient_divide"] == 0 and ex["cast_in_backward"] == 0 and ex["collectives"] == 1
    assert ex["mask_uniforms_draw"] == (1 if graph else 0)


@pytest.mark.timeout(600)
def test_trainer_two_ranks_learns_and_toggles():
    """The runnable trainer (aecf_amd/train_xray.py; ref xrays/train_xrays_example.py:312-377): 2 ranks, curriculum + missing-
    modality training switched on mid-run, loss goes down, validation mAP well above chance, toggled epochs report it."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < 2:
        env["AECF_DIST_BACKEND"] = "gloo"
    port = _free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "aecf_amd.train_xray", "--epochs", "6",
                        "--switch-epoch", "4", "--samples", "2048", "--val-samples", "512", "--batch", "128", "--lr", "2e-3"],
                       env=env, capture_output=True, text=True, timeout=560, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 6 and [x["curriculum"] for x in rows] == [False] * 4 + [True] * 2
    assert rows[3]["train_loss"] < 0.8 * rows[0]["train_loss"]
    assert rows[-1]["val_map"] > 0.4 and rows[-1]["gate_entropy"] > 0.0


def _nce_worker(rank, world, port, backend, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from aecf_amd import dp, losses
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, d = 640, 256
        g = torch.Generator().manual_seed(4)
        za = torch.randn(n, d, generator=g).to(torch.bfloat16)
        zb = (0.7 * za.float() + 0.6 * torch.randn(n, d, generator=g)).to(torch.bfloat16)
        lo, hi = dp.shard_bounds(n, rank, world)
        a = za[lo:hi].to(dev).requires_grad_(True)
        b = zb[lo:hi].to(dev).requires_grad_(True)
        loss = losses.info_nce(a, b, temperature=0.07)           # symmetric tile-GEMM form: all-gather of b, column-sum all-reduce
        loss.backward()
        torch.cuda.synchronize()
        # gradients follow the data-parallel convention (averaged over ranks later): undo the factor `world`
        q.put((rank, lo, hi, float(loss), (a.grad.float() / world).cpu().numpy(), (b.grad.float() / world).cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_info_nce_two_ranks_equal_one_rank():
    """Cross-batch negatives through dp.all_gather_rows + the column-sum all-reduce of the symmetric form: loss value and
    the gradients of both views on two ranks equal the one-rank full batch."""
    from aecf_amd import losses
    world = 2
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nce_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    dev = torch.device("cuda:0")
    n, d = 640, 256
    g = torch.Generator().manual_seed(4)
    za = torch.randn(n, d, generator=g).to(torch.bfloat16)
    zb = (0.7 * za.float() + 0.6 * torch.randn(n, d, generator=g)).to(torch.bfloat16)
    a = za.to(dev).requires_grad_(True)
    b = zb.to(dev).requires_grad_(True)
    want = losses.info_nce(a, b, temperature=0.07)
    want.backward()
    for rank, lo, hi, loss, ga, gb in res:
        assert abs(loss - float(want)) < 2e-3 * abs(float(want))          # every rank reports the global loss
        ra, rb = a.grad[lo:hi].float().cpu(), b.grad[lo:hi].float().cpu()
        assert (torch.from_numpy(ga) - ra).abs().max() / ra.abs().max() < 2e-2
        assert (torch.from_numpy(gb) - rb).abs().max() / rb.abs().max() < 2e-2
