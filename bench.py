#!/usr/bin/env python3
"""bench.py -- fused samples/sec (forward+backward) of the AECF fusion path on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N>1 launched with torch.distributed.run, one rank
per GPU over RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], the config the metric is quoted on): synthetic
x[B=65536, M=3, d=512] bf16, 8 heads, mask_prob 0.15, curriculum masking in training mode.
One "step" = pool(query.expand(B), x, return_info=True) + entropy_loss(info['entropy']) + backward with a
resident upstream gradient dy, all inputs resident in HBM.  N>1: one process per GPU; `--scaling weak` (default)
gives every rank its own B samples, `--scaling strong` shards the global B=65536 across the ranks; the parameter
gradients (4E^2+5E values) are all-reduced over RCCL inside the step (ONE collective, in place, after the backward; `--overlap`
issues it behind the dx kernel instead -- experimental) and every rank consumes ITS rows of one global uniform tensor for the
curriculum mask (N-rank masks == 1-rank masks; drawing the global tensor is inside the timed step: ~4 us at N = 1, ~10 us at N = 8).

`--config c3 --contrastive` (BASELINE configs[2]: "2-modality d=768 with cross-batch contrastive all-gather"): the step is the
pool forward on the rank's [8192, 2, 768] rows, the symmetric InfoNCE of the fused rows against the 65536 gathered rows of the
paired view (both directions from one logits block, entropy regulariser riding along: losses.gathered_contrastive_entropy_loss)
and the backward through both.  N = 1 emulates the all-gather with resident unit-norm keys (this rank's 8192 rows of the
paired view sit at their offset among them); N > 1 gathers them (dp.all_gather_rows) every step.

`python bench.py --gpus N` without a launcher starts the N ranks itself (child processes, started before this process
touches the GPU) and relays rank 0's JSON line; under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# HIP runtime knob AMD recommends for MI300-class parts: kernel arguments are written straight to device memory instead of
# host-visible memory, which shortens every kernel's start (measured here: 0.605 -> 0.585 ms per step, 2 x A/B on one
# box).  Read when the HIP runtime initialises, so it is set before torch touches the device; an explicit setting wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (B per GPU, M, E, H, dtype, mask_prob)
    "c2": (65536, 3, 512, 8, torch.bfloat16, 0.15),
    "c5": (16384, 4, 1024, 8, torch.bfloat16, 0.15),      # configs[4] per-GPU shard (B=131072 / 8)
    "c3": (8192, 2, 768, 8, torch.bfloat16, 0.15),
    "tiny": (4096, 3, 128, 4, torch.bfloat16, 0.15),
    "d256": (65536, 3, 256, 4, torch.bfloat16, 0.15),     # a narrower embedding (the example model's pool width), headline batch
    "d384": (65536, 3, 384, 6, torch.bfloat16, 0.15),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16
PROFILE_TAG = "r04"              # profiles/<tag>_<config>_traffic.json: the rocprofv3 --pmc passes of this round
STAGE_PASS_STEPS = 30            # steps of the per-stage HIP-event pass (roofline), run AFTER the timed region (warm)
SETTLE_STEPS = 30                # untimed steps of a fresh process before the W warm-up steps (stated in the line) ...
SETTLE_SECONDS = 2.0             # ... continued until the process has been on the GPU this long: the host's launch path gets ~25 %
                                 # faster ~1.5 s into a process on this pool (tools/debug/warm_trend.py; wall-clock, not step count);
                                 # only host-bound shards (configs[2] / configs[3]) notice, the device-bound headline step is flat
T_PROCESS = time.perf_counter()   # (re-stamped when the device is first touched)


def settle(one_step, extra, by_clock):
    """SETTLE_STEPS + `extra` untimed steps and -- one rank only: with collectives in the step every rank has to run the same
    number -- at least until SETTLE_SECONDS after process start; returns the settle steps run."""
    n = 0
    while n < SETTLE_STEPS or (by_clock and time.perf_counter() - T_PROCESS < SETTLE_SECONDS and n < 20000):   # (--settle-seconds)
        one_step()
        n += 1
    for _ in range(extra):
        one_step()
    return n
NCE_KEYS = 65536                 # --contrastive: gathered keys of configs[2] (8 ranks x 8192 rows)
NCE_TEMPERATURE = 0.07


def make_inputs(cfg, device, seed_offset=0):
    B, M, E, H, dtype, p = cfg
    import aecf_amd
    torch.manual_seed(2)
    query, pool = aecf_amd.create_fusion_pool(E, M, mask_prob=p, num_heads=H)
    with torch.no_grad():
        pool.attention.in_proj_bias.normal_(0.0, 0.02)
        pool.attention.out_proj.bias.normal_(0.0, 0.02)
        query.copy_(torch.randn(1, 1, E, generator=torch.Generator().manual_seed(1)) * (2.0 / E) ** 0.5)
    pool = pool.to(device=device, dtype=dtype)
    query = torch.nn.Parameter(query.detach().to(device=device, dtype=dtype))
    g = torch.Generator(device=device).manual_seed(seed_offset)      # every rank its own samples
    x = torch.randn(B, M, E, device=device, generator=g).to(dtype).requires_grad_(True)
    dy = torch.randn(B, 1, E, device=device, generator=g).to(dtype)
    pool.train()
    return pool, query, x, dy


def step(pool, query, x, dy, params, dp_on, uniforms=None, overlap=None):
    """One pass of the hot path over one resident batch: forward (+ entropy_loss) + backward (+ the gradient all-reduce
    when data-parallel: ONE collective, issued behind the backward's last kernel (dx) on a side stream, dp.GradOverlap;
    default: the same collective after the backward)."""
    B = x.shape[0]
    out, info = pool(query.expand(B, -1, -1), x, return_info=True, uniforms=uniforms)
    ent_loss = pool.curriculum_masking.entropy_loss(info["entropy"])
    x.grad = None
    for p in params:
        p.grad = None
    if dp_on and overlap is not None:
        with overlap:
            torch.autograd.backward([out], [dy])
            overlap.finish(params)
    else:
        torch.autograd.backward([out], [dy])
        if dp_on:
            # ONE RCCL all-reduce (AVG) of the 4E^2+5E values, in place over the allocation the backward wrote them into
            from aecf_amd.dp import all_reduce_grads
            all_reduce_grads(params)
    return out, ent_loss


class StageTimer:
    """hipEvents recorded by the library at its stage boundaries (include/aecf_hip.h profiling hook)."""

    def __init__(self):
        from aecf_amd import _lib
        self._lib = _lib
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventCreate.restype = ctypes.c_int
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventElapsedTime.restype = ctypes.c_int
        self.nf, self.nb = _lib.AECF_FWD_STAGES, _lib.AECF_BWD_STAGES
        self.fwd = (ctypes.c_void_p * (self.nf + 1))()
        self.bwd = (ctypes.c_void_p * (self.nb + 1))()
        for arr in (self.fwd, self.bwd):
            for i in range(len(arr)):
                ev = ctypes.c_void_p()
                assert self.hip.hipEventCreate(ctypes.byref(ev)) == 0
                arr[i] = ev
        lib = _lib.load()
        self.names = [("fwd." + lib.aecf_pool_stage_name(0, i).decode()) for i in range(self.nf)] + \
                     [("bwd." + lib.aecf_pool_stage_name(1, i).decode()) for i in range(self.nb)]
        self.samples = [[] for _ in range(self.nf + self.nb)]

    def arm(self):
        self._lib.stage_events_fwd = self.fwd
        self._lib.stage_events_bwd = self.bwd

    def disarm(self):
        self._lib.stage_events_fwd = None
        self._lib.stage_events_bwd = None

    def collect(self):
        ms = ctypes.c_float()
        k = 0
        for arr in (self.fwd, self.bwd):
            for i in range(len(arr) - 1):
                assert self.hip.hipEventElapsedTime(ctypes.byref(ms), arr[i], arr[i + 1]) == 0
                self.samples[k].append(ms.value)
                k += 1

    def median_ms(self):
        """Per-stage median over the pass (its first steps run while the device's clocks are still settling)."""
        return {n: sorted(v)[len(v) // 2] if v else 0.0 for n, v in zip(self.names, self.samples)}


def stage_model(cfg):
    """Algorithmic HBM bytes and executed MFMA flops per sample of each stage.  "Algorithmic" =
    SURVEY 8d's share of that stage: the path inputs it must read and the outputs it exists to produce -- tensors kept
    only to spare the backward a recompute are traffic (roofline.traffic shows them), not algorithm."""
    B, M, E, H, dtype, _ = cfg
    s = 2 if dtype == torch.bfloat16 else 4
    ku = (16 * M + 31) // 32
    return {
        "fwd.gate":    dict(bytes=s * M * E, flops=2 * M * E * 16 * (2 if s == 2 else 1)),
        "fwd.vproj":   dict(bytes=s * (M + 1) * E, flops=2 * M * E * E + 2 * M * E * 16 * 2),   # x -> o (scores fused)
        "fwd.outproj": dict(bytes=s * 2 * E, flops=2 * E * E),
        "bwd.dout":    dict(bytes=s * 2 * E, flops=2 * E * E),
        "bwd.dw_out":  dict(bytes=s * 2 * E, flops=2 * E * E),
        # score gradient + u = ds^T x from (do, x): P = W_v^T do per head (one E x E product), u by MFMA on the tile
        "bwd.dscore":  dict(bytes=s * (M + 1) * E, flops=2 * E * E + 2 * E * 16 * 2 * 2 * ku),
        "bwd.dx":      dict(bytes=s * (M + 1) * E, flops=2 * E * E),
        "bwd.dw_v":    dict(bytes=s * (M + 1) * E, flops=2 * E * E),
        "bwd.u":       dict(bytes=s * M * E, flops=0),          # separate pass only on shapes the fused kernel does not take
    }


def physical_cores():
    """(socket, core) pairs of /proc/cpuinfo that this process may run on -- hardware threads are not cores."""
    try:
        allowed = os.sched_getaffinity(0)
        cores, cpu, phys = set(), None, 0
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k = k.strip()
                if k == "processor":
                    cpu = int(v)
                elif k == "physical id":
                    phys = int(v)
                elif k == "core id" and cpu in allowed:
                    cores.add((phys, int(v)))
        if cores:
            return len(cores)
    except (OSError, ValueError, AttributeError):
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(cfg, seconds_budget=20.0):
    """The CPU oracle (oracle/aecf_oracle.py, a port of the reference's arithmetic) timed on this box's host
    cores on a bounded sample of the same workload: forward (+ train-mode masking + entropy loss) + explicit backward,
    float32 and bfloat16.  torch's intra-op pool is set to the thread count that measures fastest among {physical cores,
    32, 16, 8} (a 8192-row sample does not feed 64+ threads: the reference itself ran 24 k samples/s on 8 threads,
    BASELINE.md section 2); `cores` = physical cores available, `threads` = the pool size used."""
    from oracle import aecf_oracle as O
    B, M, E, H, dtype, p = cfg
    n = min(B, 8192)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, M, E, generator=g)
    q = torch.randn(1, 1, E, generator=g) * (2.0 / E) ** 0.5
    w_in = torch.randn(3 * E, E, generator=g) / E ** 0.5
    b_in = torch.randn(3 * E, generator=g) * 0.02
    w_out = torch.randn(E, E, generator=g) / E ** 0.5
    b_out = torch.randn(E, generator=g) * 0.02
    dy = torch.randn(n, 1, E, generator=g)
    U = torch.rand(n, 1, M, generator=g)

    def make(dt):
        xs, qs, wi, bi, wo, bo, dys = (t_.to(dt) for t_ in (x, q, w_in, b_in, w_out, b_out, dy))

        def one():
            qe = qs.expand(n, -1, -1)
            f = O.mha_forward(qe, xs, xs, wi, bi, wo, bo, H)
            m = O.curriculum_mask_train(f["wbar"].float(), U, p)
            O.entropy_loss(m["entropy"], M)
            O.mha_backward(qe, xs, xs, wi, bi, wo, H, f, dys, None)
        return one

    def best_of(fn, budget, max_reps):
        fn()
        best, reps, t_all = 1e30, 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t_all < budget and reps < max_reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
            reps += 1
        return best, reps

    cores = physical_cores()
    saved = torch.get_num_threads()
    one32 = make(torch.float32)
    trial = {}
    for nt in sorted({cores, 32, 16, 8}):
        if nt <= cores:
            torch.set_num_threads(nt)
            trial[nt] = best_of(one32, 0.5, 3)[0]
    threads = min(trial, key=trial.get)
    torch.set_num_threads(threads)
    best, reps = best_of(one32, seconds_budget * 0.5, 20)
    # "cores" = the threads actually used (the contract's meaning); the machine's physical / logical counts beside it
    out = dict(value=n / best, unit="samples/s", cores=threads, threads=threads, physical_cores=cores, logical_cpus=os.cpu_count(), kind="port",
               sample=f"{n} samples of the same [B,M={M},d={E}] workload, fp32, fwd (+ masking, entropy loss) + bwd, best of {reps}; "
                      f"thread sweep {({k: round(n / v) for k, v in trial.items()})} samples/s")
    try:      # bf16 leg (what torch's CPU bf16 kernels make of the same arithmetic), same step
        best_b, _ = best_of(make(torch.bfloat16), seconds_budget * 0.3, 5)
        out["value_bf16"] = n / best_b
    except Exception as e:                      # a CPU without usable bf16 kernels: the fp32 leg stands alone
        out["value_bf16"] = None
        out["bf16_note"] = str(e)[:80]
    torch.set_num_threads(saved)
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as child processes BEFORE this process
    has touched the GPU (it never does), relay their output, exit with the worst return code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def bench_c4(args):
    """BASELINE configs[3]: the example model around the pool (aecf_amd/xray.py; ref xrays/train_xrays_example.py:108-237) --
    image + text features [batch, 512] -> encoders -> presence routing -> pool (2 modalities, d = 256, 4 heads, curriculum
    masking and missing-modality training ON) -> classifier (15 labels), BCE, backward, AdamW (ref :360-377).  One step = one
    optimisation step on a resident synthetic batch; fp32 as the reference trains it.  The line reports samples/s and
    splits the step into the host's enqueue time and what is left for the device to finish."""
    import torch.distributed as dist
    from aecf_amd import dp
    from aecf_amd.xray import AECFModel, train_step
    from aecf_amd.optim import FusedAdamW
    from aecf_amd.train_xray import synthetic_split
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = local % max(torch.cuda.device_count(), 1)
    backend = None
    if world > 1:
        backend = os.environ.get("AECF_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(dev_index)
    globals()["T_PROCESS"] = time.perf_counter()
    device = torch.device("cuda", dev_index)
    B = args.batch
    image, text, labels = synthetic_split(B, 15, 512, 100 + rank, device)
    torch.manual_seed(0)
    model = AECFModel(512, 512, 15).to(device).train()
    model.toggle_curriculum(True)
    model.missing_modality_training = True
    params = list(model.parameters())
    if world > 1:
        dp.broadcast_parameters(params + list(model.buffers()))
    bucket = dp.FlatGradBucket(params) if world > 1 else None
    graphed = None
    if args.graph:
        if world > 1:
            sys.exit("--graph captures a one-rank step")
        from aecf_amd.xray import GraphedTrainStep
        opt = FusedAdamW(params, lr=1e-4, weight_decay=0.01)       # one launch, device-side step counters (aecf_adamw_step)
        crit = torch.nn.BCEWithLogitsLoss()
        graphed = GraphedTrainStep(model, opt, crit, B, 512, 512, 15, device, tune_gemm=not args.no_tune_gemm)
    else:
        opt = FusedAdamW(params, lr=1e-4, weight_decay=0.01)
        crit = torch.nn.BCEWithLogitsLoss()

    def one_step():
        if graphed is not None:
            return graphed(image, text, labels)
        return train_step(model, opt, crit, image, text, labels, bucket)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    settled = settle(one_step, args.warmup, world == 1)
    barrier()
    t0 = time.perf_counter()
    enqueue = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        one_step()
        enqueue += time.perf_counter() - h0
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0].item())
    if rank == 0:
        sec = elapsed / args.steps
        line = {
            "metric": "fused samples/sec (fwd+bwd)", "value": B * world / sec, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "settle_steps": settled, "effective_warmup": settled + args.warmup,
            "ms_per_step": sec * 1e3, "steps_per_s": 1.0 / sec, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"c4: example model step (encoders, presence routing, pool [M=2, d=256, 4 heads], classifier, BCE, "
                                   f"AdamW), batch {B} per GPU, curriculum masking + missing-modality training on"
                                   + (", the step replayed as ONE captured HIP graph (static routing)" if graphed is not None else ""),
                       "global_batch": B * world, "parallelism": f"dp{world}", "world_size": world,
                       "collectives": None if world == 1 else f"{backend}: one flat all-reduce of all {sum(p.numel() for p in params)} gradients"},
            "roofline": None,
            "gemm_selection": ("torch TunableOp: rocBLAS / hipBLASLt candidates timed per nn.Linear shape during the warm-up"
                               if graphed is not None and not args.no_tune_gemm else "torch default"),
            "host_enqueue_ms": enqueue / args.steps * 1e3,
            "device_tail_ms": max(0.0, (elapsed - enqueue) / args.steps * 1e3),
            "note": ("one graph replay per step: the host only copies the batch into the captured buffers and launches the graph; "
                     "what is left is the device walking the graph's kernel nodes" if graphed is not None else
                     "host-bound: the step is ~50 small launches (4 nn.Linear layers each way, routing, pool, AdamW) and one "
                     "device->host read of the routing counts; host_enqueue_ms ~ ms_per_step means the GPU waits for the host"),
            "cpu_baseline": None,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS) + ["c4"])
    ap.add_argument("--batch", type=int, default=64, help="--config c4: rows per step per GPU (the reference's 64)")
    ap.add_argument("--no-tune-gemm", action="store_true",
                    help="c4 --graph: leave the nn.Linear GEMMs on torch's default BLAS pick (default: TunableOp during warm-up)")
    ap.add_argument("--graph", action="store_true",
                    help="one rank: the whole step captured once as a HIP graph and replayed (c4: static routing; the pool "
                         "configurations: forward + entropy loss + backward; meant for the host-bound shards)")
    ap.add_argument("--settle-seconds", type=float, default=SETTLE_SECONDS,
                    help="one rank: settle steps continue until this long after the device was first touched (0: the fixed "
                         f"{SETTLE_STEPS} only -- what the profiling scripts pass, a counter pass records every launch)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = B per GPU fixed (default), strong = the config's B is the GLOBAL batch, sharded")
    ap.add_argument("--contrastive", action="store_true",
                    help="configs[2]: add the symmetric InfoNCE against 65536 gathered keys (+ entropy loss) to the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hilo", action="store_true",
                    help="AECF_HILO_GRADS: weight-gradient products on bf16 hi + lo operand pairs (float32-accurate parameter "
                         "gradients; a verification / high-accuracy mode, timed for the record)")
    ap.add_argument("--overlap", action="store_true",
                    help="N>1: issue the gradient all-reduce behind the backward's dx kernel on a side stream (dp.GradOverlap: "
                         "EXPERIMENTAL -- never run over RCCL on real multi-GPU hardware by this build; default: one all-reduce "
                         "after the backward)")
    ap.add_argument("--no-overlap", action="store_true", help="(default behaviour; kept for older command lines)")
    args = ap.parse_args()
    globals()["SETTLE_SECONDS"] = args.settle_seconds
    if args.hilo:
        from aecf_amd import layer as _layer
        _layer._HILO_GRADS = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.config == "c4":
        return bench_c4(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev_index = local % max(ndev, 1)          # rehearsal on a 1-GPU box: several ranks may share device 0
    backend = None
    if world > 1:
        import torch.distributed as dist
        # RCCL ("nccl") over xGMI is the real path; AECF_DIST_BACKEND=gloo only rehearses the N>1 code on one GPU
        backend = os.environ.get("AECF_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(dev_index)
    globals()["T_PROCESS"] = time.perf_counter()
    device = torch.device("cuda", dev_index)
    from aecf_amd import dp
    Bc, M, E, H, dtype, p = CONFIGS[args.config]
    if args.scaling == "strong" and world > 1:
        lo, hi = dp.shard_bounds(Bc, rank, world)
        B, B_global = hi - lo, Bc
    else:
        lo, B, B_global = rank * Bc, Bc, world * Bc
    cfg = (B, M, E, H, dtype, p)
    pool, query, x, dy = make_inputs(cfg, device, seed_offset=rank)
    params = [query] + list(pool.parameters())
    if world > 1:
        dp.broadcast_parameters(params)      # replicas start identical whatever the ranks' construction RNG did
        # (no dp.probe_avg_support here: the step uses divide + ReduceOp.SUM, which every backend takes; ReduceOp.AVG would
        #  save one 1 M-element divide per step and its start-up probe is one more thing that has never run on real RCCL)
    # curriculum-mask uniforms (public `uniforms=` argument).  Weak scaling: the global draw is DEFINED slot by slot -- rows
    # [r Bc, (r + 1) Bc) come from the generator seeded 1234 + r -- so a rank draws exactly its own Bc rows per step (per-rank work
    # does not grow with N) and any number of ranks, or one rank walking the slots, sees the same global tensor.  Strong scaling
    # (shard sizes change with N): every rank draws the same global tensor from the shared seed and uses its rows.
    ugen = None
    if world > 1:
        ugen = torch.Generator(device=device).manual_seed(1234 + (rank if args.scaling != "strong" else 0))
    overlap = dp.GradOverlap(params=params) if (world > 1 and args.overlap and not args.no_overlap) else None

    # --contrastive (configs[2]): the paired view.  zb_local = this rank's rows of the other view (resident, as the output of a
    # second tower would be); N = 1: the 65536 gathered unit-norm keys are resident too (rank 3 of 8: offset 3 x 8192) and the
    # local rows are written into their slots each step; N > 1: dp.all_gather_rows gathers them over RCCL each step.
    nce = None
    if args.contrastive:
        if args.config != "c3":
            sys.exit("--contrastive is defined for --config c3 (BASELINE configs[2])")
        from aecf_amd import losses
        gk = torch.Generator(device=device).manual_seed(77 + rank)
        zb_local = (0.8 * x.detach()[:, 0].float() + 0.6 * torch.randn(B, E, device=device, generator=gk)).to(dtype)
        if world == 1:
            keys = losses.l2_normalize(torch.randn(NCE_KEYS, E, device=device, generator=gk).to(dtype)).detach()
            nce = dict(offset=3 * B, keys=keys, cols=NCE_KEYS)
        else:
            nce = dict(offset=lo, keys=None, cols=B_global)
        nce["zb"] = zb_local

    def contrastive_step(u):
        """pool forward -> fused rows z; symmetric InfoNCE of z against every rank's rows of the paired view + entropy
        regulariser (one operator); backward through the loss and the pool (+ the gradient all-reduce when N > 1)."""
        out, info = pool(query.expand(B, -1, -1), x, return_info=True, uniforms=u)
        nb = losses.l2_normalize(nce["zb"])
        if world == 1:
            nb_all = nce["keys"]
            nb_all[nce["offset"]:nce["offset"] + B] = nb            # the gather's local slot
        else:
            nb_all = dp.all_gather_rows(nb, sizes=[B] * world if args.scaling != "strong" else None)
        loss = losses.gathered_contrastive_entropy_loss(out.squeeze(1), nb_all, nce["offset"], pool.curriculum_masking,
                                                        info["entropy"], temperature=NCE_TEMPERATURE)
        x.grad = None
        for prm in params:
            prm.grad = None
        loss.backward()
        if world > 1:
            from aecf_amd.dp import all_reduce_grads
            all_reduce_grads(params)
        return out, loss

    def one_step():
        u = None
        if ugen is not None:
            if args.scaling == "strong":
                u = torch.rand(B_global, 1, M, device=device, dtype=torch.float32, generator=ugen)[lo:lo + B]
            else:
                u = torch.rand(B, 1, M, device=device, dtype=torch.float32, generator=ugen)
        if nce is not None:
            return contrastive_step(u)
        return step(pool, query, x, dy, params, world > 1, u, overlap)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    graph = None
    if args.graph:
        if world > 1:
            sys.exit("--graph captures a one-rank step")
        eager_step = one_step
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            eager_step()

        def one_step():
            graph.replay()

    # A fresh process needs ~25 steps (~15 ms) before its step time settles (tools/step_trend.py: 0.66, 0.65 ... 0.59 ms; 300 ms
    # of unrelated device work beforehand does not shorten it): SETTLE_STEPS untimed steps come first and are reported in the line
    # ("settle_steps", "effective_warmup"), then the W warm-up steps, then exactly K timed steps.
    # the clock-based part of the settle only for host-bound call sizes: the device-bound configurations gain nothing from it
    # and their first timed step after the barrier gets slower behind a long settle (1.0-1.5 ms against 0.66-0.8 ms)
    settled = settle(one_step, args.warmup, world == 1 and B * M * E <= (1 << 25))
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step times (same stream)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        one_step()
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    per_step_raw = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if os.environ.get("AECF_BENCH_DUMP_STEPS"):
        print("step_ms", [round(v, 4) for v in per_step_raw], file=sys.stderr)
    per_step = sorted(per_step_raw)
    median_ms = per_step[len(per_step) // 2]
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed, median_ms], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, median_ms = float(tt[0].item()), float(tt[1].item())

    # per-stage durations (HIP events the library records on its launch stream), measured live over STAGE_PASS_STEPS extra steps
    # AFTER the timed region (device warm: the stage times are those of steps like the timed ones), per-stage median
    roofline = None
    stages = None
    st = StageTimer()                    # (every rank runs the pass -- the same tail for all -- rank 0 reports it)
    for _ in range(STAGE_PASS_STEPS):
        st.arm()
        step(pool, query, x, dy, params, False, None)
        st.disarm()
        torch.cuda.synchronize()
        st.collect()
    # N > 1, weak scaling (what the driver's one run per N measures): the STRONG-scaling point of the same N rides along in the
    # same line (field "strong_scaling"), after the timed region and the stage pass -- the config's global batch sharded over the
    # ranks, same step, same collective -- so that one lease of a multi-GPU node yields both curves
    strong = None
    if world > 1 and args.scaling == "weak" and nce is None and graph is None:
        import torch.distributed as dist
        lo_s, hi_s = dp.shard_bounds(Bc, rank, world)
        bs = hi_s - lo_s
        xs = x.detach()[:bs].clone().requires_grad_(True)
        dys = dy[:bs]
        sgen = torch.Generator(device=device).manual_seed(4321)

        def strong_step():
            us = torch.rand(Bc, 1, M, device=device, dtype=torch.float32, generator=sgen)[lo_s:hi_s]
            return step(pool, query, xs, dys, params, True, us, overlap)
        for _ in range(max(5, args.warmup)):
            strong_step()
        barrier()
        ts0 = time.perf_counter()
        for _ in range(args.steps):
            strong_step()
        barrier()
        tt = torch.tensor([time.perf_counter() - ts0], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sec_s = float(tt[0].item()) / args.steps
        strong = {"value": Bc / sec_s, "unit": "samples/s", "ms_per_step": sec_s * 1e3, "global_batch": Bc,
                  "per_gpu_batch": bs, "scaling": "strong", "steps": args.steps}
    nce_ms = None
    if nce is not None:                  # the two C-ABI calls of the loss side, HIP events on the launch stream
        from aecf_amd import _lib
        from aecf_amd.layer import _ptr, _stream
        lib = _lib.load()
        rows, cols = B, nce["cols"]
        na = losses.l2_normalize(x.detach()[:, 0].contiguous())
        nb_all = nce["keys"] if world == 1 else losses.l2_normalize(torch.randn(cols, E, device=device).to(dtype))
        f32 = dict(dtype=torch.float32, device=device)
        ws_bytes = lib.aecf_nce_sym_workspace_bytes(rows, cols, E)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        cs, lr = torch.empty(cols, **f32), torch.empty(rows, **f32)
        da, db = torch.empty(rows, E, dtype=dtype, device=device), torch.empty(cols, E, dtype=dtype, device=device)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t1, t2 = [], []
        for _ in range(10):
            ev[0].record()
            _lib.check(lib.aecf_nce_sym_pass1(rows, cols, E, NCE_TEMPERATURE, _ptr(na), _ptr(nb_all), _ptr(ws), ws_bytes, _ptr(cs),
                                              _stream()), "aecf_nce_sym_pass1")
            ev[1].record()
            _lib.check(lib.aecf_nce_sym_loss(rows, cols, nce["offset"], E, NCE_TEMPERATURE, _ptr(na), _ptr(nb_all), _ptr(cs), _ptr(ws),
                                             ws_bytes, _ptr(lr), 0, 2, 0.0, None, 1.0, None, None, _stream()), "aecf_nce_sym_loss")
            _lib.check(lib.aecf_nce_sym_grads(rows, cols, nce["offset"], E, NCE_TEMPERATURE, 0.5 / cols, _ptr(na), _ptr(nb_all),
                                              _ptr(ws), ws_bytes, None, _lib.AECF_BF16, _ptr(da), _ptr(db), _stream()),
                       "aecf_nce_sym_grads")
            ev[2].record()
            torch.cuda.synchronize()
            t1.append(ev[0].elapsed_time(ev[1]))
            t2.append(ev[1].elapsed_time(ev[2]))
        nce_ms = {"nce.pass1 (logits GEMM + exp + sums)": sorted(t1)[5], "nce.loss + nce.grads (normalisers, weights, da, db)": sorted(t2)[5]}
        del ws, da, db
    if rank == 0:
        stages = st.median_ms()
        model = stage_model(cfg)
        dom = max((k for k in stages if k in model), key=lambda k: stages[k])
        # HBM bytes per launch of that stage from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE,
        # corrected as MI355X_MICROARCH.md prescribes); measured at the config's full per-GPU batch
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{args.config}_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f)["bytes_per_launch"].get(dom)
            if traffic is not None and B != Bc:
                traffic = traffic * B / Bc
        t_s = stages[dom] * 1e-3
        gbs = model[dom]["bytes"] * B / t_s / 1e9
        tfl = model[dom]["flops"] * B / t_s / 1e12
        both = dict(hbm_frac=gbs / HBM_PEAK_GBS, mfma_frac=tfl / MFMA_PEAK_TFLOPS, alg_GBps=gbs, executed_TFLOPs=tfl)
        if gbs / HBM_PEAK_GBS >= tfl / MFMA_PEAK_TFLOPS:
            roofline = dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                            traffic=traffic, kernel=dom, kernel_ms=stages[dom], **both)
        else:
            roofline = dict(bound="mfma", achieved=tfl, peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=tfl / MFMA_PEAK_TFLOPS, traffic=traffic, kernel=dom, kernel_ms=stages[dom], **both)
        if nce_ms is not None:
            # the step's dominant kernel is the logits GEMM of pass 1: 2 rows cols d MFMA flops; its algorithmic bytes are the
            # two embedding matrices read + E written (rows x cols bf16)
            t1 = nce_ms["nce.pass1 (logits GEMM + exp + sums)"] * 1e-3
            fl = 2.0 * B * nce["cols"] * E
            by = 2.0 * (B * E + nce["cols"] * E + B * nce["cols"])
            roofline = dict(bound="mfma", achieved=fl / t1 / 1e12, peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=fl / t1 / 1e12 / MFMA_PEAK_TFLOPS, traffic=None,
                            kernel="nce_gemm_kernel<logits> (aecf_nce_sym_pass1; + the 30 us sum kernel)", kernel_ms=t1 * 1e3,
                            hbm_frac=by / t1 / 1e9 / HBM_PEAK_GBS, mfma_frac=fl / t1 / 1e12 / MFMA_PEAK_TFLOPS,
                            alg_GBps=by / t1 / 1e9, executed_TFLOPs=fl / t1 / 1e12,
                            loss_side_executed_TFLOPs=6.0 * B * nce["cols"] * E / (sum(nce_ms.values()) * 1e-3) / 1e12)
            stages = dict(stages, **nce_ms)

    if rank == 0:
        sec = elapsed / args.steps
        s_bytes = 2 if dtype == torch.bfloat16 else 4
        path_bytes = s_bytes * E * (3 * M + 2)                    # fwd+bwd algorithmic bytes per sample
        path_flops = sum(v["flops"] for v in stage_model(cfg).values())     # MFMA flops this decomposition executes
        if nce is not None:
            path_flops += 6.0 * nce["cols"] * E                   # per local sample: logits + the two gradient products
            path_bytes += s_bytes * (2 * E + 2 * nce["cols"]) + 4 * E        # z, dz, the E row written + re-read twice ... per sample
        cb = None
        if not (args.no_cpu_baseline or world > 1):
            cb = cpu_baseline(cfg)
            cb["cpu_model"] = cpu_model()
        workload = f"{args.config}: [B={B} per GPU, M={M}, d={E}, {H} heads] mask_prob={p} train-mode curriculum masking, fwd+bwd"
        coll = None if world == 1 else (
            f"{backend}: all-reduce of {4 * E * E + 5 * E} grads/step" +
            (" (one call, behind the dx kernel)" if overlap is not None and nce is None else " (one call, after the backward)"))
        if nce is not None:
            workload += (f" + symmetric InfoNCE of the fused rows against {nce['cols']} gathered rows of the paired view "
                         f"(T={NCE_TEMPERATURE}) + entropy loss, one logits block for both directions")
            if world > 1:
                coll += f"; all-gather of [{B},{E}] rows + reduce-scatter of their gradient; all-reduce of {nce['cols']} column sums"
        line = {
            "metric": "fused samples/sec (fwd+bwd)", "value": B_global / sec, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": settled,
            "effective_warmup": settled + args.warmup, "ms_per_step": sec * 1e3,
            "ms_per_step_median": median_ms,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "bf16" if dtype == torch.bfloat16 else "f32", "data": "synthetic",
            "config": {"workload": workload, "global_batch": B_global,
                       "parallelism": f"dp{world}", "world_size": world, "collectives": coll},
            "roofline": roofline,
            "path_hbm_frac": path_bytes * B / sec / 1e9 / HBM_PEAK_GBS,
            "path_mfma_frac": path_flops * B / sec / 1e12 / MFMA_PEAK_TFLOPS,
            "stage_ms": stages,
            "stage_pass": {"steps": STAGE_PASS_STEPS, "when": "after the timed region", "stat": "median"},
            "graph_replay": graph is not None,
            "cpu_baseline": cb,
            "strong_scaling": strong,
            "collective_library": None if world == 1 else (
                f"RCCL {'.'.join(str(v) for v in torch.cuda.nccl.version())}" if backend == "nccl" else backend),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                 # rank 0 is still measuring its stage times: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
