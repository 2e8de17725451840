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
PROFILE_TAG = "r05"              # profiles/<tag>_<config>_traffic.json: the rocprofv3 --pmc passes of this round
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


def make_inputs(cfg, device, seed_offset=0, f32_params=False):
    """f32_params: float32 MASTER parameters under bf16 activations (the usual mixed-precision training setup): the parameter
    gradients are then float32-stored and the backward's weight-gradient products run on bf16 hi + lo operand pairs."""
    B, M, E, H, dtype, p = cfg
    import aecf_amd
    torch.manual_seed(2)
    query, pool = aecf_amd.create_fusion_pool(E, M, mask_prob=p, num_heads=H)
    with torch.no_grad():
        pool.attention.in_proj_bias.normal_(0.0, 0.02)
        pool.attention.out_proj.bias.normal_(0.0, 0.02)
        query.copy_(torch.randn(1, 1, E, generator=torch.Generator().manual_seed(1)) * (2.0 / E) ** 0.5)
    pdt = torch.float32 if f32_params else dtype
    pool = pool.to(device=device, dtype=pdt)
    query = torch.nn.Parameter(query.detach().to(device=device, dtype=pdt))
    g = torch.Generator(device=device).manual_seed(seed_offset)      # every rank its own samples
    x = torch.randn(B, M, E, device=device, generator=g).to(dtype).requires_grad_(True)
    dy = torch.randn(B, 1, E, device=device, generator=g).to(dtype)
    pool.train()
    return pool, query, x, dy


def step(pool, query, x, dy, params, dp_on, shard=None, overlap=None):
    """One pass of the hot path over one resident batch: forward (+ entropy_loss) + backward (+ the gradient all-reduce
    when data-parallel: ONE sum collective after the backward -- the backward has already folded 1 / world into the gradients
    it stores, dp.attach -- or, `--overlap`, issued behind the backward's last kernel (dx) on a side stream, dp.GradOverlap).
    `shard` = (first row, global batch): the rank's rows of ONE global mask draw, evaluated inside the statistics kernel."""
    B = x.shape[0]
    out, info = pool(query.expand(B, -1, -1), x, return_info=True, batch_shard=shard)
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
            # ONE RCCL all-reduce (sum of pre-scaled values) of the 4E^2+5E gradients, over the run the backward wrote them into
            from aecf_amd.dp import all_reduce_grads
            all_reduce_grads(params, rehearse=True)
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

    def arm(self, pool):
        pool.options.stage_events = (self.fwd, self.bwd)

    def disarm(self, pool):
        pool.options.stage_events = None

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


def cpu_baseline(cfg, seconds_budget=30.0):
    """The CPU oracle (oracle/aecf_oracle.py, a port of the reference's arithmetic) timed on this box's host cores ON THE
    BENCHMARK'S OWN CONFIGURATION (SURVEY 8d / BASELINE.md section 3: same B, M, d; threads = physical cores): forward (+ train-mode
    masking + entropy loss) + explicit backward in float32, one warm-up + best of 3.  torch's intra-op pool: a quick sweep over
    {physical cores, 32, 16, 8} on an 8192-row sample picks a second candidate; the full batch is timed at the physical core
    count AND at the sweep's best when they differ, both reported, `value` = the faster.  The 8192-row numbers (float32 sweep
    and the bf16 leg -- torch's CPU bf16 kernels on the same arithmetic, ~5x slower than float32) stay in the line as `sample_8192`."""
    from oracle import aecf_oracle as O
    B, M, E, H, dtype, p = cfg
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, M, E, generator=g)
    q = torch.randn(1, 1, E, generator=g) * (2.0 / E) ** 0.5
    w_in = torch.randn(3 * E, E, generator=g) / E ** 0.5
    b_in = torch.randn(3 * E, generator=g) * 0.02
    w_out = torch.randn(E, E, generator=g) / E ** 0.5
    b_out = torch.randn(E, generator=g) * 0.02
    dy = torch.randn(B, 1, E, generator=g)
    U = torch.rand(B, 1, M, generator=g)

    def make(dt, n):
        xs, qs, wi, bi, wo, bo, dys = (t_.to(dt) for t_ in (x[:n], q, w_in, b_in, w_out, b_out, dy[:n]))
        Un = U[:n]

        def one():
            qe = qs.expand(n, -1, -1)
            f = O.mha_forward(qe, xs, xs, wi, bi, wo, bo, H)
            m = O.curriculum_mask_train(f["wbar"].float(), Un, p)
            O.entropy_loss(m["entropy"], M)
            O.mha_backward(qe, xs, xs, wi, bi, wo, H, f, dys, None)
        return one

    def best_of(fn, reps):
        fn()                                               # warm-up
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    t_start = time.perf_counter()
    cores = physical_cores()
    saved = torch.get_num_threads()
    ns = min(B, 8192)
    small32 = make(torch.float32, ns)
    trial = {}
    for nt in sorted({cores, 32, 16, 8}):
        if nt <= cores:
            torch.set_num_threads(nt)
            trial[nt] = best_of(small32, 3)
    sweep_best = min(trial, key=trial.get)
    full32 = make(torch.float32, B) if B > ns else small32
    full = {}
    for nt in ([cores] if sweep_best == cores else [cores, sweep_best]):
        torch.set_num_threads(nt)
        full[nt] = best_of(full32, 3) if B > ns else trial[nt]
        if time.perf_counter() - t_start > seconds_budget:
            break
    threads = min(full, key=full.get)
    out = dict(value=B / full[threads], unit="samples/s", cores=threads, threads=threads, physical_cores=cores,
               logical_cpus=os.cpu_count(), kind="port",
               sample=f"the full [B={B}, M={M}, d={E}] batch of this config, fp32, fwd (+ masking, entropy loss) + bwd, one warm-up + "
                      f"best of 3; samples/s by thread count {({k: round(B / v) for k, v in full.items()})}",
               sample_8192={"rows": ns, "threads_sweep_samples_per_s": {k: round(ns / v) for k, v in trial.items()}})
    if time.perf_counter() - t_start < seconds_budget:
        try:      # bf16 leg (what torch's CPU bf16 kernels make of the same arithmetic), 8192-row sample
            torch.set_num_threads(sweep_best)
            out["sample_8192"]["value_bf16"] = ns / best_of(make(torch.bfloat16, ns), 2)
        except Exception as e:                      # a CPU without usable bf16 kernels: the fp32 leg stands alone
            out["sample_8192"]["bf16_note"] = str(e)[:80]
    out["seconds"] = round(time.perf_counter() - t_start, 1)
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
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


_REAL_STDOUT = None


def emit(line):
    """The ONE JSON line, on the descriptor that was stdout when the process started (see main)."""
    text = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(text.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, text)


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
    ap.add_argument("--f32-params", action="store_true",
                    help="float32 master parameters under bf16 activations: float32-stored parameter gradients, the weight-gradient "
                         "products on hi + lo operand pairs (north_star's 1e-3 on the gradients as they are stored)")
    ap.add_argument("--force-dp", action="store_true",
                    help="one rank: run the N > 1 step (attach, sharded in-kernel mask draw, the collective) on a ONE-rank RCCL "
                         "group -- rehearses the RCCL call (and, with --graph, its capture) on a one-GPU box")
    ap.add_argument("--no-strong-graph", action="store_true",
                    help="N>1 weak scaling: do not attempt the graph-captured form of the strong-scaling point")
    ap.add_argument("--rows", type=int, default=0,
                    help="pool configurations: rows per GPU instead of the config's batch (strong-scaling shards on one GPU)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=30.0, help="budget of the CPU-oracle leg")
    args = ap.parse_args()
    globals()["SETTLE_SECONDS"] = args.settle_seconds
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner on stdout when its first communicator comes
    # up (seen on this image), other libraries may too -- from here on file descriptor 1 is the process's stderr, and the line
    # is written to the descriptor stdout had
    sys.stdout.flush()
    globals()["_REAL_STDOUT"] = os.dup(1)
    os.dup2(2, 1)
    if args.config == "c4":
        return bench_c4(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev_index = local % max(ndev, 1)          # rehearsal on a 1-GPU box: several ranks may share device 0
    backend = None
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        # RCCL ("nccl") over xGMI is the real path; AECF_DIST_BACKEND=gloo only rehearses the N>1 code on one GPU
        backend = os.environ.get("AECF_DIST_BACKEND", "nccl")
        if world == 1:                               # --force-dp: a one-rank group of our own
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(dev_index)
    globals()["T_PROCESS"] = time.perf_counter()
    device = torch.device("cuda", dev_index)
    from aecf_amd import dp
    Bc, M, E, H, dtype, p = CONFIGS[args.config]
    if args.rows > 0:
        Bc = args.rows
    if args.scaling == "strong" and world > 1:
        lo, hi = dp.shard_bounds(Bc, rank, world)
        B, B_global = hi - lo, Bc
    else:
        lo, B, B_global = rank * Bc, Bc, world * Bc
    cfg = (B, M, E, H, dtype, p)
    pool, query, x, dy = make_inputs(cfg, device, seed_offset=rank, f32_params=args.f32_params)
    if args.hilo:
        pool.options.hilo_grads = True
    params = [query] + list(pool.parameters())
    dp_on = world > 1 or args.force_dp
    shard = None
    if dp_on:
        if world > 1:
            dp.broadcast_parameters(params)  # replicas start identical whatever the ranks' construction RNG did
        # The N > 1 step is the one-rank step plus ONE collective: the backward folds 1 / world into the gradients as it stores
        # them and (bf16 parameters) leaves its float32 sums for the collective, which rounds the mean once into the allocation
        # autograd holds -- no divide launch, no cast in the backward (defer_rounding: this loop always reduces before it reads).
        dp.attach(pool, defer_rounding=True, world=max(world, 2) if args.force_dp else None)
        # curriculum mask: every rank seeds its device generator alike and names its rows of the global batch; the statistics
        # kernel evaluates the rank's elements of ONE global draw (ABI v9) -- no torch.rand launch, no uniforms tensor, and the
        # masks are those of one rank running the global batch
        torch.cuda.manual_seed(1234)
        shard = (lo, B_global)
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

    def contrastive_step():
        """pool forward -> fused rows z; symmetric InfoNCE of z against every rank's rows of the paired view + entropy
        regulariser (one operator); backward through the loss and the pool (+ the gradient all-reduce when N > 1)."""
        out, info = pool(query.expand(B, -1, -1), x, return_info=True, batch_shard=shard)
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
        if dp_on:
            from aecf_amd.dp import all_reduce_grads
            all_reduce_grads(params, rehearse=True)
        return out, loss

    def one_step():
        if nce is not None:
            return contrastive_step()
        return step(pool, query, x, dy, params, dp_on, shard, overlap)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    graph = None
    if args.graph:
        if world > 1 and backend != "nccl":
            sys.exit("--graph with N > 1 captures the RCCL collective with the step; the gloo rehearsal has no stream collectives")
        if overlap is not None:
            sys.exit("--graph and --overlap are alternatives")
        # N > 1 (round 5): the collective is captured with the step (RCCL enqueues on the capturing stream) -- UNTESTED on real
        # multi-GPU hardware by this build, exercised on one GPU with a one-rank RCCL group (--force-dp); inside a capture the
        # mask uniforms are the tensor path (torch's graph-safe generator state; the kernel cannot read a host-side offset)
        eager_step = one_step
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local" if dp_on else "global"):
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
        st.arm(pool)
        step(pool, query, x, dy, params, False, None)
        st.disarm(pool)
        torch.cuda.synchronize()
        st.collect()
    # N > 1, weak scaling (what the driver's one run per N measures): the STRONG-scaling point of the same N rides along in the
    # same line (field "strong_scaling"), after the timed region and the stage pass -- the config's global batch sharded over the
    # ranks, same step, same collective -- so that one lease of a multi-GPU node yields both curves
    # the collective on its own: HIP events on the launch stream around all_reduce_grads (which includes the one rounding of the
    # float32 mean into bf16 parameters' gradients), over extra steps after the timed region; median over the steps, max over ranks
    collective_ms = None
    if dp_on and nce is None:
        from aecf_amd.dp import all_reduce_grads
        cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(STAGE_PASS_STEPS)]
        for e0, e1 in cev:
            out, _ = pool(query.expand(B, -1, -1), x, return_info=True, batch_shard=shard)
            x.grad = None
            for p_ in params:
                p_.grad = None
            torch.autograd.backward([out], [dy])
            e0.record()
            all_reduce_grads(params, rehearse=True)
            e1.record()
        torch.cuda.synchronize()
        cms = sorted(e0.elapsed_time(e1) for e0, e1 in cev)
        collective_ms = cms[len(cms) // 2]
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([collective_ms], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            collective_ms = float(tt[0].item())
    strong = None
    if world > 1 and args.scaling == "weak" and nce is None and graph is None:
        import torch.distributed as dist
        lo_s, hi_s = dp.shard_bounds(Bc, rank, world)
        bs = hi_s - lo_s
        xs = x.detach()[:bs].clone().requires_grad_(True)
        dys = dy[:bs]
        torch.cuda.manual_seed(4321)                      # (every rank alike: the strong point's global mask draw)

        def strong_eager():
            return step(pool, query, xs, dys, params, True, (lo_s, Bc), overlap)

        def time_strong(fn):
            for _ in range(max(5, args.warmup)):
                fn()
            barrier()
            ts0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            barrier()
            tt = torch.tensor([time.perf_counter() - ts0], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt[0].item()) / args.steps
        sec_s = time_strong(strong_eager)
        strong = {"value": Bc / sec_s, "unit": "samples/s", "ms_per_step": sec_s * 1e3, "global_batch": Bc,
                  "per_gpu_batch": bs, "scaling": "strong", "steps": args.steps, "graph_replay": False}
        # ... and the same point with the step (collective included) captured once and replayed: at B / N rows per GPU the eager
        # step is bound by the host's ~0.2 ms of Python per step, which is not what strong scaling is asked about.  RCCL enqueues
        # on the capturing stream; this has never run on real multi-GPU hardware by this build, so any failure to capture keeps
        # the eager number (every rank still issues one collective per step either way)
        if backend == "nccl" and overlap is None and not args.no_strong_graph:
            captured, why = None, None
            try:
                side = torch.cuda.Stream(device=device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        strong_eager()
                torch.cuda.current_stream().wait_stream(side)
                captured = torch.cuda.CUDAGraph()
                # (thread_local: RCCL's proxy / watchdog threads may touch the runtime while this thread captures)
                with torch.cuda.graph(captured, capture_error_mode="thread_local"):
                    strong_eager()
            except Exception as e:                        # noqa: BLE001
                captured, why = None, f"{type(e).__name__}: {str(e)[:120]}"
            ok = torch.tensor([1 if captured is not None else 0], device=device, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)     # replay only if EVERY rank captured
            if int(ok.item()) == 1:
                sec_g = time_strong(captured.replay)
                strong["graph"] = {"value": Bc / sec_g, "ms_per_step": sec_g * 1e3, "graph_replay": True,
                                   "note": "step + RCCL all-reduce captured as one HIP graph (untested on RCCL before this run)"}
            else:
                strong["graph"] = {"value": None, "graph_replay": False, "note": why or "another rank failed to capture"}
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
            cb = cpu_baseline(cfg, args.cpu_baseline_seconds)
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
        # every hipEventRecord of the stage pass is a marker the stream has to process (~5 us): a stage with no launch in it
        # reads that much.  Reported as measured; `sum_ms` adds everything up, `launch_stages_sum_ms` leaves the empty stages
        # out -- the figure to hold against ms_per_step (it still carries one marker per non-empty stage)
        empty = [k for k, v in stages.items() if k.startswith(("fwd.", "bwd.")) and v < 0.008]
        pool_stage_sum = sum(v for k, v in stages.items() if k.startswith(("fwd.", "bwd.")))
        stage_pass = {"steps": STAGE_PASS_STEPS, "when": "after the timed region", "stat": "median",
                      "sum_ms": pool_stage_sum, "empty_stages": empty,
                      "event_overhead_ms_per_mark": (sum(stages[k] for k in empty) / len(empty)) if empty else None,
                      "launch_stages_sum_ms": pool_stage_sum - sum(stages[k] for k in empty)}
        pdt = next(iter(pool.parameters())).dtype
        hilo_on = pool.options.hilo_grads if pool.options.hilo_grads is not None else (pdt != dtype)
        grad_mode = ("float32-stored parameter gradients, weight-gradient products on bf16 hi + lo operand pairs (AECF_HILO_GRADS)"
                     if (hilo_on and dtype == torch.bfloat16) else
                     ("bf16-stored parameter gradients (bf16 parameters): float32 batch sums rounded once"
                      if pdt == torch.bfloat16 else "float32"))
        extra_launches = None
        if dp_on:
            extra_launches = {"mask_uniforms_draw": 0 if (graph is None and pool.options.draw_in_kernel) else 1,
                              "gradient_divide": 0, "cast_in_backward": 0,
                              "round_mean_into_bf16_grads": 1 if pdt == torch.bfloat16 else 0,
                              "collectives": 1}
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
            "stage_pass": stage_pass,
            "weight_grad_mode": grad_mode,
            "graph_replay": graph is not None,
            "cpu_baseline": cb,
            "collective_ms": collective_ms,
            "extra_launches_per_step": extra_launches,
            "strong_scaling": strong,
            "collective_library": None if world == 1 else (
                f"RCCL {'.'.join(str(v) for v in torch.cuda.nccl.version())}" if backend == "nccl" else backend),
        }
        emit(line)
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        dist.barrier()                 # rank 0 is still measuring its stage times: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
