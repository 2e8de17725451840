#!/bin/bash
# tools/gpu_jobs/bench_modes.sh: every bench.py mode once (sanity after host-side changes)
out=gpurun_out
summ='import sys,json; l=json.loads(sys.stdin.read()); print(sys.argv[1], "ok", round(l["ms_per_step"],4), l["n_gpus"], l.get("collective_ms"), (l.get("strong_scaling") or {}).get("ms_per_step"))'
timeout -k 10 300 python bench.py --config c3 --contrastive --steps 20 --warmup 5 --no-cpu-baseline 2>$out/v1.err | tee $out/r05_c3nce_bench.json | python -c "$summ" c3_contrastive || tail -5 $out/v1.err
AECF_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --config tiny --steps 10 --warmup 3 --overlap 2>$out/v2.err | python -c "$summ" tiny_gloo2_overlap || tail -5 $out/v2.err
AECF_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --config c3 --contrastive --steps 5 --warmup 2 2>$out/v3.err | python -c "$summ" c3_contrastive_gloo2 || tail -5 $out/v3.err
timeout -k 10 300 python bench.py --config tiny --steps 10 --warmup 3 --hilo --no-cpu-baseline 2>$out/v4.err | python -c "$summ" tiny_hilo || tail -5 $out/v4.err
timeout -k 10 300 python bench.py --config c5 --steps 10 --warmup 3 --f32-params --no-cpu-baseline 2>$out/v5.err | python -c "$summ" c5_f32params || tail -5 $out/v5.err
timeout -k 10 300 python bench.py --config d256 --steps 10 --warmup 3 --f32-params --force-dp --no-cpu-baseline 2>$out/v6.err | python -c "$summ" d256_f32params_forcedp || tail -5 $out/v6.err
timeout -k 10 300 python bench.py --config c4 --batch 64 --graph --steps 50 --warmup 10 2>$out/v7.err | python -c "$summ" c4_graph || tail -5 $out/v7.err
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config tiny --steps 5 --warmup 2 2>$out/v8.err | python -c "$summ" torchrun2_rccl_same_gpu || tail -8 $out/v8.err
