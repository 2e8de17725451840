#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_large.py tests/test_pool_gpu_shapes.py -m gpu -q 2>&1 | tail -4
summ='import sys,json; l=json.loads(sys.stdin.read()); s=l["stage_ms"]; print(sys.argv[1], round(l["ms_per_step"],4), round(l["ms_per_step_median"],4), " ".join("%s=%.0f"%(k.split(".")[1],x*1e3) for k,x in s.items() if x>0.008))'
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$summ" default
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --f32-params 2>/dev/null | tee gpurun_out/r05_c2_f32params_bench.json | python -c "$summ" f32_params_hilo
done
