#!/bin/bash
# the one-launch hi/lo weight-gradient products: parity, per-tensor errors at the headline shape, and the step both ways
out=gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_pool_gpu.py -m gpu -q -x -k "hilo or grad_scale or shard or in_kernel" 2>&1 | tail -15
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -k "two_ranks_equal" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -60
timeout -k 10 600 python -m pytest tests/test_pool_gpu_large.py -m gpu -q 2>&1 | tail -15
timeout -k 10 300 python tools/debug/hilo_errors.py 65536 2>&1 | tail -4
summ='import sys,json; l=json.loads(sys.stdin.read()); s=l["stage_ms"]; print(sys.argv[1], round(l["ms_per_step"],4), round(l["ms_per_step_median"],4), " ".join("%s=%.0f"%(k.split(".")[1],x*1e3) for k,x in s.items() if x>0.003))'
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$summ" default
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --f32-params 2>/dev/null | python -c "$summ" hilo_f32params
done
