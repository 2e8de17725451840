#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_large.py tests/test_pool_gpu_random.py tests/test_pool_gpu_shapes.py -q -x > $out/p_t.txt 2>&1; tail -3 $out/p_t.txt
for c in c3 c5; do for e in "" no_ws; do echo "== $c $e"; AECF_DEBUG=$e timeout -k 10 300 python bench.py --config $c --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(round(l['ms_per_step'],4), round(l['ms_per_step_median'],4), {k:round(v*1e3) for k,v in l['stage_ms'].items()})"; done; done
