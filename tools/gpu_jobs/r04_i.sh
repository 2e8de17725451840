#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_large.py tests/test_pool_gpu_random.py tests/test_pool_gpu_shapes.py tests/test_dp_gpu.py -q -x > $out/i_t.txt 2>&1; tail -4 $out/i_t.txt
echo "== dw_v / dx ablations"; tools/gpu_jobs/ab_libs.sh 1 c2 main tn_NODMA tn_NOX tn_NOPOOL tn_NOMMA dx_NODMA dx_NOSTORE dx_NOWEIGHT dx_NOSTAGE main 2>&1 | tee $out/i_abl.txt
for c in c3 c5; do echo "== $c"; timeout -k 10 300 python bench.py --config $c --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(round(l['ms_per_step'],4), {k:round(v*1e3) for k,v in l['stage_ms'].items()})"; done
