#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_large.py tests/test_pool_gpu_random.py tests/test_pool_gpu_shapes.py -q -x > $out/j_t.txt 2>&1; tail -3 $out/j_t.txt
echo "== dx deferred stores"; tools/gpu_jobs/ab_libs.sh 2 c2 main dx_atend 2>&1 | tee $out/j_ab.txt
