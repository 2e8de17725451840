#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_large.py tests/test_pool_gpu_random.py tests/test_pool_gpu_shapes.py -q -x > $out/f_t.txt 2>&1; tail -5 $out/f_t.txt
echo "== A/B dsu"; tools/ab_env.sh AECF_DEBUG dsu_var=2 dsu_var=3 dsu_var=2 dsu_var=3 2>&1 | tee $out/f_ab.txt
