#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_pool_gpu_large.py tests/test_pool_gpu_shapes.py -q -x > $out/q_t.txt 2>&1; tail -3 $out/q_t.txt
for c in c5 c3; do echo "== $c"; tools/gpu_jobs/ab_libs.sh 1 $c main unw8 unw4 main; done 2>&1 | tee $out/q_ab.txt
