#!/bin/bash
out=gpurun_out; mkdir -p $out
python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_shapes.py -m gpu -x -q 2>&1 | tail -3
tools/gpu_jobs/ab_libs.sh 3 c2 pl2 main
tools/gpu_jobs/ab_libs.sh 2 c3shard pl2 main
