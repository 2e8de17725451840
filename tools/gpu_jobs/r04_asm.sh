#!/bin/bash
python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_shapes.py tests/test_pool_gpu_large.py -m gpu -x -q 2>&1 | tail -2
tools/gpu_jobs/ab_libs.sh 3 c2 noasm main
