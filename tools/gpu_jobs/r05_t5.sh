#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -12
timeout -k 10 600 python -m pytest tests/test_pool_gpu_shapes.py tests/test_pool_gpu_random.py -m gpu -q 2>&1 | tail -12
