#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_pool_gpu.py -m gpu -q -x -k "hilo or shard" 2>&1 | tail -5
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -k "two_ranks_equal or force_dp" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -30
timeout -k 10 300 python tools/debug/hilo_errors.py 65536 2>&1 | tail -2
EXTRA=--f32-params bash tools/gpu_jobs/ab_libs.sh 2 c2 main wide p24
