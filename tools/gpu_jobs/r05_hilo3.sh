#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_pool_gpu.py -m gpu -q -x -k "hilo" 2>&1 | tail -3
AECF_LIB_PATH=$PWD/aecf_amd/lib/var/ringdef/libaecf_hip.so timeout -k 10 600 python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_shapes.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -k "two_ranks_equal" 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -12
EXTRA=--f32-params bash tools/gpu_jobs/ab_libs.sh 2 c2 main ring2
EXTRA= bash tools/gpu_jobs/ab_libs.sh 3 c2 main ringdef
