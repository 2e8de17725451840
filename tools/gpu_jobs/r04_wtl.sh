#!/bin/bash
python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_large.py -m gpu -x -q 2>&1 | tail -2
AECF_LIB_PATH=$PWD/build/var/wtl/libaecf_hip.so PYTHONPATH=$PWD python tools/debug/ws_timeline.py 2>&1 | grep -v amdgpu.ids | head -4 | cut -c1-130
tools/gpu_jobs/ab_libs.sh 2 c2 smoffscoff smoff sm4 main sm12
