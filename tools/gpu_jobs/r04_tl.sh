#!/bin/bash
python -m pytest tests/test_pool_gpu.py -m gpu -x -q 2>&1 | tail -2
AECF_LIB_PATH=$PWD/build/var/tl/libaecf_hip.so PYTHONPATH=$PWD python tools/debug/tn_timeline.py 2>&1 | grep -v amdgpu.ids
tools/gpu_jobs/ab_libs.sh 3 c2 prev main
tools/gpu_jobs/ab_libs.sh 2 c5 prev main
