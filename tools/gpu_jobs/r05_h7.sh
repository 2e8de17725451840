#!/bin/bash
AECF_LIB_PATH=$PWD/aecf_amd/lib/var/wmap/libaecf_hip.so timeout -k 10 300 python -m pytest tests/test_pool_gpu.py tests/test_abi_guards_gpu.py -m gpu -q -k "hilo or guards" 2>&1 | tail -2
EXTRA=--f32-params bash tools/gpu_jobs/ab_libs.sh 3 c2 main wmap
