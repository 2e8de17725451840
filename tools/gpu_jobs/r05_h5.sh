#!/bin/bash
EXTRA=--f32-params bash tools/gpu_jobs/ab_libs.sh 3 c2 main pf pk pfpk
AECF_LIB_PATH=$PWD/aecf_amd/lib/var/pfpk/libaecf_hip.so timeout -k 10 300 python -m pytest tests/test_pool_gpu.py -m gpu -q -k hilo 2>&1 | tail -2
