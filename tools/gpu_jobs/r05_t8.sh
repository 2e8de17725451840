#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_pool_gpu.py tests/test_abi_guards_gpu.py tests/test_pool_gpu_shapes.py tests/test_xray_static_gpu.py -m gpu -q 2>&1 | tail -6
timeout -k 10 200 python tools/infer_time.py 2>&1 | tail -6
