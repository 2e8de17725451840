#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_abi_guards_gpu.py -m gpu -q 2>&1 | tail -25
timeout -k 10 600 python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_shapes.py tests/test_pool_gpu_large.py -m gpu -q 2>&1 | tail -5
