#!/bin/bash
python -m pytest tests/test_dp_gpu.py tests/test_xray_static_gpu.py tests/test_optim_gpu.py -m gpu -x -q 2>&1 | tail -3
python -m aecf_amd.train_xray --epochs 3 --switch-epoch 1 --samples 2048 --val-samples 512 2>&1 | grep -v amdgpu | tail -3 | cut -c1-250
