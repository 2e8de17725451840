#!/bin/bash
python -m pytest tests/test_xray_static_gpu.py -m gpu -x -q 2>&1 | tail -3
