#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_pool_gpu.py -q -x -k "hilo" > $out/n_t.txt 2>&1; tail -5 $out/n_t.txt
tools/gpu_jobs/suite.sh
