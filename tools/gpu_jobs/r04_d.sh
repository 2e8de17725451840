#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_losses_gpu.py -q -k low_temperature > $out/d_t.txt 2>&1; tail -3 $out/d_t.txt
echo "== A/B PF"; tools/gpu_jobs/ab_libs.sh 2 c2 main vpf2 vpf5 wpf5 dxpf5 2>&1 | tee $out/d_ab_pf.txt
