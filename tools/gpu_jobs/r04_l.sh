#!/bin/bash
out=gpurun_out; mkdir -p $out
echo "== plain GEMM copy spread"; tools/gpu_jobs/ab_libs.sh 2 c2 main wsp2 wsp4 wsp8 2>&1 | tee $out/l_ab.txt
