#!/bin/bash
out=gpurun_out; mkdir -p $out
echo "== vproj spread"; tools/gpu_jobs/ab_libs.sh 2 c2 main vsp2 vsp4 vsp6 2>&1 | tee $out/h_ab.txt
