#!/bin/bash
out=gpurun_out; mkdir -p $out
echo "== vproj ablations"; tools/gpu_jobs/ab_libs.sh 1 c2 main abl_NOSOFTMAX abl_NODMA abl_NOSCORES abl_NOSTORE abl_NOLDSREAD abl_ALL main 2>&1 | tee $out/e_abl.txt
