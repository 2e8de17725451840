#!/bin/bash
out=gpurun_out; mkdir -p $out
bash tools/gpu_jobs/ab_libs.sh 2 c3 base main
bash tools/gpu_jobs/ab_libs.sh 2 c2 base main
bash tools/gpu_jobs/ab_libs.sh 1 c5 base main
bash tools/gpu_jobs/ab_libs.sh 1 d384 base main
bash tools/gpu_jobs/suite.sh > $out/r05_xcd_suite.txt 2>&1
grep -E "^==|passed|failed|^FAILED" $out/r05_xcd_suite.txt
