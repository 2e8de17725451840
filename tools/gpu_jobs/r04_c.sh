#!/bin/bash
out=gpurun_out; mkdir -p $out
tools/gpu_jobs/suite.sh
echo "== A/B c2"; tools/gpu_jobs/ab_base.sh 2 c2 2>&1 | tee $out/c_ab_c2.txt
echo "== A/B tn_map"; tools/ab_env.sh AECF_DEBUG tn_map=0 tn_map=1 tn_map=0 tn_map=1 2>&1 | tee $out/c_ab_tn.txt
echo "== ktrace"; tools/ktrace.sh c2 24 2>&1 | tee $out/c_ktrace.txt
