#!/bin/bash
out=gpurun_out; mkdir -p $out
echo "== full gpu suite"; timeout -k 10 1500 python -m pytest tests -m gpu -q > $out/b_fullsuite.txt 2>&1; tail -25 $out/b_fullsuite.txt
for v in 0 1; do echo "== pool tests dsu_var=$v"; AECF_DEBUG=dsu_var=$v timeout -k 10 600 python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_large.py -q -x > $out/b_var$v.txt 2>&1; tail -3 $out/b_var$v.txt; done
echo "== A/B c2"; tools/gpu_jobs/ab_base.sh 2 c2 2>&1 | tee $out/b_ab_c2.txt
echo "== A/B dsu"; tools/ab_env.sh AECF_DEBUG dsu_var=0 dsu_var=1 dsu_var=2 dsu_var=0 dsu_var=1 dsu_var=2 2>&1 | tee $out/b_ab_dsu.txt
echo "== ktrace"; tools/ktrace.sh c2 24 2>&1 | tee $out/b_ktrace.txt
