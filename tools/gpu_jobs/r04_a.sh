#!/bin/bash
# round 4, first GPU job: new tests, A/B against the round-3 build, kernel trace, full GPU suite
out=gpurun_out; mkdir -p $out
echo "== new tests"
timeout -k 10 600 python -m pytest -x -q tests/test_pool_gpu.py -k "in_kernel or drawn_in_the_kernel or partials or entropy_loss" > $out/a_newtests.txt 2>&1; tail -15 $out/a_newtests.txt
echo "== A/B c2"; tools/gpu_jobs/ab_base.sh 2 c2 2>&1 | tee $out/a_ab_c2.txt
echo "== A/B dsu"; tools/ab_env.sh AECF_DEBUG none dsu_narrow none dsu_narrow 2>&1 | tee $out/a_ab_dsu.txt
echo "== ktrace"; tools/ktrace.sh c2 24 2>&1 | tee $out/a_ktrace.txt
echo "== full gpu suite"; timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $out/a_fullsuite.txt 2>&1; tail -15 $out/a_fullsuite.txt
