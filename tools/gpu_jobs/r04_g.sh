#!/bin/bash
out=gpurun_out; mkdir -p $out
echo "== dsu_slab ablations"; tools/gpu_jobs/ab_libs.sh 1 c2 main dsu_NOX dsu_NODS dsu_NOU dsu_NODOT main 2>&1 | tee $out/g_abl.txt
