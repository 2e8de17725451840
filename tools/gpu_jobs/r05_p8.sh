#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rocprofv3 --kernel-trace --stats -d $out/ksh -o r -- python3 bench.py --settle-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline --f32-params > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py $out/ksh/r_results.db $out/r05_c2_f32params_kernel_stats.csv | head -16
rm -rf $out/ksh
