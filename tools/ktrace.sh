#!/bin/bash
# kernel-trace summary of one bench config: tools/ktrace.sh c5
cfg=${1:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/kt_$cfg
rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$cfg -o r -- python3 bench.py --settle-seconds 0 --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py gpurun_out/kt_$cfg/r_results.db gpurun_out/kt_${cfg}.csv | head -${2:-22}
