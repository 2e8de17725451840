#!/bin/bash
# HBM traffic per stage of one config from two rocprofv3 --pmc passes: tools/traffic_only.sh c2 r02
cfg=${1:-c2}; tag=${2:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out; mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pf_$cfg -o pmc -- python3 bench.py --settle-seconds 0 --config $cfg --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pw_$cfg -o pmc -- python3 bench.py --settle-seconds 0 --config $cfg --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 tools/traffic_from_pmc.py $out/pf_$cfg $out/pw_$cfg $cfg $out/${tag}_${cfg}_traffic.json
rm -rf $out/pf_$cfg $out/pw_$cfg
