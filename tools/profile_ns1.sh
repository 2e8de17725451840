#!/bin/bash
# north_star's fusion shape (scores, softmax, statistics, mask, value projection, pooling AND out-projection in ONE forward
# kernel: aecf_row_fwd.hip, AECF_DEBUG=fused_fwd) against the default two-kernel forward, same box, C2:
#   gpurun_out/<tag>_c2_fusedfwd_{kernel_stats.csv,traffic.json,bench.json} and <tag>_c2_default_* beside them
set -o pipefail
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out; mkdir -p $out
for arm in default fusedfwd; do
  if [ $arm = fusedfwd ]; then export AECF_DEBUG=fused_fwd; else unset AECF_DEBUG; fi
  rocprofv3 --kernel-trace --stats -d $out/ks_$arm -o r -- python3 bench.py --settle-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/kernel_stats_from_db.py $out/ks_$arm/r_results.db $out/${tag}_c2_${arm}_kernel_stats.csv | head -6
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pf_$arm -o pmc -- python3 bench.py --settle-seconds 0 --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pw_$arm -o pmc -- python3 bench.py --settle-seconds 0 --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/traffic_from_pmc.py $out/pf_$arm $out/pw_$arm c2 $out/${tag}_c2_${arm}_traffic.json | tail -1
  python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/${tag}_c2_${arm}_bench.json 2> /dev/null
  python3 -c "
import json; l=json.load(open('$out/${tag}_c2_${arm}_bench.json')); s=l['stage_ms']
print('$arm', round(l['ms_per_step'],4), 'fwd.vproj', round(s['fwd.vproj']*1e3), 'fwd.outproj', round(s['fwd.outproj']*1e3))"
done
