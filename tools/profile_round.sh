#!/bin/bash
# Regenerates the per-round artefacts under gpurun_out/ on an MI355X box (copy the ones to keep into profiles/):
#   tools/profile_round.sh r02     -> gpurun_out/r02_{c2,c3,c5}_{bench.json,traffic.json}, r02_c2_kernel_stats.csv
# rocprofv3 counter passes are separate runs (FETCH_SIZE / WRITE_SIZE do not fit one pass), kernel-trace only.
set -o pipefail
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
mkdir -p $out
for cfg in c2 c3 c5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pf_$cfg -o pmc -- python3 bench.py --settle-seconds 0 --config $cfg --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pw_$cfg -o pmc -- python3 bench.py --settle-seconds 0 --config $cfg --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/traffic_from_pmc.py $out/pf_$cfg $out/pw_$cfg $cfg $out/${tag}_${cfg}_traffic.json > /dev/null
  cp $out/${tag}_${cfg}_traffic.json profiles/${tag}_${cfg}_traffic.json      # bench.py reads roofline.traffic from here
  rm -rf $out/pf_$cfg $out/pw_$cfg                       # (the raw counter databases: tens of MB each; gpurun returns at most 64 MiB)
  echo "traffic $cfg done"
done
rocprofv3 --kernel-trace --stats -d $out/ks -o r -- python3 bench.py --settle-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py $out/ks/r_results.db $out/${tag}_c2_kernel_stats.csv | head -16
rm -rf $out/ks
# the N > 1 step on a one-rank RCCL group (what a rank launches per step: pool kernels + the collective + one rounding pass) and the
# float32-master-parameter step (hi + lo weight-gradient products), kernel traces
rocprofv3 --kernel-trace --stats -d $out/ksd -o r -- python3 bench.py --settle-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline --force-dp > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py $out/ksd/r_results.db $out/${tag}_c2_forcedp_kernel_stats.csv | head -16
rm -rf $out/ksd
rocprofv3 --kernel-trace --stats -d $out/ksh -o r -- python3 bench.py --settle-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline --f32-params > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py $out/ksh/r_results.db $out/${tag}_c2_f32params_kernel_stats.csv | head -14
rm -rf $out/ksh
python3 bench.py --steps 50 --warmup 10 > $out/${tag}_c2_bench.json 2> /dev/null
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --force-dp > $out/${tag}_c2_forcedp_bench.json 2> /dev/null
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --force-dp --graph > $out/${tag}_c2_forcedp_graph_bench.json 2> /dev/null
python3 bench.py --config c3 --steps 50 --warmup 10 --no-cpu-baseline --graph > $out/${tag}_c3shard_graph_bench.json 2> /dev/null
AECF_DEBUG=no_ws python3 bench.py --config c5 --steps 30 --warmup 10 --no-cpu-baseline > $out/${tag}_c5shard_no_ws_bench.json 2> /dev/null
python3 bench.py --config c3 --steps 50 --warmup 10 --no-cpu-baseline > $out/${tag}_c3shard_bench.json 2> /dev/null
python3 bench.py --config c5 --steps 50 --warmup 10 --no-cpu-baseline > $out/${tag}_c5shard_bench.json 2> /dev/null
for f in c2 c2_forcedp c2_forcedp_graph c3shard c3shard_graph c5shard c5shard_no_ws; do python3 -c "
import json; l=json.load(open('$out/${tag}_${f}_bench.json')); r=l['roofline']
print('$f', round(l['ms_per_step'],4), round(l['value']/1e6,1), 'M/s', r['kernel'], round(r['frac'],3), r['traffic'], round(l['path_hbm_frac'],3), round(l['path_mfma_frac'],3))"; done
# configs[3] (the example model): captured step and eager step at the reference's batch and at a device-bound batch
python3 bench.py --config c4 --batch 64 --graph --steps 300 --warmup 20 > $out/${tag}_c4_b64_graph_bench.json 2> /dev/null
python3 bench.py --config c4 --batch 16384 --graph --steps 100 --warmup 10 > $out/${tag}_c4_b16384_graph_bench.json 2> /dev/null
python3 bench.py --config c4 --batch 64 --steps 100 --warmup 10 > $out/${tag}_c4_b64_bench.json 2> /dev/null
python3 bench.py --config c4 --batch 16384 --steps 100 --warmup 10 > $out/${tag}_c4_b16384_bench.json 2> /dev/null
for f in c4_b64_graph c4_b16384_graph c4_b64 c4_b16384; do python3 -c "
import json; l=json.load(open('$out/${tag}_${f}_bench.json'))
print('$f', round(l['ms_per_step'],4), 'host', round(l['host_enqueue_ms'],3), 'tail', round(l['device_tail_ms'],3))"; done
rocprofv3 --kernel-trace --stats -d $out/ks4 -o r -- python3 bench.py --settle-seconds 0 --config c4 --batch 64 --graph --steps 20 --warmup 5 > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py $out/ks4/r_results.db $out/${tag}_c4_b64_graph_kernel_stats.csv | head -8
rm -rf $out/ks4
