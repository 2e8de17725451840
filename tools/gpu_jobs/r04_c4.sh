#!/bin/bash
out=gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for b in 64 16384; do
python bench.py --config c4 --batch $b --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('eager', $b, round(l['ms_per_step'],3), round(l['host_enqueue_ms'],3), round(l['device_tail_ms'],3))"
python bench.py --config c4 --batch $b --graph --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('graph', $b, round(l['ms_per_step'],3), round(l['host_enqueue_ms'],3), round(l['device_tail_ms'],3))"
done
rm -rf $out/kt_c4
rocprofv3 --kernel-trace --stats -d $out/kt_c4 -o r -- python3 bench.py --settle-seconds 0 --config c4 --batch 64 --graph --steps 20 --warmup 5 > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py $out/kt_c4/r_results.db $out/kt_c4_b64_graph.csv | head -70
rm -rf $out/kt_c4
