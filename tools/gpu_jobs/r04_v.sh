#!/bin/bash
out=gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tools/gpu_jobs/suite.sh > $out/suite.txt 2>&1; grep -E "^==|passed|failed|FAILED|Error" $out/suite.txt | head -40
run() { python bench.py --config c4 --batch $2 $3 --steps 200 --warmup 20 2>$out/err_$1.txt | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$1', $2, '$3', round(l['ms_per_step'],3), round(l['host_enqueue_ms'],3), round(l['device_tail_ms'],3))"; }
run tuned 64 --graph
run tuned 64 --graph
run tuned 16384 --graph
rm -rf $out/kt_c4
rocprofv3 --kernel-trace --stats -d $out/kt_c4 -o r -- python3 bench.py --settle-seconds 0 --config c4 --batch 64 --graph --steps 20 --warmup 5 > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py $out/kt_c4/r_results.db $out/kt_c4_b64_graph.csv | head -12
rm -rf $out/kt_c4
