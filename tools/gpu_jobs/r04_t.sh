#!/bin/bash
out=gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_xray_static_gpu.py tests/test_pool_gpu.py tests/test_pool_gpu_shapes.py tests/test_mha_general_gpu.py -m gpu -x -q 2>&1 | tail -8
run() { python bench.py --config c4 --batch $2 $3 --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$1', $2, '$3', round(l['ms_per_step'],3), round(l['host_enqueue_ms'],3), round(l['device_tail_ms'],3))"; }
run default 64 --graph
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=$out/tunable.csv run tunable 64 --graph
run default 16384 --graph
run default 64 ""
rm -rf $out/kt_c4
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=$out/tunable.csv rocprofv3 --kernel-trace --stats -d $out/kt_c4 -o r -- python3 bench.py --settle-seconds 0 --config c4 --batch 64 --graph --steps 20 --warmup 5 > /dev/null 2>&1
python3 tools/kernel_stats_from_db.py $out/kt_c4/r_results.db $out/kt_c4_b64_graph.csv | head -40
rm -rf $out/kt_c4
