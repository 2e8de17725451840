#!/bin/bash
out=gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_pool_gpu.py tests/test_pool_gpu_shapes.py tests/test_pool_gpu_random.py tests/test_xray_static_gpu.py tests/test_mha_general_gpu.py -m gpu -x -q 2>&1 | tail -4
run() { python bench.py --config c4 --batch $2 $3 --steps 300 --warmup 20 2>$out/err_$1.txt | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$1', $2, '$3', round(l['ms_per_step'],3), round(l['host_enqueue_ms'],3), round(l['device_tail_ms'],3))"; }
run tuned 64 --graph
run tuned 64 --graph
run tuned 64 --graph
