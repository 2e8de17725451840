#!/bin/bash
out=gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py --config c4 --batch 64 --graph --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$1', round(l['ms_per_step'],3), round(l['host_enqueue_ms'],3), round(l['device_tail_ms'],3))"; }
run default
TORCH_BLAS_PREFER_HIPBLASLT=0 run rocblas
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=$out/tunable.csv PYTORCH_TUNABLEOP_VERBOSE=0 run tunable
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=$out/tunable.csv run tuned_replay
ls $out | grep tunable
